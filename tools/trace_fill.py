#!/usr/bin/env python3
"""Timeline of the persistent fill kernel from the trace build (make trace): per wavefront the
100 MHz wall-clock stamps start / fill-done / walk-done of each item it processed.
usage: COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python tools/trace_fill.py [pairs] [codons per pair]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
codons = int(sys.argv[2]) if len(sys.argv) > 2 else None
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *(host.synth_encoded(0, pairs) if codons is None else host.synth_encoded(0, pairs, n_codons=codons)))
tr = np.zeros(4096 * 16, np.uint64)
for _ in range(3):
    hip.load().coati_hip_debug_trace(tr.ctypes.data_as(C.c_void_p))  # read + clear
    batch.viterbi_launch()
    batch.sync()
f, w = batch.viterbi_timing()
rc = hip.load().coati_hip_debug_trace(tr.ctypes.data_as(C.c_void_p))
assert rc == 0
raw = tr.reshape(4096, 16)
used = raw[:, 0] > 0
hw = raw[used, 15]
tr = raw[used].astype(np.float64)
tr[:, 15] = 0
t0 = tr[:, 0].min()
us = lambda x: (x - t0) / 100.0  # 100 MHz -> microseconds
print(f"kernel {f*1e3:.0f} us by HIP events; {used.sum()} wavefronts traced")
print("wave start spread (us): min %.1f max %.1f" % (us(tr[:, 0]).min(), us(tr[:, 0]).max()))
n_items = ((tr[:, 1:15] > 0).sum(axis=1) // 2)
print("items per wave: ", dict(zip(*np.unique(n_items, return_counts=True))))
for k in range(int(n_items.max())):
    sel = n_items > k
    fill_end, walk_end = tr[sel, 1 + 2 * k], tr[sel, 2 + 2 * k]
    begin = tr[sel, 0] if k == 0 else tr[sel, 2 * k]
    print(f"item {k}: waves {sel.sum():5d}  fill {np.mean(fill_end-begin)/100:8.1f} us (min {np.min(fill_end-begin)/100:.1f} max {np.max(fill_end-begin)/100:.1f})"
          f"  walk {np.mean(walk_end-fill_end)/100:7.1f} us (max {np.max(walk_end-fill_end)/100:.1f})  ends at {us(walk_end).mean():8.1f} us (min {us(walk_end).min():.1f} max {us(walk_end).max():.1f})")
last = np.array([tr[i, 2 * n] for i, n in enumerate(n_items) if n > 0])
print("last stamp per wave (us): p5 %.1f p50 %.1f p95 %.1f max %.1f" % tuple(np.percentile(us(last), [5, 50, 95, 100])))

# where did each wave run?  HW_ID (gfx9): simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]
hw_id = (hw & 0xFFFFFFFF).astype(np.int64)
xcc = ((hw >> 32) & 0xF).astype(np.int64)
simd, cu, sh, se = (hw_id >> 4) & 3, (hw_id >> 8) & 15, (hw_id >> 12) & 1, (hw_id >> 13) & 7
place = xcc * 10000 + se * 1000 + sh * 100 + cu
print("distinct (xcc,se,sh,cu):", len(np.unique(place)), " waves per CU: ", dict(zip(*np.unique(np.unique(place, return_counts=True)[1], return_counts=True))))
key = place * 4 + simd
print("waves per SIMD: ", dict(zip(*np.unique(np.unique(key, return_counts=True)[1], return_counts=True))))
fill1 = (tr[:, 3] - tr[:, 2]) / 100.0  # second item's fill
ok = n_items > 1
for x in range(8):
    m = ok & (xcc == x)
    if m.any():
        print(f"xcc {x}: waves {m.sum():4d}  item-1 fill mean {fill1[m].mean():7.1f} us  min {fill1[m].min():7.1f}  max {fill1[m].max():7.1f}")
percu = {}
for pl in np.unique(place):
    m = ok & (place == pl)
    if m.any():
        percu[pl] = fill1[m].mean()
v = np.array(list(percu.values()))
print("per-CU mean of item-1 fill: min %.1f p50 %.1f max %.1f" % (v.min(), np.median(v), v.max()))

# ---- utilisation over time: wavefronts that hold an item (between a stamp pair), per 250 us
t_end = us(last).max()
edges = np.arange(0.0, t_end + 250.0, 250.0)
busy = np.zeros(len(edges) - 1)
filling = np.zeros(len(edges) - 1)
for i in range(tr.shape[0]):
    n = int(n_items[i])
    if n == 0:
        continue
    b, e = us(tr[i, 0]), us(tr[i, 2 * n])  # a wavefront is busy from its start to its last stamp
    lo, hi = np.clip(np.searchsorted(edges, [b, e]) - 1, 0, len(busy) - 1)
    for k in range(lo, hi + 1):
        busy[k] += max(0.0, min(e, edges[k + 1]) - max(b, edges[k])) / 250.0
print("busy wavefronts per 250 us bin:", " ".join("%d" % x for x in busy))
# when does each SIMD go idle for good?
simd_last = {}
for i in range(tr.shape[0]):
    if n_items[i] > 0:
        simd_last[key[i]] = max(simd_last.get(key[i], 0.0), us(tr[i, 2 * int(n_items[i])]))
sl = np.array(list(simd_last.values()))
print("SIMD's last stamp (us): p5 %.0f p25 %.0f p50 %.0f p75 %.0f p95 %.0f max %.0f; mean idle before the end %.0f us" %
      (*np.percentile(sl, [5, 25, 50, 75, 95, 100]), (sl.max() - sl).mean()))

# which wavefronts (launch index = blockIdx * waves per workgroup + wave) are the starved ones (one item in the whole launch)?
idx = np.nonzero(used)[0]
starved = idx[n_items == 1]
if len(starved):
    print("wavefronts with ONE item: %d; launch index quartiles %s; share with index >= 3072: %.2f" %
          (len(starved), np.percentile(starved, [0, 25, 50, 75, 100]).astype(int), float((starved >= 3072).mean())))
    rich = idx[n_items >= 3]
    print("wavefronts with >= 3 items: %d; launch index quartiles %s" % (len(rich), np.percentile(rich, [0, 25, 50, 75, 100]).astype(int)))
