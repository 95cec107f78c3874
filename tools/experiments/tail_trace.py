#!/usr/bin/env python3
"""The end of a viterbi_ck launch, from the trace build (make trace): how much wavefront time is idle before the
kernel ends, when the ticket queue ran dry, what the last items were.
usage: COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/experiments/tail_trace.py [pairs]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *host.synth_encoded(0, pairs))
tr = np.zeros(4096 * 16, np.uint64)
for _ in range(3):
    hip.load().coati_hip_debug_trace(tr.ctypes.data_as(C.c_void_p))
    batch.viterbi_launch()
    batch.sync()
f, _ = batch.viterbi_timing()
assert hip.load().coati_hip_debug_trace(tr.ctypes.data_as(C.c_void_p)) == 0
raw = tr.reshape(4096, 16)
used = raw[:, 0] > 0
hw = raw[used, 15]
st = raw[used, :15].astype(np.float64)
t0 = st[:, 0].min()
st = np.where(st > 0, (st - t0) / 100.0, np.nan)  # us
n_items = (np.isfinite(st[:, 1:]).sum(axis=1) + 1) // 2
last = np.nanmax(st, axis=1)
T = last.max()
print(f"kernel {f*1e3:.0f} us by events, last stamp {T:.0f} us; items per wave: {dict(zip(*np.unique(n_items, return_counts=True)))} (7 = seven or more: the last record is shared)")
print(f"idle wavefront time before the end: {np.sum(T - last) / (len(last) * T) * 100:.2f} % of wavefronts x time "
      f"(mean {np.mean(T - last):.0f} us; p50 {np.percentile(T - last, 50):.0f} p90 {np.percentile(T - last, 90):.0f})")
# start of each wave's LAST item and its length
starts, lens = [], []
for w in range(st.shape[0]):
    n = n_items[w]
    if n == 0 or n >= 7:  # (from the seventh item on the record is shared: the item's start is not known)
        continue
    starts.append(st[w, 2 * (n - 1)])
    lens.append(st[w, 2 * n] - st[w, 2 * (n - 1)] if np.isfinite(st[w, 2 * n]) else np.nan)
starts, lens = np.array(starts), np.array(lens)
print(f"last items: start p5 {np.percentile(starts,5):.0f} p50 {np.percentile(starts,50):.0f} p95 {np.percentile(starts,95):.0f} max {starts.max():.0f} us (~ when the queue ran dry); "
      f"length p5 {np.nanpercentile(lens,5):.0f} p50 {np.nanpercentile(lens,50):.0f} p95 {np.nanpercentile(lens,95):.0f} us")
hw_id = (hw & 0xFFFFFFFF).astype(np.int64)
xcc = ((hw >> 32) & 0xF).astype(np.int64)
simd_key = (xcc * 10000 + ((hw_id >> 13) & 7) * 1000 + ((hw_id >> 12) & 1) * 100 + ((hw_id >> 8) & 15)) * 4 + ((hw_id >> 4) & 3)
bins = np.arange(max(0.0, T - 1500), T + 50, 50.0)
active = [(last > b).sum() for b in bins]
print("busy wavefronts at t (us):", " ".join(f"{int(b)}:{a}" for b, a in zip(bins, active)))
# per SIMD: how many of its wavefronts are still busy, averaged
keys = np.unique(simd_key)
lastm = np.array([np.sort(last[simd_key == k]) for k in keys if (simd_key == k).sum() == 4])
print("per SIMD, finishing times of its 1st/2nd/3rd/4th wavefront to finish (mean us):", np.round(lastm.mean(axis=0), 0),
      " spread of the SIMD's last: p5 %.0f p95 %.0f" % tuple(np.percentile(lastm[:, 3], [5, 95])))
# throughput model: fraction of SIMD-time with k busy wavefronts over the last ms
for k in range(0, 5):
    frac = np.mean([(np.sum(lastm > b, axis=1) == k).mean() for b in np.arange(T - 1000, T, 10.0)])
    print(f"  last 1000 us: SIMDs with {k} busy wavefronts: {frac*100:.1f} %")
