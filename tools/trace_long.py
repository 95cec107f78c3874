#!/usr/bin/env python3
"""Timeline of ONE long pair in viterbi_lp from the trace build (make trace): when each strip's wavefront finished
its fill, and how long the traceback took.
usage: COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python tools/trace_long.py [codons]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host  # noqa: E402

codons = int(sys.argv[1]) if len(sys.argv) > 1 else 53334
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *host.synth_encoded(0, 1, n_codons=codons))
tr = np.zeros(4096 * 4, np.uint64)
lib = hip.load()
for _ in range(3):
    lib.coati_hip_debug_trace_lp(tr.ctypes.data_as(C.c_void_p))  # read + clear
    batch.viterbi_launch()
    batch.sync()
f, w = batch.viterbi_timing()
assert lib.coati_hip_debug_trace_lp(tr.ctypes.data_as(C.c_void_p)) == 0
raw = tr.reshape(4096, 4)
used = raw[:, 1] > 0
t = raw[used].astype(np.float64)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
order = np.argsort(t[:, 3])
fill_end = us(t[order, 1])
n = len(fill_end)
print(f"kernel {f:.2f} ms by HIP events; {n} strips traced; wave start spread {us(t[:, 0]).max():.1f} us")
print("first strip done at %.1f us; last strip done at %.1f us" % (fill_end[0], fill_end[-1]))
d = np.diff(fill_end)
print("lag between consecutive strips' ends (us): mean %.2f p5 %.2f p50 %.2f p95 %.2f max %.2f" % (d.mean(), *np.percentile(d, [5, 50, 95, 100])))
for lo in range(0, len(d), max(1, len(d) // 8)):
    print("  strips %4d..: mean lag %.2f us" % (lo, d[lo:lo + max(1, len(d) // 8)].mean()))
walk = t[:, 2] > 0
print("traceback: %.1f us" % ((t[walk, 2] - t[walk, 1]).max() / 100.0))
