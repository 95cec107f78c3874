#!/usr/bin/env python3
"""Wall time of ONE coati_hip_viterbi_batch call (host arrays in, host arrays out) on small inputs -- what a caller
that aligns a pair at a time through the C ABI pays per call.  usage: oneshot_latency.py [pairs ...]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host

model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
for n in [int(x) for x in sys.argv[1:]] or [1, 4, 16, 64, 256, 1024]:
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n)
    ts = []
    for r in range(30):
        t0 = time.perf_counter()
        model.viterbi(a_cat, a_off, b_cat, b_off)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e3
    print(f"{n:5d} pairs of 1 kb: one-shot call median {np.median(ts):.3f} ms (min {ts.min():.3f})  = {np.median(ts) / n * 1e3:.1f} us per pair")
