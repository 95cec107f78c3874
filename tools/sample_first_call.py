#!/usr/bin/env python3
"""BASELINE configs[3] as `coati sample` pays it: ONE forward + ONE sampleback call on a fresh model (16 pairs x 1 000
samples), result arrays fresh (first touched under the download) or already touched (a C++ caller's zero-filled vectors),
next to the warm repeat.  usage: sample_first_call.py [reps]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
table, consts = host.set_subst("mar-mg"), host.gap_consts()
enc = host.synth_encoded(0, 16)
states = np.array([host.rng_seed(["42"]) for _ in range(16)], np.uint64)
warm = hip.Model(table, consts, 1); wb = hip.Batch(warm, *enc); wb.forward_launch(); wb.sync(); wb.close(); warm.close()  # (HIP bring-up)
for touched, prepare in ((False, False), (True, False), (True, True)):
    rows = []
    for _ in range(reps):
        m = hip.Model(table, consts, 1)
        bt = hip.Batch(m, *enc)
        t0 = time.perf_counter(); bt.forward_launch()
        if prepare: bt.sampleback_prepare(1000)  # (the sampler's allocations under the Forward kernel: coati-sample does this)
        bt.sync(); t_f = time.perf_counter() - t0
        out = None
        if touched:
            total = int(1000 * bt.lens.sum())
            out = (np.ones((16, 1000), np.float32), np.ones(total, np.uint8), np.ones((16, 1000), np.uint64), np.ones((16, 1000), np.uint32), np.ones((16, 2), np.uint64))
        t0 = time.perf_counter(); res = bt.sampleback(1000, states, out=out); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); res = bt.sampleback(1000, states, out=res); t2 = time.perf_counter() - t0
        rows.append((t_f * 1e3, t1 * 1e3, t2 * 1e3))
        bt.close(); m.close()
    r = np.median(np.array(rows), axis=0)
    print(f"result arrays {'touched' if touched else 'fresh  '}{', prepared' if prepare else ''}: forward {r[0]:.2f} ms, first sampleback {r[1]:.2f} ms, second {r[2]:.2f} ms")
