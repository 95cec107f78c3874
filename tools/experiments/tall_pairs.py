#!/usr/bin/env python3
"""What would chaining pairs in the lane pipeline be worth?  A wavefront's 64 lanes start (and finish) a pair one step
apart: 63 of a 1 kb pair's 1 065 steps are half idle.  Upper bound of what removing that gains: the same cells as pairs
that are TWICE / FOUR TIMES as tall (ancestors of 2 / 4 synthetic pairs stacked, one descendant): the skew per cell halves /
quarters, everything else (cells per step, checkpoint bytes per cell) stays.
usage: python3 tools/experiments/tall_pairs.py [pairs]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n)
for stack in (1, 2, 4):
    m = n // stack
    # ancestor of stacked pair q = ancestors of pairs stack*q .. stack*q+stack-1; descendant of pair stack*q
    A_off = a_off[::stack][:m + 1].copy()
    keep = np.concatenate([np.arange(int(b_off[stack * q]), int(b_off[stack * q + 1])) for q in range(m)])
    B_cat = b_cat[keep]
    B_off = np.concatenate([[0], np.cumsum([int(b_off[stack * q + 1] - b_off[stack * q]) for q in range(m)])]).astype(np.uint64)
    batch = hip.Batch(model, a_cat[:int(A_off[-1])], A_off, B_cat, B_off)
    for fill_only in (0, 1):
        import os
        os.environ["COATI_HIP_CK_DEBUG"] = str(fill_only)
        hip.load().coati_hip_debug_reload_env()
        t = []
        for _ in range(7):
            batch.viterbi_launch()
            batch.sync()
            t.append(batch.viterbi_timing()[0])
        ms = float(np.median(t[2:]))
        print(f"stack {stack}: {m} pairs of {int(A_off[1])} x ~1000, {batch.cells/1e9:.2f} Gcells, {'fill only' if fill_only else 'fill + traceback'}: {ms:.3f} ms  {batch.cells / ms / 1e6:.0f} GCUPS", flush=True)
    batch.close()
