#!/usr/bin/env python3
"""Per-pair cost of a two-strip pair in a throughput-bound launch: the bench set's 333 two-strip pairs (descendants of 1 025-1 082
nt) x 30 against its 333 longest ONE-strip pairs x 30, one launch each, alternating.  usage: r6_two_strip_throughput.py [launches]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coati_amd import hip, host  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 10000)
enc = [(a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]) for p in range(10000)]
two = [e for e in enc if len(e[1]) > 1024]
one = sorted((e for e in enc if len(e[1]) <= 1024), key=lambda e: -len(e[1]))[:len(two)]
sets = {"two-strip pairs x 30": two * 30, "longest one-strip pairs x 30": one * 30}
batches = {k: hip.Batch(model, *hip.pack_pairs(v)) for k, v in sets.items()}
times = {k: [] for k in sets}
for r in range(reps):
    for k, bt in batches.items():
        bt.viterbi_launch(); bt.sync()
        if r >= 2:
            times[k].append(sum(bt.viterbi_timing()))
for k, bt in batches.items():
    t = float(np.median(times[k]))
    print(f"{k}: {len(sets[k])} pairs, {bt.cells / 1e9:.3f} G cells, {t:.3f} ms, {t / len(sets[k]) * 1e3:.4f} us per pair, {bt.cells / t / 1e6:.0f} GCUPS")
