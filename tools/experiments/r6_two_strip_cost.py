#!/usr/bin/env python3
"""What the headline batch's 333 two-strip pairs (descendants of 1 025-1 082 nt: a second strip of 1-58 columns) cost: the bench's
10 000 synthetic pairs against 10 000 pairs of the same generator whose descendants all fit ONE strip (<= 1 024 nt), and against the
two-strip pairs alone.  usage: r6_two_strip_cost.py [launches]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coati_amd import hip, host  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 10600)
enc = [(a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]) for p in range(10600)]
sets = {"bench set (first 10 000)": enc[:10000], "one strip each (first 10 000 with lb <= 1024)": [e for e in enc if len(e[1]) <= 1024][:10000],
        "the two-strip pairs of the bench set alone": [e for e in enc[:10000] if len(e[1]) > 1024]}
batches = {k: hip.Batch(model, *hip.pack_pairs(v)) for k, v in sets.items()}
times = {k: [] for k in sets}
for r in range(reps):
    for k, bt in batches.items():  # alternating
        bt.viterbi_launch(); bt.sync()
        if r >= 2:
            times[k].append(sum(bt.viterbi_timing()))
for k, bt in batches.items():
    t = float(np.median(times[k]))
    print(f"{k}: {len(sets[k])} pairs, {bt.cells / 1e9:.3f} G cells, {t:.3f} ms, {bt.cells / t / 1e6:.0f} GCUPS")
