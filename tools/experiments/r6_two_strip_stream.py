#!/usr/bin/env python3
"""The streamed call (page-locked arrays) on the bench's 10 000 pairs against 10 000 pairs of the same generator that all fit one
strip: what the 333 two-strip pairs cost THERE.  usage: r6_two_strip_stream.py [calls]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coati_amd import hip, host  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
os.environ["COATI_HIP_PIPE"] = "stream"
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 10600)
enc = [(a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]) for p in range(10600)]
sets = {"bench set": enc[:10000], "one strip each": [e for e in enc if len(e[1]) <= 1024][:10000]}
packed = {k: hip.pack_pairs(v) for k, v in sets.items()}
pinned = {k: (hip.pinned_copy(v[0]), v[1], hip.pinned_copy(v[2]), v[3]) for k, v in packed.items()}
out = {k: None for k in sets}
times = {k: [] for k in sets}
for r in range(reps):
    for k, (pa, ao, pb, bo) in pinned.items():
        t0 = time.perf_counter()
        out[k] = model.viterbi(pa, ao, pb, bo, out=out[k], pinned=True)
        if r >= 4:
            times[k].append((time.perf_counter() - t0) * 1e3)
for k in sets:
    print(f"{k}: median {np.median(times[k]):.3f} ms, best {min(times[k]):.3f} ms")
