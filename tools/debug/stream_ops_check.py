#!/usr/bin/env python3
"""Which pairs of a streamed call (page-locked arrays) differ from the resident batch's ops, and where.
usage: stream_ops_check.py [n_synth_pairs]   (COATI_HIP_STREAM_HELPERS etc. from the environment)"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from coati_amd import hip, host  # noqa: E402
import util  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
table, consts = host.set_subst("mar-mg"), host.gap_consts()
rng = np.random.default_rng(77)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n)
enc = [(a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]) for p in range(n)]
extra = util.encode_pairs(util.make_pairs(rng, 40, 0, 60, L=1, amb=0.05))
for la, lb in [(900, 2300), (1200, 1025), (300, 3100), (2502, 700), (3, 1500), (1500, 1)]:
    extra.append((rng.integers(0, 183, la).astype(np.uint8), rng.integers(0, 4, lb).astype(np.uint8)))
order = rng.permutation(len(enc) + len(extra))
enc = [(enc + extra)[i] for i in order]
a_cat, a_off, b_cat, b_off = hip.pack_pairs(enc)
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
batch.viterbi_launch()
want = batch.viterbi_fetch()
batch.close()
os.environ["COATI_HIP_PIPE"] = "stream"
pa, pb = hip.pinned_copy(a_cat), hip.pinned_copy(b_cat)
for rep in range(3):
    sc, ops, off, ln = model.viterbi(pa, a_off, pb, b_off, pinned=True)
    bad = 0
    for p in range(len(enc)):
        s, l = int(want[2][p]), int(want[3][p])
        w, g = want[1][s:s + l], ops[int(off[p]):int(off[p]) + int(ln[p])]
        if int(off[p]) != s or int(ln[p]) != l or not (w == g).all():
            bad += 1
            if bad <= 12:
                d = np.nonzero(w != g)[0] if len(w) == len(g) else []
                print(f"rep {rep} pair {p}: la {len(enc[p][0])} lb {len(enc[p][1])} start {s} len {l}; got start {int(off[p])} len {int(ln[p])}; "
                      f"{len(d)} bytes differ, first at {d[:4]}, got {g[d[:4]] if len(d) else ''} want {w[d[:4]] if len(d) else ''}")
    print(f"rep {rep}: {bad} of {len(enc)} pairs differ")
