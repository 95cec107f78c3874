#!/usr/bin/env python3
"""Extremely ragged pairs (a few rows x tens of thousands of columns and the reverse, empty sides)
through the Viterbi, Forward and sampling paths against the oracle.  Complements the committed tests,
which cover ragged pairs up to a few thousand nt."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from oracle import pyoracle as orc
from tests import util

rng = np.random.default_rng(99)
table = host.set_subst("mar-mg")
consts = host.gap_consts()
SHAPES = {1: [(1, 50000), (1, 1), (10000, 4), (10000, 1), (0, 5000), (1000, 0), (2, 1025), (341, 65), (3, 16385), (5000, 5000)],
          3: [(1, 49998), (1, 3), (10000, 3), (0, 4998), (1000, 0), (2, 1026), (341, 66), (3, 16386), (2000, 6000)]}
for L in (1, 3):
    shapes = SHAPES[L]
    pairs = []
    for n_cod, lb in shapes:
        anc = util.random_anc(rng, n_cod) if n_cod else ""
        des = "".join(rng.choice(list("ACGT"), lb)) if lb else ""
        pairs.append((anc, des))
    enc = util.encode_pairs(pairs)
    model = hip.Model(table, consts, L)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.viterbi_launch()
    scores, ops, off, ln = batch.viterbi_fetch()
    batch.forward_launch()
    final = batch.forward_final()
    st = np.stack([host.rng_seed(["7", str(p)]) for p in range(len(enc))])
    lw, sops, soff, sln, sto = batch.sampleback(5, st, independent=False)
    for p, (a, b) in enumerate(enc):
        w_ops, w_sc = orc.viterbi(table, consts, L, a, b, lowmem=len(a) * len(b) > 4_000_000)
        got = ops[int(off[p]):int(off[p]) + int(ln[p])]
        assert len(got) == len(w_ops) and (got == w_ops).all(), ("viterbi ops", p, shapes[p])
        assert np.float32(scores[p]).view(np.uint32) == np.float32(w_sc).view(np.uint32), ("score", p, shapes[p])
        if len(a) * len(b) <= 30_000_000:
            M, D, I = orc.fill(orc.LOG, table, consts, L, a, b)
            want = np.array([M[-1, -1], D[-1, -1], I[-1, -1]], np.float32)
            assert util.same_bits(final[p], want), ("forward final", p, shapes[p], final[p], want)
            r = orc.rng_seed(["7", str(p)])
            for s in range(5):
                g = sops[int(soff[p, s]):int(soff[p, s]) + int(sln[p, s])]
                wo, wl = orc.sampleback_mdi(M, D, I, table, consts, L, a, b, r)
                assert len(wo) == len(g) and (wo == g).all(), ("sample", p, s, shapes[p])
                assert np.float32(lw[p, s]).view(np.uint32) == np.float32(wl).view(np.uint32), ("lw", p, s)
        print("ok gap_len", L, shapes[p], "len_a", len(a), "len_b", len(b), "columns", int(ln[p]))
    batch.close(); model.close()
print("ragged_check ok")
