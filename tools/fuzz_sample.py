#!/usr/bin/env python3
"""Randomised campaign for the Forward + sampleback path: random batches (gap_len 1..3, several
tables, random gap parameters).  Every GPU sample must be a valid path and the final RNG state must
equal state * MULT^(total draws).  Log-weights are compared with the oracle's evaluation of the same
path: with MODEL tables (mar-mg / mar-ecm at random branch lengths and omega) the 1e-5 relative bound
is asserted; with uniform random tables (scores -8..2: near-ties everywhere at magnitudes of several
hundred, where one fp32 ulp of M/D/I is 3e-5) the deviations are only recorded -- there the bound
depends on glibc's and the GPU's log1p(exp()) agreeing in the last bit.  In the default (bit-exact)
build every exact-stream sample is also compared with the oracle's own sample: same ops, same
log-weight bits.
usage: fuzz_sample.py [seconds] [seed]"""
import os
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from oracle import pyoracle as orc
from tests import util

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed)
MULT = 0xda942042e4dd58b5
t_end = time.time() + budget
exact = util.forward_exact()
rounds = samples = samples_random = over_random = identical = 0
worst = worst_random = 0.0
while time.time() < t_end:
    L = int(rng.choice([1, 1, 2, 3]))
    n_tables = int(rng.integers(1, 3))
    model_tables = bool(rng.random() < 0.6)
    if model_tables:
        tables = np.stack([host.set_subst(str(rng.choice(["mar-mg", "mar-ecm"])), br_len=float(rng.choice([0.0133, 0.1, 0.5, 1.5])),
                                          omega=float(rng.choice([0.2, 1.0]))) for _ in range(n_tables)])
    else:
        tables = np.stack([util.random_table(rng) for _ in range(n_tables)])
    consts = orc.gap_consts(float(rng.choice([0.001, 0.02])), float(rng.choice([5 / 6, 0.6])))
    pairs = util.make_pairs(rng, int(rng.integers(1, 12)), 1, int(rng.choice([15, 60, 130])), L=L, amb=0.02)
    enc = util.encode_pairs(pairs)
    tix = rng.integers(0, n_tables, len(enc)).astype(np.uint32)
    n_s = int(rng.choice([1, 3, 7, 40, 150]))
    model = hip.Model(tables, consts, L)
    batch = hip.Batch(model, *hip.pack_pairs(enc), table_index=tix)
    batch.forward_launch()
    st = np.stack([host.rng_seed([str(seed), str(rounds), str(p)]) for p in range(len(enc))])
    indep = bool(rng.random() < 0.3)
    # the exact-stream sampler's own switches: candidate budget (1 024: dozens of rounds, windows cut by the share) and where
    # the rounds are planned and resolved (device: default; host: round 3's loop)
    for k in ("COATI_HIP_SPEC_CANDS", "COATI_HIP_SPEC_HOST_ROUNDS", "COATI_HIP_SPEC_Z", "COATI_HIP_SAMPLE_BAND"):
        os.environ.pop(k, None)
    if rng.random() < 0.35:  # a step table of a few diagonals: the walkers compute the entries of every cell off them
        os.environ["COATI_HIP_SAMPLE_BAND"] = str(int(rng.choice([1, 2, 5, 12])))
    pick = rng.random()
    if pick < 0.3:
        os.environ["COATI_HIP_SPEC_CANDS"] = str(int(rng.choice([1024, 4096])))
        if rng.random() < 0.5:
            os.environ["COATI_HIP_SPEC_Z"] = str(float(rng.choice([0.5, 1.0, 4.0])))
    elif pick < 0.45:
        os.environ["COATI_HIP_SPEC_HOST_ROUNDS"] = "1"
    lw, ops, off, ln, so = batch.sampleback(n_s, st, independent=indep)
    for p, (a, b) in enumerate(enc):
        M, D, I = orc.fill(orc.LOG, tables[tix[p]], consts, L, a, b)
        draws = 0
        ref_rng = orc.rng_seed([str(seed), str(rounds), str(p)]) if (exact and not indep) else None
        for s in range(n_s):
            got = ops[int(off[p, s]):int(off[p, s]) + int(ln[p, s])]
            if ref_rng is not None:  # default build: the oracle's own sample, draw for draw
                w_ops, w_lw = orc.sampleback_mdi(M, D, I, tables[tix[p]], consts, L, a, b, ref_rng)
                assert len(w_ops) == len(got) and (w_ops == got).all(), ("sample differs", seed, rounds, p, s)
                assert np.float32(lw[p, s]).view(np.uint32) == np.float32(w_lw).view(np.uint32), ("lw bits", seed, rounds, p, s)
                identical += 1
            nm, nd, ni = int((got == 0).sum()), int((got == 1).sum()), int((got == 2).sum())
            assert nm + nd == len(a) and nm + ni == len(b), ("invalid path", seed, rounds, p, s)
            want = float(orc.path_logweight(M, D, I, tables[tix[p]], consts, L, a, b, got))
            dev = abs(float(lw[p, s]) - want) / max(1.0, abs(want))
            if model_tables:
                worst = max(worst, dev)
                assert dev <= 1e-5, ("log-weight", seed, rounds, p, s, float(lw[p, s]), want)
            else:
                worst_random = max(worst_random, dev)
                over_random += dev > 1e-5
                samples_random += 1
            draws += 1 + nm + (nd + ni) // L
            samples += 1
        if not indep:
            s0 = (int(st[p, 1]) << 64) | int(st[p, 0])
            s1 = (s0 * pow(MULT, draws, 1 << 128)) % (1 << 128)
            assert (int(so[p, 0]), int(so[p, 1])) == (s1 & ((1 << 64) - 1), s1 >> 64), ("rng state", seed, rounds, p)
    batch.close(); model.close()
    rounds += 1
print(f"fuzz_sample ok: {rounds} batches, {samples} samples, seed {seed}; model tables: worst relative log-weight deviation "
      f"{worst:.2e}; uniform random tables: {samples_random} samples, worst {worst_random:.2e}, {over_random} above 1e-5; {identical} exact-stream samples "
      f"identical to the oracle's (ops and log-weight bits)" + ("" if exact else " [fast build: not compared]"))
