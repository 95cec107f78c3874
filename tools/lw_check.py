#!/usr/bin/env python3
"""One-off: largest relative deviation of GPU-sampled log-weights from the oracle's evaluation of the
same paths on full 1 kb pairs (the committed tests use shorter pairs)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from oracle import pyoracle as orc
table, consts = host.set_subst("mar-mg"), host.gap_consts()
n_pairs, n_samples = 4, 200
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n_pairs)
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
batch.forward_launch()
final = batch.forward_final()
st = np.stack([host.rng_seed(["42", str(p)]) for p in range(n_pairs)])
lw, ops, off, ln, _ = batch.sampleback(n_samples, st, independent=False)
worst = 0.0
for p in range(n_pairs):
    a, b = a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]
    M, D, I = orc.fill(orc.LOG, table, consts, 1, a, b)
    fin = np.array([M[-1, -1], D[-1, -1], I[-1, -1]], np.float64)
    print("pair", p, "final cell rel err", np.max(np.abs(final[p] - fin) / np.maximum(1, np.abs(fin))))
    for s in range(n_samples):
        got = ops[int(off[p, s]):int(off[p, s]) + int(ln[p, s])]
        want = float(orc.path_logweight(M, D, I, table, consts, 1, a, b, got))
        worst = max(worst, abs(float(lw[p, s]) - want) / max(1.0, abs(want)))
print("worst relative log-weight deviation over", n_pairs * n_samples, "samples:", worst)
