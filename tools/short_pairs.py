#!/usr/bin/env python3
"""GCUPS of the Viterbi path on synthetic pairs of a given codon count (default 100 codons = 300 nt):
how the strip shapes (4/8/16 columns per lane) keep short descendants efficient."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host
codons = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *host.synth_encoded(0, pairs, n_codons=codons))
ts = []
for r in range(8):
    batch.viterbi_launch(); batch.sync()
    f, w = batch.viterbi_timing()
    if r >= 2: ts.append(f + w)
t = float(np.median(ts))
print(f"{pairs} pairs of {codons} codons: {t:.3f} ms  {batch.cells/t/1e6:.1f} GCUPS  {pairs/t/1e3:.2f} M pairs/s")
