#!/usr/bin/env python3
"""A fixed workload for counter collection: `launches` Viterbi launches on one resident batch of
`pairs` synthetic 1 kb pairs (mar-mg).  Library: COATI_HIP_LIB or the in-tree build.
usage: fill_loop.py [pairs] [launches]        (under rocprofv3: -- python3 tools/fill_loop.py ...)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from coati_amd import hip, host  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, pairs)
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
t = []
for _ in range(launches):
    batch.viterbi_launch()
    batch.sync()
    t.append(batch.viterbi_timing()[0])
print(f"pairs {pairs} launches {launches} fill ms median {np.median(t):.3f} -> {batch.cells / np.median(t) / 1e6:.1f} GCUPS")
