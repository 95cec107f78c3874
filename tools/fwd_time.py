#!/usr/bin/env python3
"""Time coati_hip_forward_launch on a resident synthetic batch (library chosen by COATI_HIP_LIB)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
gap_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), gap_len)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, pairs)
if gap_len > 1:  # descendant lengths must be multiples of the gap unit: trim
    lens = (np.diff(b_off) // gap_len * gap_len).astype(np.uint64)
    keep = np.concatenate([np.arange(int(b_off[p]), int(b_off[p]) + int(lens[p])) for p in range(pairs)])
    b_cat = b_cat[keep]
    b_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
ts = []
for r in range(6):
    t0 = time.perf_counter(); batch.forward_launch(); batch.sync(); ts.append(time.perf_counter() - t0)
t = float(np.median(ts[1:]))
print(f"pairs {pairs} gap_len {gap_len}: forward {t*1e3:.3f} ms  {batch.cells/t/1e9:.1f} GCUPS  {batch.cells*12/t/1e9:.0f} GB/s algorithmic")
