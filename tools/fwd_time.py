#!/usr/bin/env python3
"""Time coati_hip_forward_launch on a resident synthetic batch (library chosen by COATI_HIP_LIB)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *host.synth_encoded(0, pairs))
ts = []
for r in range(6):
    t0 = time.perf_counter(); batch.forward_launch(); batch.sync(); ts.append(time.perf_counter() - t0)
t = float(np.median(ts[1:]))
print(f"pairs {pairs}: forward {t*1e3:.3f} ms  {batch.cells/t/1e9:.1f} GCUPS  {batch.cells*12/t/1e9:.0f} GB/s algorithmic")
