#!/usr/bin/env python3
"""Where the PCIe-inclusive time of a one-shot call goes: batch_create (allocations + H2D), launch +
sync, fetch (D2H), destroy."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
data = host.synth_encoded(0, pairs)
for rep in range(3):
    t0 = time.perf_counter(); batch = hip.Batch(model, *data)
    t1 = time.perf_counter(); batch.viterbi_launch(); batch.sync()
    t2 = time.perf_counter(); res = batch.viterbi_fetch()
    t3 = time.perf_counter(); batch.close()
    t4 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.2f} ms  kernel {1e3*(t2-t1):.2f}  fetch {1e3*(t3-t2):.2f}  destroy {1e3*(t4-t3):.2f}  total {1e3*(t4-t0):.2f}  ({batch.device_bytes/1e9:.2f} GB)")
t0 = time.perf_counter(); model.viterbi(*data); print(f"one-shot call {1e3*(time.perf_counter()-t0):.2f} ms")
