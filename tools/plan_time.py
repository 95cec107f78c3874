import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from coati_amd import hip, host
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
enc = host.synth_encoded(0, 2000)
for r in range(3):
    t0 = time.perf_counter(); b = hip.Batch(model, *enc); t1 = time.perf_counter(); b.close()
    print("batch create 2000 pairs: %.3f ms" % ((t1 - t0) * 1e3), file=sys.stderr)
