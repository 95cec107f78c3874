import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from coati_amd import hip, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
a = host.synth_encoded(0, n)
pa, pb = hip.pinned_copy(a[0]), hip.pinned_copy(a[2])
out = None
for r in range(4):
    if r == 3: os.environ["COATI_HIP_PIPE_TIMING"] = "1"
    t0 = time.perf_counter(); out = model.viterbi(pa, a[1], pb, a[3], out=out, pinned=True); dt = time.perf_counter() - t0
    print("call", r, "%.2f ms" % (dt * 1e3), file=sys.stderr)
