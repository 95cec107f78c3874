import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from coati_amd import hip, host, dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
comm = dist.Comm(dist.unique_id(), 1, 0, 0)
model = hip.Model(host.set_subst("mar-ecm"), host.gap_consts(), 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n)
res = None
for r in range(3):
    t0 = time.perf_counter()
    res = comm.viterbi_shard(model, a_cat, 0, a_off, b_cat, 0, b_off, reuse=res)
    print(f"job {r}: {time.perf_counter()-t0:.3f} s", flush=True)
