"""Where does the local-results job (coati_hip_dist_viterbi_shard_local, world 1) spend its time beside the streamed call?
Wall time of the job against the library's own trace line of the streamed call inside it (COATI_HIP_STREAM_HELPERS=63).
usage: python tools/local_job_probe.py [pairs]"""
import os, sys, time
os.environ.setdefault("COATI_HIP_STREAM_HELPERS", "63")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coati_amd import hip, host, dist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
comm = dist.Comm(dist.unique_id(), 1, 0, device=0)
model = hip.Model(host.set_subst("mar-ecm"), host.gap_consts(), 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n)
loc = None
for r in range(4):
    t0 = time.perf_counter()
    loc, summary = comm.viterbi_shard_local(model, a_cat, 0, a_off, b_cat, 0, b_off, root=0, reuse=loc)
    print(f"job {r}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for r in range(3):
    t0 = time.perf_counter()
    out = model.viterbi(a_cat, a_off, b_cat, b_off, out=loc, pinned=True)
    print(f"plain call {r}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
