"""Forward fill of n synthetic 1 kb pairs: quad strips against the planner's ordinary strips, GPU kept warm (ms, wall of launch + sync)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from coati_amd import hip, host
table, consts = host.set_subst("mar-mg"), host.gap_consts()
sizes = [int(x) for x in sys.argv[1:]] or [1, 4, 16, 32, 64, 128, 256]
os.environ.pop("COATI_HIP_FWD_QUAD", None)
model = hip.Model(table, consts, 1)
warm = hip.Batch(model, *host.synth_encoded(1, 3072))
def warm_up():
    for _ in range(2):
        warm.forward_launch(); warm.sync()
LB = int(os.environ.get("QUAD_SWEEP_LB", "0"))  # truncate the descendants (how a strip count scales)
for n in sizes:
    enc = host.synth_encoded(0, n)
    if LB:
        a_cat, a_off, b_cat, b_off = enc
        keep = np.concatenate([np.arange(int(b_off[p]), int(b_off[p]) + min(LB, int(b_off[p + 1] - b_off[p]))) for p in range(n)])
        lens = np.array([min(LB, int(b_off[p + 1] - b_off[p])) for p in range(n)], np.uint64)
        enc = (a_cat, a_off, b_cat[keep], np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64))
    line = f"{n:5d} pairs{(' lb ' + str(LB)) if LB else ''}:"
    for q in ("0", "1"):
        os.environ["COATI_HIP_FWD_QUAD"] = q
        hip.reload_env()
        batch = hip.Batch(model, *enc)
        warm_up()
        ts = []
        for r in range(9):
            t0 = time.perf_counter()
            batch.forward_launch(); batch.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        line += f"   quad={q}: {np.median(ts[1:]):.3f} ms (min {min(ts):.3f})"
        batch.close()
    print(line, flush=True)
