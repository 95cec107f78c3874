"""Forward fill of ONE pair, ancestors of 999 nt against descendants of 16 ... 1000 nt: quad strips against 1-column strips (ms by events)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from coati_amd import hip, host
from tests import util
rng = np.random.default_rng(11)
table, consts = host.set_subst("mar-mg"), host.gap_consts()
anc = util.random_anc(rng, 999)
full = util.mutate(rng, anc)
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for lb in (16, 32, 64, 128, 256, 512, 1000):
    enc = util.encode_pairs([(anc, full[:lb])] * npairs)
    line = f"lb {lb:5d} x {npairs} pairs:"
    for q in ("0", "1"):
        os.environ["COATI_HIP_FWD_QUAD"] = q
        os.environ["COATI_HIP_FWD_W"] = "1"
        model = hip.Model(table, consts, 1)
        batch = hip.Batch(model, *hip.pack_pairs(enc))
        ts = []
        for r in range(7):
            t0 = time.perf_counter()
            batch.forward_launch(); batch.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        line += f"   quad={q}: {np.median(ts[2:]):.3f} ms"
        batch.close(); model.close()
    print(line, flush=True)
