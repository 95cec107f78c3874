"""Forward of ONE long pair: quad strips against 1-column strips (ms), and the final cells' bits"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from coati_amd import hip, host
from tests import util
rng = np.random.default_rng(5)
table, consts = host.set_subst("mar-mg"), host.gap_consts()
for n in (4000, 12000, 24000):
    anc = util.random_anc(rng, n // 3 * 3)
    enc = util.encode_pairs([(anc, util.mutate(rng, anc))])
    fins = {}
    line = f"la {len(enc[0][0])} lb {len(enc[0][1])}:"
    for q in ("0", "1"):
        os.environ["COATI_HIP_FWD_QUAD"] = q
        os.environ["COATI_HIP_FWD_W"] = "1"
        model = hip.Model(table, consts, 1)
        batch = hip.Batch(model, *hip.pack_pairs(enc))
        ts = []
        for r in range(4):
            t0 = time.perf_counter()
            batch.forward_launch(); batch.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        fins[q] = batch.forward_final()[0].copy()
        line += f"   quad={q}: {min(ts):.2f} ms"
        batch.close(); model.close()
    print(line, " finals equal:", np.array_equal(fins["0"].view(np.uint32), fins["1"].view(np.uint32)), flush=True)
