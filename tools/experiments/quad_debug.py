"""forward_l1 quad strips against the 1-column strips: first mismatching cell per pair (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from coati_amd import hip, host
from tests import util
rng = np.random.default_rng(7)
table, consts = host.set_subst("mar-mg"), host.gap_consts()
pairs = [(util.random_anc(rng, n), None) for n in (3, 9, 15, 18, 30, 66, 120, 300)]
pairs = [(a, util.mutate(rng, a)) for a, _ in pairs]
enc = util.encode_pairs(pairs)
res = {}
for q in ("0", "1"):
    os.environ["COATI_HIP_FWD_QUAD"] = q
    os.environ["COATI_HIP_FWD_W"] = "1"
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.forward_launch()
    fin = batch.forward_final()
    mats = [batch.debug_forward_matrices(p) for p in range(len(enc))]
    res[q] = (fin, mats)
    batch.close(); model.close()
for p, (a, b) in enumerate(enc):
    f0, f1 = res["0"][0][p], res["1"][0][p]
    print(f"pair {p}: la {len(a)} lb {len(b)} final equal {np.array_equal(f0.view(np.uint32), f1.view(np.uint32))}  {f0} {f1}")
    for name, m0, m1 in zip("MDI", res["0"][1][p], res["1"][1][p]):
        bad = np.argwhere(m0.view(np.uint32) != m1.view(np.uint32))
        if len(bad):
            i, j = bad[0]
            print(f"   {name}: {len(bad)} of {m0.size} cells differ, first at row {i} col {j}: want {m0[i, j]} got {m1[i, j]}; rows with diffs {sorted(set(bad[:,0]))[:8]} cols {sorted(set(bad[:,1]))[:12]}")
