#!/usr/bin/env python3
"""Round 6 (after viterbi_ck's spliced traceback and single-store boundaries): the planner's choice against viterbi_ck with 8 and
16 columns per lane and viterbi_lp with 4 / 3 / 2, on batches of related pairs.  Kernel ms, median of 3 after a warm-up.
usage: r6_plan_sweep.py [pairs x kb ...]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from tests import util
cases = [tuple(int(x) for x in c.split("x")) for c in sys.argv[1:]] or [(n, kb) for kb in (2, 4, 8, 16, 32) for n in (16, 32, 64, 128, 256, 512) if n * kb <= 4096]
table, consts = host.set_subst("mar-mg"), host.gap_consts()
rng = np.random.default_rng(7)
variants = [("plan", {}), ("ck16", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "16"}), ("ck8", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "8"}),
            ("ck4", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "4"}), ("lp4", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "4"}),
            ("lp3", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "3"}), ("lp2", {"COATI_HIP_VITERBI_BITS": "1", "COATI_HIP_STRIP_W": "2"})]
for n, kb in cases:
    base = []
    for _ in range(min(n, 16)):
        a = util.random_anc(rng, kb * 1000 // 3)
        base.append((a, util.mutate(rng, a, n_indel=2 * kb, mean_len=6)))
    enc = util.encode_pairs([base[p % len(base)] for p in range(n)])
    line, best = [f"{n:4d} x {kb:2d} kb:"], None
    res = {}
    for name, env in variants:
        for k in ("COATI_HIP_VITERBI_CK", "COATI_HIP_VITERBI_BITS", "COATI_HIP_STRIP_W"):
            os.environ.pop(k, None)
        os.environ.update(env)
        m = hip.Model(table, consts, 1)
        bt = hip.Batch(m, *hip.pack_pairs(enc))
        ts = []
        for _ in range(4):
            bt.viterbi_launch(); bt.sync(); ts.append(sum(bt.viterbi_timing()))
        res[name] = float(np.median(ts[1:]))
        bt.close(); m.close()
    best = min(v for k, v in res.items() if k != "plan")
    print(" ".join(line + [f"{k} {v:.3f}" for k, v in res.items()]) + f"   plan / best {res['plan'] / best:.2f}", flush=True)
