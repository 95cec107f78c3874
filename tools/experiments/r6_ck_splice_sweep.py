#!/usr/bin/env python3
"""Round 6: where does viterbi_ck's spliced traceback pay?  Batches of related pairs of several strips, kernel time with
COATI_HIP_CK_SPLICE = 0 / default / 1 / nobridge, results compared with level 0 (all pairs, scores and ops).
usage: r6_ck_splice_sweep.py [pairs x kb ...]   e.g. 64x8 256x8 1024x4"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from tests import util
cases = [tuple(int(x) for x in c.split("x")) for c in (sys.argv[1:] or ["16x16", "64x8", "64x16", "128x8", "256x8", "512x4", "1024x4", "256x16", "2048x4", "4096x2"])]
table, consts = host.set_subst("mar-mg"), host.gap_consts()
rng = np.random.default_rng(7)
for n, kb in cases:
    base = []
    for _ in range(min(n, 32)):
        a = util.random_anc(rng, kb * 1000 // 3)
        base.append((a, util.mutate(rng, a, n_indel=2 * kb, mean_len=6)))
    pairs = [base[p % len(base)] for p in range(n)]
    enc = util.encode_pairs(pairs)
    want = None
    line = [f"{n} x {kb} kb:"]
    for name, lv in (("off", "0"), ("default", None), ("on", "1"), ("nobridge", "nobridge")):
        if lv is None:
            os.environ.pop("COATI_HIP_CK_SPLICE", None)
        else:
            os.environ["COATI_HIP_CK_SPLICE"] = lv
        m = hip.Model(table, consts, 1)
        bt = hip.Batch(m, *hip.pack_pairs(enc))
        ts = []
        for _ in range(4):
            bt.viterbi_launch(); bt.sync(); ts.append(sum(bt.viterbi_timing()))
        sc, ops, off, ln = bt.viterbi_fetch()
        got = (sc.view(np.uint32).copy(), [ops[int(off[p]):int(off[p]) + int(ln[p])].copy() for p in range(n)])
        if want is None:
            want = got
        same = bool((got[0] == want[0]).all()) and all(np.array_equal(x, y) for x, y in zip(got[1], want[1]))
        line.append(f"{name} {np.median(ts[1:]):.3f} ms{'' if same else ' DIFFERENT'}")
        bt.close(); m.close()
    print("  ".join(line), flush=True)
