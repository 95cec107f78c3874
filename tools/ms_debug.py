#!/usr/bin/env python3
"""Debug aid: decision bytes of multi-strip pairs through viterbi_ck (forced, 4- or 8-column strips) against the oracle, per strip."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["COATI_HIP_VITERBI_CK"] = "1"
os.environ["COATI_HIP_STRIP_W"] = sys.argv[1] if len(sys.argv) > 1 else "4"
from coati_amd import hip, host
from oracle import pyoracle as orc
from tests import util
rng = np.random.default_rng(5)
table, consts = util.random_table(rng), orc.gap_consts()
pairs = [(util.random_anc(rng, 100), "".join(rng.choice(list(util.NT), n))) for n in (700, 1200, 300, 1500)]
enc = util.encode_pairs(pairs)
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, *hip.pack_pairs(enc))
batch.viterbi_launch()
sc, ops, off, ln = batch.viterbi_fetch()
w = int(os.environ["COATI_HIP_STRIP_W"])
for p, (a, b) in enumerate(enc):
    want_ops, want_score = orc.viterbi(table, consts, 1, a, b)
    got = ops[int(off[p]):int(off[p]) + int(ln[p])]
    M, D, I = orc.fill(orc.TROPICAL, table, consts, 1, a, b)
    want = orc.tb_flags(M, D, I, consts)[1:, 1:].copy()
    gf = batch.debug_flags(p)
    want[-1, -1] = gf[-1, -1]
    bad = gf != want
    per_strip = [int(bad[:, s:s + 64 * w].sum()) for s in range(0, len(b), 64 * w)]
    rows = np.argwhere(bad.any(axis=1)).ravel()
    print(p, len(a), len(b), "score ok", np.float32(sc[p]).view(np.uint32) == np.float32(want_score).view(np.uint32), "ops ok", len(got) == len(want_ops) and (got == want_ops).all(),
          "bad cells per strip", per_strip, "bad rows", (rows[:5], rows[-5:]) if len(rows) else None)
    if bad.any():
        rr, cc = np.nonzero(bad)
        strip = cc // (64 * w)
        s0 = int(np.bincount(strip).argmax())
        m = strip == s0
        t = (cc[m] - s0 * 64 * w) // w
        c = (rr[m] + t) // 16
        tiles = sorted(set(zip(c.tolist(), t.tolist())))
        print("   strip", s0, "bad tiles", len(tiles), "bands", min(x[0] for x in tiles), max(x[0] for x in tiles), "lanes", min(x[1] for x in tiles), max(x[1] for x in tiles))
        byband = {}
        for cb, tl in tiles:
            byband.setdefault(cb, []).append(tl)
        for cb in sorted(byband)[:24]:
            print("      band", cb, "lanes", byband[cb][0], "..", byband[cb][-1], "n", len(byband[cb]))
