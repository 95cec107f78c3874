"""fill / traceback split of the golden long pairs (viterbi_timing), and the same with the traceback switched off where a switch exists"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from tests import util
for key in (sys.argv[1:] or ["160k"]):
    a, b, case, doc = util.load_long_pair(key)
    m = hip.Model(np.load(ROOT / "tests" / "golden" / doc["table"]), host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)
    bt = hip.Batch(m, *hip.pack_pairs([(a, b)]))
    for _ in range(4):
        bt.viterbi_launch(); bt.sync()
        f, w = bt.viterbi_timing()
        print(f"{key}: fill {f:.2f} ms  walk {w:.2f} ms", flush=True)
    sc, ops, off, ln = bt.viterbi_fetch()
    o = ops[int(off[0]):int(off[0]) + int(ln[0])]
    runs = 1 + int((o[1:] != o[:-1]).sum())
    edges = np.flatnonzero(o[1:] != o[:-1])
    lens = np.diff(np.concatenate([[-1], edges, [len(o) - 1]]))
    iters = int(np.ceil(lens / 64).sum())
    print("   alignment columns", len(o), " matches", int((o == 0).sum()), " runs", runs, " walker iterations (one per run and 64 moves)", iters,
          " run length p50/p90/max", int(np.median(lens)), int(np.percentile(lens, 90)), int(lens.max()))
    bt.close(); m.close()
