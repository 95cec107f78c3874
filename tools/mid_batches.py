#!/usr/bin/env python3
"""Kernel time of 64 copies of the reference's benchmark pairs (bm_4k ... bm_32k) under the current environment switches
(planner experiments: COATI_HIP_STRIP_W, COATI_HIP_VITERBI_CK, COATI_HIP_LP_SPLICE ...), bit-exactness checked.
usage: mid_batches.py [copies] [key ...]"""
import sys, zlib
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from tests import util
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 64
keys = sys.argv[2:] or ["4k", "8k", "16k", "32k"]
doc = util.load_bench_pair("156")[3]
m = hip.Model(np.load(ROOT / "tests" / "golden" / doc["table"]), host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)
out = []
for k in keys:
    a, b, case, _ = util.load_bench_pair(k)
    bt = hip.Batch(m, *hip.pack_pairs([(a, b)] * copies))
    ts = []
    for _ in range(4):
        bt.viterbi_launch(); bt.sync(); ts.append(sum(bt.viterbi_timing()))
    sc, ops, off, ln = bt.viterbi_fetch()
    ok = all(int(np.float32(sc[p]).view(np.uint32)) == int(case["score_bits"], 16) and
             "%08x" % zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes()) == case["ops_crc32"] for p in range(copies))
    ms = float(np.median(ts[1:]))
    out.append(f"bm_{k}x{copies}: {ms:.2f} ms {len(a) * len(b) * copies / ms / 1e6:.0f} GCUPS {'ok' if ok else 'WRONG'}")
    bt.close(); m.trim()
print("; ".join(out))
