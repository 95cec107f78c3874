#!/usr/bin/env python3
"""Kernel time of the golden long pairs (tests/golden/long_pairs.npz) under the current environment switches, with the
bit-exactness check against the fixture.  usage: long_golden_time.py [key ...]   (default: 160k)"""
import sys, zlib
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from tests import util
for key in (sys.argv[1:] or ["160k"]):
    a, b, case, doc = util.load_long_pair(key)
    m = hip.Model(np.load(ROOT / "tests" / "golden" / doc["table"]), host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)
    bt = hip.Batch(m, *hip.pack_pairs([(a, b)]))
    ts = []
    for _ in range(4):
        bt.viterbi_launch(); bt.sync(); ts.append(sum(bt.viterbi_timing()))
    sc, ops, off, ln = bt.viterbi_fetch()
    got = ops[int(off[0]):int(off[0]) + int(ln[0])]
    ok = int(np.float32(sc[0]).view(np.uint32)) == int(case["score_bits"], 16) and "%08x" % zlib.crc32(got.tobytes()) == case["ops_crc32"]
    print(f"{key}: {np.median(ts[1:]):.2f} ms  bit-exact {ok}  device bytes {bt.device_bytes / 1e9:.2f} GB", flush=True)
    bt.close(); m.close()
