#!/usr/bin/env python3
"""BASELINE.json configs[2]: one long synthetic pair (default 160 002 nt, 53 334 codons)
through the product C ABI, gap_len 1.  Its 157 strips run on different wavefronts,
pipelined through the strip-boundary arrays (viterbi_l1.hip).

Checks (size independent; the full oracle would need 3 x 102 GB of fp32 matrices):
  * the ops consume both sequences exactly (count of M/D/I columns);
  * oracle.path_score re-derives the Viterbi value ALONG the returned path with the
    reference's expressions: for the optimal path it must equal the GPU score bit for bit;
  * optionally (--lowmem) the O(cols)-memory oracle recomputes score and path on the CPU
    (minutes for 160 kb).
Prints one JSON line.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (checker only)

ap = argparse.ArgumentParser()
ap.add_argument("--codons", type=int, default=53334)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--lowmem", action="store_true")
ap.add_argument("--gap-len", type=int, default=1)
args = ap.parse_args()

table, consts = host.set_subst("mar-mg"), host.gap_consts()
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 1, n_codons=args.codons)
L = args.gap_len
if L > 1:  # lengths must be multiples of the gap unit
    b_off = np.array([0, int(b_off[1]) // L * L], np.uint64)
    b_cat = b_cat[:int(b_off[1])]
la, lb = int(a_off[1]), int(b_off[1])
model = hip.Model(table, consts, L)
t0 = time.time()
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
t_create = time.time() - t0
ms, fw = [], []
for r in range(args.reps + 1):
    batch.viterbi_launch()
    batch.sync()
    f, w = batch.viterbi_timing()
    if r:
        ms.append(f + w)
        fw.append((f, w))
scores, ops, off, ln = batch.viterbi_fetch()
path = ops[int(off[0]):int(off[0]) + int(ln[0])]
n_m, n_d, n_i = (int((path == k).sum()) for k in (0, 1, 2))
consumed_ok = (n_m + n_d == la) and (n_m + n_i == lb)
ps = orc.path_score(table, consts, L, a_cat[:la], b_cat[:lb], path)
bit_equal = bool(np.float32(ps).view(np.uint32) == np.float32(scores[0]).view(np.uint32))
out = {
    "workload": f"1 pair {la} x {lb} nt, gap_len {L} (synthetic, configs[2])", "cells": la * lb,
    "device_bytes": batch.device_bytes, "batch_create_s": round(t_create, 3), "ms_median": float(np.median(ms)),
    "ms_min": float(np.min(ms)), "fill_ms": float(np.median([x[0] for x in fw])), "traceback_ms": float(np.median([x[1] for x in fw])), "gcups": la * lb / float(np.median(ms)) / 1e6, "score": float(scores[0]),
    "columns": int(ln[0]), "ops_consume_both_sequences": consumed_ok, "path_score": float(ps),
    "path_score_bit_equal": bit_equal,
}
if args.lowmem:
    t0 = time.time()
    o_ops, o_sc = orc.viterbi(table, consts, L, a_cat[:la], b_cat[:lb], lowmem=True)
    out["oracle_lowmem_s"] = round(time.time() - t0, 1)
    out["oracle_score_bit_equal"] = bool(np.float32(o_sc).view(np.uint32) == np.float32(scores[0]).view(np.uint32))
    out["oracle_path_equal"] = bool(len(o_ops) == len(path) and np.array_equal(o_ops, path))
print(json.dumps(out))
sys.exit(0 if consumed_ok and bit_equal else 1)
