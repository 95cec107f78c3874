#!/usr/bin/env python3
"""A/B harness: time the Viterbi kernels of several builds of libcoati_hip.so on the
same resident batch, interleaved rounds in ONE process per library (ctypes cannot
load two copies of one soname cleanly, so each variant runs in a subprocess that
repeats the measurement `rounds` times; medians are compared).

usage: ab_fill.py [--pairs N] [--rounds R] lib1.so lib2.so ...
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip, host
pairs, rounds, gap_len = %d, %d, %d
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, gap_len)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, pairs)
if gap_len > 1:  # descendant lengths must be multiples of the gap unit: trim
    lens = (np.diff(b_off) // gap_len * gap_len).astype(np.uint64)
    keep = np.concatenate([np.arange(int(b_off[p]), int(b_off[p]) + int(lens[p])) for p in range(pairs)])
    b_cat = b_cat[keep]
    b_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
fills, walks = [], []
for r in range(rounds + 2):
    batch.viterbi_launch(); batch.sync()
    f, w = batch.viterbi_timing()
    if r >= 2: fills.append(f); walks.append(w)
scores, ops, off, ln = batch.viterbi_fetch()
print(json.dumps({"fill_med": float(np.median(fills)), "fill_min": float(np.min(fills)), "walk_med": float(np.median(walks)),
                  "cells": batch.cells, "checksum": float(np.float64(scores).sum()), "ops": int(ln.sum())}))
'''

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=10000)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--gap-len", type=int, default=1)
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
for lib in args.libs:
    env = dict(os.environ, COATI_HIP_LIB=str(Path(lib).resolve()))
    out = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), args.pairs, args.rounds, args.gap_len)], env=env,
                         capture_output=True, text=True)
    if out.returncode != 0:
        print(f"{lib}: FAILED\n{out.stderr[-2000:]}")
        continue
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(f"{Path(lib).name:32s} fill med {r['fill_med']:.3f} ms (min {r['fill_min']:.3f})  {r['cells']/r['fill_med']/1e6:8.1f} GCUPS  "
          f"walk {r['walk_med']:.3f} ms  checksum {r['checksum']:.6f} ops {r['ops']}")
