#!/usr/bin/env python3
"""BASELINE.json configs[3] (`coati sample -n 1000` on 1 kb pairs): time the Forward fill (log
semiring, M/D/I fp32 resident in HBM, 12 B/cell) and the stochastic tracebacks on the GPU, and the
generic (any gap_len) Viterbi kernel next to the hand-scheduled one.  Prints one JSON line.
Parity of these paths is what tests/test_gpu_sample.py and tests/test_gpu_generic.py check."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=6144)
ap.add_argument("--sample-pairs", type=int, default=16)
ap.add_argument("--samples", type=int, default=1000)
args = ap.parse_args()
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
out = {}


def timed(fn, sync, reps=5):
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        fn()
        sync()
        if r:
            ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


# ---- Forward fill throughput
batch = hip.Batch(model, *host.synth_encoded(0, args.pairs))
t = timed(batch.forward_launch, batch.sync)
out["forward_fill"] = {"pairs": args.pairs, "cells": batch.cells, "ms": t * 1e3, "gcups": batch.cells / t / 1e9,
                       "hbm_write_GBps_algorithmic_12B_per_cell": batch.cells * 12 / t / 1e9}
batch.close()

# ---- config 4: 16 pairs x 1000 samples
batch = hip.Batch(model, *host.synth_encoded(0, args.sample_pairs))
t_fwd = timed(batch.forward_launch, batch.sync)
states = np.array([host.rng_seed(["42"]) for _ in range(args.sample_pairs)], np.uint64)
for mode, indep in (("exact_stream", False), ("independent_streams", True)):
    ts, res = [], None
    for r in range(4):  # (the caller's result arrays are written again from the second call on)
        t0 = time.perf_counter()
        res = batch.sampleback(args.samples, states, independent=indep, out=res)
        ts.append(time.perf_counter() - t0)
    lw, ops, off, ln, _ = res
    out[f"sampleback_{mode}"] = {"pairs": args.sample_pairs, "samples_per_pair": args.samples, "ms": float(np.min(ts)) * 1e3, "first_call_ms": ts[0] * 1e3,
                                 "samples_per_s": args.sample_pairs * args.samples / float(np.min(ts)),
                                 "mean_columns": float(ln.mean()), "finite": bool(np.isfinite(lw).all())}
out["forward_fill_16_pairs_ms"] = t_fwd * 1e3
batch.close()

# ---- CPU port beside it (checker library, one thread): Forward with the reference's 11 matrices +
# sampleback, one pair, scaled to the GPU workload above
try:
    from oracle import pyoracle as orc  # baseline only
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 1)
    a, b = a_cat[:a_off[1]], b_cat[:b_off[1]]
    t0 = time.perf_counter()
    M, D, I, E = orc.fill(orc.LOG, table, consts, 1, a, b, edges=True)
    t_fwd_cpu = time.perf_counter() - t0
    mats = np.concatenate([np.stack([M, D, I]), E])
    rng = orc.rng_seed(["42"])
    t0 = time.perf_counter()
    n_cpu = 200
    for _ in range(n_cpu):
        orc.sampleback(mats, 1, rng)
    t_s_cpu = (time.perf_counter() - t0) / n_cpu
    out["cpu_port_1_thread"] = {"forward_ms_per_pair": t_fwd_cpu * 1e3, "forward_gcups": len(a) * len(b) / t_fwd_cpu / 1e9,
                                "sampleback_us_per_sample": t_s_cpu * 1e6,
                                "config4_estimate_ms": args.sample_pairs * (t_fwd_cpu + args.samples * t_s_cpu) * 1e3}
except Exception as exc:  # the baseline never fails the tool
    out["cpu_port_1_thread"] = {"error": repr(exc)}
print(json.dumps(out))
