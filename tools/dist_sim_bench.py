#!/usr/bin/env python3
"""Host-side cost of the sharded job's result exchange, measured WITHOUT GPUs: coati_hip_dist_simulate runs every rank of
the gather-all job (and _simulate_local of the local-results job) as a thread of this process, over host memory and an
in-process transport -- the same job loop, chunk plan, count exchange, landing zone, unpack / offset rebase / placement
code as the RCCL entry points (csrc/dist.hip), with memcpy where those call ncclSend/ncclRecv/hipMemcpyAsync.  The
ranks' "kernels" are copies out of given per-pair results, so what is timed is the plan + exchange + placement work the
root (and every rank) does per round for BASELINE configs[4]'s size: DESIGN.md section 6 budgets it as "per-round host
work".  No device is touched.
usage: python3 tools/dist_sim_bench.py [pairs] [out.json]"""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import dist, host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
la, lb = host.synth_lengths(0, n)
a_off = np.concatenate([[0], np.cumsum(la)]).astype(np.uint64)
b_off = np.concatenate([[0], np.cumsum(lb)]).astype(np.uint64)
width = (la + lb).astype(np.uint64)
total = int(width.sum())
rng = np.random.default_rng(1)
scores = rng.standard_normal(n).astype(np.float32)
ops_len = (np.maximum(la, lb) + rng.integers(0, 8, n)).astype(np.uint32)  # (a realistic alignment is about max(la, lb) columns)
ops = np.empty(total, np.uint8)
ops[:] = 1  # (first touch here, not under the clock)
rec = {"pairs": n, "op_slot_bytes": total, "op_bytes": int(ops_len.sum()), "host_cores": os.cpu_count(),
       "what": "wall time of coati_hip_dist_simulate / _simulate_local (all ranks as threads, host memory, memcpy transport; median of 3 "
               "after one warm-up), and where the ROOT's thread spent the last run (coati_hip_dist_debug_job_times).  In this "
               "simulation the ranks' kernels, ncclSend/ncclRecv and the root's D2H are host memcpy (waiting_for_own_chunks, "
               "send_receive_group, own_copy_out, most of unpack_placement_rebase; landing_zone_reserve is a vector fill here, a cached "
               "HBM block there): what carries over to hardware is count_exchanges + loop_logic_plans_validation + the per-block "
               "bookkeeping inside unpack"}
from coati_amd import hip  # noqa: E402  (only its pointer helper; no device call)

lib = dist.load()
o_scores, o_ops = np.zeros(n, np.float32), np.zeros(total, np.uint8)  # (outputs exist and are touched before the clock starts)
o_off, o_len = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
all_scores, all_len = np.zeros(n, np.float32), np.zeros(n, np.uint32)


def gather_all(world):
    rc = lib.coati_hip_dist_simulate(world, 0, n, hip._ptr(a_off), hip._ptr(b_off), 0, hip._ptr(scores), hip._ptr(ops), hip._ptr(ops_len),
                                     hip._ptr(o_scores), hip._ptr(o_ops), total, hip._ptr(o_off), hip._ptr(o_len))
    assert rc == 0, dist.load().coati_hip_dist_last_error()
    assert (o_scores.view(np.uint32) == scores.view(np.uint32)).all() and (o_len == ops_len).all()


def local(world):
    rc = lib.coati_hip_dist_simulate_local(world, 0, n, hip._ptr(a_off), hip._ptr(b_off), 0, 1, hip._ptr(scores), hip._ptr(ops), hip._ptr(ops_len),
                                           hip._ptr(o_scores), hip._ptr(o_ops), total, hip._ptr(o_off), hip._ptr(o_len), hip._ptr(all_scores),
                                           hip._ptr(all_len))
    assert rc == 0, dist.load().coati_hip_dist_last_error()
    assert (all_len == ops_len).all()


for world in (1, 2, 4, 8):
    for name, fn in (("gather_all", gather_all), ("local", local)):
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            fn(world)
            ts.append(time.perf_counter() - t0)
        _, rounds = dist.chunk_plan(a_off, b_off, world)
        t6 = (C.c_double * 8)()
        assert lib.coati_hip_dist_debug_job_times(t6) == 0  # (of the last run: the root's thread)
        loop_s, own_wait, exchange, transfer, unpack = (float(t6[i]) for i in range(5))
        rec[f"world{world}_{name}"] = {"s": round(float(np.median(ts[1:])), 4), "rounds": rounds,
                                       "root_thread_s": {"job_loop": round(loop_s, 4), "waiting_for_own_chunks": round(own_wait, 4),
                                                         "count_exchanges": round(exchange, 4), "send_receive_group": round(transfer, 4),
                                                         "unpack_placement_rebase": round(unpack, 4),
                                                         "landing_zone_reserve": round(float(t6[6]), 4), "own_copy_out_local_form": round(float(t6[7]), 4),
                                                         "loop_logic_plans_validation": round(loop_s - own_wait - exchange - transfer - unpack - float(t6[6]) - float(t6[7]), 4)}}
        print(world, name, rec[f"world{world}_{name}"], flush=True)
text = json.dumps(rec, indent=1)
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(text + "\n")
print(text)
