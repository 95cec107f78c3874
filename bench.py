#!/usr/bin/env python3
"""Headline benchmark: GCUPS (DP cell updates/s) and pairs/s of the marginal
Viterbi path (fill + traceback) on synthetic 1 kb x 1 kb pairs, mar-mg94
(BASELINE.json configs[1]), inputs resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (fill kernel + traceback walker, and for
N > 1 the gather of all results to rank 0 over RCCL) over one batch of
`--pairs` pairs PER GPU (weak scaling: the reference aligns one pair per
process, pairs are independent, shards need no data-path collective).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Secondary (and for this recurrence the binding) ceiling, DESIGN.md §4: the gap_len-1 cell is
# 25 VALU instructions; non-packed fp32 VALU issues 32 lanes/clk/SIMD (measured, tools/ubench):
# 1024 SIMDs x 32 x 2.4 GHz / 25 = 3.1 TCUPS.
VALU_PEAK_GCUPS = 1024 * 32 * 2.4 / 25.0
ALGO_BYTES_PER_CELL = 1.0      # 1 B/cell packed traceback written ...
# ... + (len_a + len_b) B of sequence read per pair (added per pair below)


def cpu_baseline(table, consts, a_cat, a_off, b_cat, b_off, budget_s=12.0):
    """The CPU side of the comparison on this box's host cores, bounded to ~budget_s of work per leg:
    the UNMODIFIED reference engine where oracle/_ref travelled with the snapshot (kind "reference":
    viterbi_mem + traceback_viterbi of /root/reference, compiled in the build container), else the
    bit-identical port (kind "port").  One pair per thread."""
    import time as _t
    from concurrent.futures import ThreadPoolExecutor

    from coati_amd import host as _host
    from oracle import pyoracle as orc  # checker/baseline only -- never on the product path

    cores = os.cpu_count() or 1
    n_all = len(a_off) - 1
    la = np.diff(a_off).astype(np.float64)
    lb = np.diff(b_off).astype(np.float64)
    # ---- the port: calibrate on a few pairs (1 thread), then ~budget_s on all cores
    n_cal = 4
    t_cal, _ = orc.viterbi_batch_timed(table, consts, 1, a_cat, a_off[:n_cal + 1], b_cat, b_off[:n_cal + 1], 1)
    per_pair = max(t_cal / n_cal, 1e-4)
    n = int(min(n_all, max(cores, budget_s / per_pair * cores * 0.6)))
    secs, _ = orc.viterbi_batch_timed(table, consts, 1, a_cat, a_off[:n + 1], b_cat, b_off[:n + 1], cores)
    port = {"gcups": float((la[:n] * lb[:n]).sum() / secs / 1e9), "pairs_per_s": n / secs, "pairs": n, "seconds": secs,
            "single_thread_gcups": float((la[:n_cal] * lb[:n_cal]).sum() / t_cal / 1e9)}
    out = {"value": port["gcups"], "unit": "GCUPS", "cores": cores, "kind": "port", "pairs_per_s": port["pairs_per_s"],
           "sample": f"first {n} pairs of the same synthetic set, oracle viterbi_mem+traceback (3 fp32 matrices incl. "
                     f"fill), {cores} threads, {secs:.1f} s", "port": port}
    if not orc.ref_available():
        return out
    try:
        g, e = 0.001, float(np.float32(1.0) - np.float32(1.0) / np.float32(6.0))

        def one(i):
            anc, des = raw[i]
            orc.ref_viterbi(table, g, e, 1, anc, des, a_cat[a_off[i]:a_off[i + 1]], b_cat[b_off[i]:b_off[i + 1]],
                            want_matrices=False)

        raw = [_host.synth_raw(i) for i in range(n_cal)]
        t0 = _t.perf_counter()
        for i in range(n_cal):
            one(i)
        t_one = (_t.perf_counter() - t0) / n_cal
        n_ref = int(min(n_all, max(cores, budget_s / max(t_one, 1e-4) * cores * 0.5)))
        raw = [_host.synth_raw(i) for i in range(n_ref)]
        with ThreadPoolExecutor(cores) as ex:  # (ctypes releases the GIL inside the engine)
            t0 = _t.perf_counter()
            list(ex.map(one, range(n_ref)))
            t_all = _t.perf_counter() - t0
        ref = {"gcups": float((la[:n_ref] * lb[:n_ref]).sum() / t_all / 1e9), "pairs_per_s": n_ref / t_all, "pairs": n_ref,
               "seconds": t_all, "single_thread_gcups": float((la[:n_cal] * lb[:n_cal]).sum() / (t_one * n_cal) / 1e9)}
        out.update({"value": ref["gcups"], "kind": "reference", "pairs_per_s": ref["pairs_per_s"], "reference": ref,
                    "sample": f"first {n_ref} pairs of the same synthetic set through the unmodified reference engine "
                              f"(oracle/_ref: viterbi_mem + traceback_viterbi), {cores} threads, {t_all:.1f} s"})
    except Exception as exc:  # the baseline must never fail the bench
        out["reference_error"] = repr(exc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=10000, help="pairs per GPU per step")
    ap.add_argument("--model", default="mar-mg", choices=["mar-mg", "mar-ecm"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational two_stream_pipeline and pcie_inclusive measurements (their extra "
                         "launches would be mixed into a rocprofv3 --stats summary of this command)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: consecutive steps alternate between two resident copies of the shard on two library "
                         "streams, so the ragged end of one launch overlaps the start of the next (DESIGN.md 4.1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from coati_amd import distributed as cd
    from coati_amd import hip, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # COATI_BENCH_SELFTEST_GATHER=1 runs the N > 1 code path (second slot, packed gather, overlap)
    # with a one-rank RCCL group: a functional check on a 1-GPU box, not a benchmark configuration
    selftest = world == 1 and os.environ.get("COATI_BENCH_SELFTEST_GATHER") == "1"
    multi = world > 1 or selftest
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if selftest:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available() or hip.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no gfx950 device visible (there is no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    # ---- model: built on rank 0, broadcast over RCCL --------------------------
    if rank == 0:
        table, consts, gap_len = host.set_subst(args.model), host.gap_consts(), 1
    else:
        table = consts = gap_len = None
    if multi:
        table, consts, gap_len = cd.broadcast_model(table, consts, gap_len, device)
    model = hip.Model(table, consts, gap_len, device=local_rank)

    # ---- this rank's shard of the synthetic workload, uploaded once ------------
    first = rank * args.pairs
    a_cat, a_off, b_cat, b_off = host.synth_encoded(first, args.pairs)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    cells = batch.cells
    seq_bytes = int(a_off[-1] + b_off[-1])

    # N > 1: every step's results go to rank 0 (one packed RCCL gather per step).  Two resident
    # copies of the shard alternate, so that the gather of step i runs on torch's collective stream
    # while the kernel of step i+1 runs: nothing is skipped, K launches and K gathers happen inside
    # the timed region.  By default both copies use ONE library stream (kernels never overlap each
    # other; the gather waits for its own launch only, coati_hip_viterbi_wait); --streams 2 gives each
    # copy its own stream.
    slots = [(model, batch)]
    gathers = []
    if multi or args.streams == 2:
        model2 = hip.Model(table, consts, gap_len, device=local_rank) if args.streams == 2 else model
        slots.append((model2, hip.Batch(model2, a_cat, a_off, b_cat, b_off)))
    if multi:
        gathers = [cd.PackedGather(cd.batch_result_tensors(bt, device), dst=0) for _, bt in slots]
    state = {"i": 0, "pending": None}

    def step():
        cur = state["i"] % len(slots)
        slots[cur][1].viterbi_launch()
        if multi:
            drain()
            state["pending"] = cur
        state["i"] += 1

    def drain():
        pend = state["pending"]
        if pend is not None:
            slots[pend][1].wait()  # its own launch only: the launch just issued keeps running
            gathers[pend](async_op=True)
            state["pending"] = None

    def fence():
        if multi:
            drain()
            for g in gathers:
                g.finish()
            dist.barrier()
        for _, bt in slots:
            bt.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel device times of the timed launches: HIP events the library recorded on its
    # own stream around each kernel (it keeps the last 64 launches)
    n_timed = max(1, min(args.steps // len(slots), 64))
    timed = [bt.viterbi_timing(i) for _, bt in slots for i in range(n_timed)]
    fill_ms = [t[0] for t in timed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([cells], dtype=torch.float64, device=device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = float(c.item())
    else:
        total_cells = float(cells)

    # sanity: results are real (score of pair 0 finite, ops consume both sequences)
    scores, ops, ops_off, ops_len = batch.viterbi_fetch()
    assert np.isfinite(scores).all() and int(ops_len.min()) >= 900

    if rank == 0:
        fill = float(np.mean(fill_ms))
        kernel_ms = {"viterbi_l1 (fill + fused traceback)": fill}
        if args.streams == 2:
            # launches overlap: the event pair around one kernel also spans its wait behind the other
            # stream's kernel, so the average launch duration is taken as region time / launches
            kernel_ms["viterbi_l1 event span incl. queueing behind the other stream"] = fill
            fill = elapsed / args.steps * 1e3
            kernel_ms["viterbi_l1 (fill + fused traceback)"] = fill
        algo_bytes = cells * ALGO_BYTES_PER_CELL + seq_bytes
        traffic = None
        tfile = ROOT / "profiles" / "traffic_latest.json"
        if tfile.exists() and args.pairs == 10000 and args.model == "mar-mg":
            try:
                traffic = json.loads(tfile.read_text()).get("viterbi_l1_bytes_per_launch_10000_pairs")
            except Exception:
                traffic = None
        # Informational (never `value`): the same K steps alternating between two resident copies of the
        # batch on two library streams, i.e. how a pipeline of batches runs -- the ragged end of one
        # launch overlaps the start of the next.  Skipped when --streams 2 is the mode
        # being measured anyway.
        pipelined = None
        if world == 1 and not multi and args.streams == 1 and not args.no_extras:
            m2 = hip.Model(table, consts, gap_len, device=local_rank)
            b2 = hip.Batch(m2, a_cat, a_off, b_cat, b_off)
            pp = [batch, b2]
            for i in range(4):
                pp[i % 2].viterbi_launch()
            for bt in pp:
                bt.sync()
            tp = time.perf_counter()
            for i in range(args.steps):
                pp[i % 2].viterbi_launch()
            for bt in pp:
                bt.sync()
            tp = time.perf_counter() - tp
            pipelined = {"gcups": cells * args.steps / tp / 1e9, "ms_per_step": tp / args.steps * 1e3,
                         "what": "K launches alternating between two resident copies of the batch on two streams "
                                 "(bench.py --streams 2 measures this mode as `value`)"}
            b2.close()
            m2.close()
        # PCIe-inclusive rate of the one-shot ABI call (upload + kernels + download of ops/scores);
        # reported next to `value`, never as `value`
        e2e = 1e30
        res = None
        for _ in range(0 if args.no_extras else 3):
            t0 = time.perf_counter()
            res = model.viterbi(a_cat, a_off, b_cat, b_off, out=res)
            e2e = min(e2e, time.perf_counter() - t0)
        out = {
            "metric": "GCUPS (DP cell updates/s), marginal Viterbi fill+traceback, mar-mg94 1kb x 1kb pairs",
            "value": total_cells * args.steps / elapsed / 1e9,
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.pairs} synthetic 1 kb x 1 kb pairs per GPU, {args.model} "
                                   "(BASELINE.json configs[1]; generator SURVEY.md §8(d))",
                       "streams": args.streams, "pairs_per_gpu": args.pairs, "global_pairs": args.pairs * world, "gap_len": 1,
                       "parallelism": f"pairs sharded over {world} GPU(s), model broadcast + result gather (RCCL)"},
            "pairs_per_s": args.pairs * world * args.steps / elapsed,
            "kernel_ms": kernel_ms,
            "roofline": {"bound": "hbm", "achieved": algo_bytes / (fill * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": algo_bytes / (fill * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "viterbi_l1", "algorithmic_bytes_per_launch": algo_bytes,
                         "valu_ceiling_gcups": VALU_PEAK_GCUPS, "valu_frac": cells / (fill * 1e-3) / 1e9 / VALU_PEAK_GCUPS,
                         "note": "the kernel is VALU-issue bound, not HBM bound: see DESIGN.md §4"},
            "two_stream_pipeline": pipelined,
            "pcie_inclusive": None if args.no_extras else {
                "gcups": cells / e2e / 1e9, "pairs_per_s": args.pairs / e2e, "ms": e2e * 1e3,
                "what": "coati_hip_viterbi_batch on rank 0: H2D of the encoded batch + kernels + D2H of scores/ops, "
                        "pageable host memory, one call (best of 3; calls after the first reuse the workspace the model cached "
                        "and write into result arrays whose pages exist)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table, consts, a_cat, a_off, b_cat, b_off)
        print(json.dumps(out), flush=True)
    if selftest and rank == 0:
        # the gathered copy of the last step equals what a direct fetch returns
        sc, ops, off, ln = gathers[(state["i"] - 1) % len(slots)].unpack(0)
        f_sc, f_ops, f_off, f_ln = slots[(state["i"] - 1) % len(slots)][1].viterbi_fetch()
        assert (sc.cpu().numpy().view(np.uint32) == f_sc.view(np.uint32)).all() and (ln.cpu().numpy() == f_ln).all()
        assert (ops.cpu().numpy() == f_ops).all()
        print("selftest: gathered results identical to a direct fetch", file=sys.stderr)
    for md, bt in slots:
        bt.close()
    for md in {id(m): m for m, _ in slots}.values():
        md.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
