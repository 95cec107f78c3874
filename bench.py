#!/usr/bin/env python3
"""Headline benchmark: GCUPS (DP cell updates/s) and pairs/s of the marginal
Viterbi path (fill + traceback) on synthetic 1 kb x 1 kb pairs, mar-mg94
(BASELINE.json configs[1]), inputs resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (fill kernel + traceback walker, and for
N > 1 the gather of all results to rank 0's HBM over RCCL) over one batch of
`--pairs` pairs PER GPU (weak scaling: the reference aligns one pair per
process, pairs are independent, shards need no data-path collective).
N > 1 runs on the product's own multi-GPU layer, libcoati_hip_dist.so (RCCL linked
directly: coati_hip_dist_init / _broadcast_model / _gather / _viterbi_shard); the
ncclUniqueId travels through the launcher's TCP store (MASTER_ADDR / MASTER_PORT)
before anything touches a GPU.  torch is used for that store and for
torch.cuda.synchronize() only.  The N > 1 line also carries `strong_1M`: BASELINE
configs[4] (1 000 000 pairs, mar-ecm) as ONE sharded job, each rank generating
only its shard.  Rank 0 prints ONE JSON line.
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
# Secondary (and for this recurrence the binding) ceiling, DESIGN.md §4 / §5b: the gap_len-1 fill cell (viterbi_ck) is
# 15 VALU instructions; its register-only replay with the constants in VGPRs issues at 13.2 ns per 64-lane cell per SIMD
# with 4 wavefronts per SIMD at 2.35 GHz (tools/ubench/gen_step.py "cell vgpr", profiles/r03/ubench_step_model.txt;
# round 2's 18.9 ns was the same cell with SGPR constants): 1024 SIMDs x 64 lanes / 13.2 ns = 4.96 TCUPS.
VALU_PEAK_GCUPS = 1024 * 64 / 13.2
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
                              f"(oracle/_ref: viterbi_mem + traceback_viterbi), {cores} threads, {t_all:.1f} s; the all-core figure is "
                              "bound by the engine's own allocations (three 4 MB matrices per pair, first-touched by every thread at once), "
                              f"not by arithmetic: {ref['gcups'] / ref['single_thread_gcups']:.0f}x one thread's {ref['single_thread_gcups']:.2f} GCUPS on {cores} threads"})
    except Exception as exc:  # the baseline must never fail the bench
        out["reference_error"] = repr(exc)
    return out


def kernel_sources_sha16():
    """Hash of the sources of the dominant kernel: ties a recorded PMC traffic figure to the build it was taken from."""
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCE_SETS["viterbi_ck"]:
        h.update((ROOT / "coati_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


KERNEL_SOURCE_SETS = {
    "viterbi_ck": ("viterbi_ck.hip", "viterbi_cell.hpp", "common.hpp", "plan.hip"),  # (plan.hip: which items the kernel works through)
    "viterbi_lp": ("viterbi_lp.hip", "viterbi_lp_block.inc", "gen_viterbi_lp.py", "viterbi_cell.hpp", "common.hpp", "plan.hip"),
    "forward_l1": ("forward_l1.hip", "glibc_math.hpp", "common.hpp", "plan.hip"),
}


def sources_sha16(kernel):
    """Hash of the sources of one kernel family: ties a recorded rocprofv3 figure to the build it was taken from."""
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCE_SETS[kernel]:
        h.update((ROOT / "coati_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def recorded_kernel_profile(kernel):
    """profiles/kernel_profiles_latest.json (tools/install_profiles.py): the rocprofv3 record of `kernel` -- average duration
    under --kernel-trace --stats, PMC counters of separate passes -- or None when its sources have changed since."""
    try:
        rec = json.loads((ROOT / "profiles" / "kernel_profiles_latest.json").read_text())[kernel]
        return rec if rec.get("sources_sha16") == sources_sha16(rec.get("family", kernel)) else None
    except Exception:
        return None


def recorded_profile(name):
    """A tracked record under profiles/ that tools/install_profiles.py tied to the kernel sources it was measured on (PMC
    passes cannot run inside a timed bench run): the record, or None when the kernel sources have changed since."""
    f = ROOT / "profiles" / name
    try:
        rec = json.loads(f.read_text())
        return rec if rec.get("kernel_sources_sha16") == kernel_sources_sha16() else None
    except Exception:
        return None


def roofline_record(algo_bytes, fill_ms, cells, traffic, traffic_rec, sq_rec, band_rec, power):
    """The headline kernel against the HBM roof SURVEY.md 8(d) prices it on -- and what actually binds it.  `bound` names
    the roof `achieved`/`peak`/`frac` refer to (the contract's figure); `binding` is what the counters say limits the
    kernel: vector instruction issue, at the clock the board's power limit leaves."""
    achieved = algo_bytes / (fill_ms * 1e-3) / 1e9
    rec = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": "recorded" if traffic is not None else None,
           "kernel": "viterbi_ck", "algorithmic_bytes_per_launch": algo_bytes, "band": band_rec, "binding": "valu_issue"}
    if traffic_rec:
        rec["traffic_detail"] = {"WRITE_SIZE_bytes": traffic_rec["WRITE_SIZE_KB"] * 1024, "FETCH_SIZE_bytes_raw": traffic_rec["FETCH_SIZE_KB_raw"] * 1024,
                                 "bytes_with_fetch_x2": traffic, "bytes_with_fetch_raw": (traffic_rec["WRITE_SIZE_KB"] + traffic_rec["FETCH_SIZE_KB_raw"]) * 1024,
                                 "source": traffic_rec.get("source"),
                                 "note": "recorded by tools/profile.sh on this kernel build (source hash checked), not measured in this run; the "
                                         "gfx950 FETCH_SIZE x 2 correction is specified for wide coalesced streams -- the traceback's reads are 8-byte "
                                         "and 16-byte gathers, so `traffic` is an upper bound and bytes_with_fetch_raw a lower one"}
    sclk = watts = cap = None
    if power and not power.get("error"):
        sclk = float(np.mean(power["sclk_mhz"])) if power.get("sclk_mhz") else None
        watts = float(np.mean(power["package_power_w"])) if power.get("package_power_w") else None
        cap = power.get("package_power_cap_w")
    valu = {"sclk_mhz": sclk, "watts": watts, "power_cap_w": cap, "power_source": "rocm-smi in this run (extra.power)" if watts else None}
    if sq_rec:
        ipc = sq_rec["SQ_INSTS_VALU"] * 64.0 / sq_rec["cells"]
        cyc = (sq_rec["GRBM_GUI_ACTIVE"] / 8.0) / (sq_rec["SQ_INSTS_VALU"] / 1024.0)  # shader cycles per VALU instruction and SIMD
        valu.update({"instr_per_cell": ipc, "cycles_per_instr_per_simd": cyc, "issue_frac": 2.0 / cyc,
                     "issue_frac_note": "against one wave64 VALU instruction per 2 cycles per SIMD (MI355X_MICROARCH.md); the cell's mix "
                                        "(11 v_add_f32 at ~2.15 cycles, v_max_f32 and 2 v_max3_f32 at ~4.2, DPP / readlane / LDS at their "
                                        "own cadence: tools/ubench) cannot issue at 2",
                     "clock_under_counters_mhz": sq_rec["GRBM_GUI_ACTIVE"] / 8.0 / (sq_rec["kernel_ms"] * 1e-3) / 1e6 if sq_rec.get("kernel_ms") else None,
                     "lds_bank_conflict_frac": sq_rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(sq_rec.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0),
                     "counters_source": "recorded: " + str(sq_rec.get("source"))})
        if sclk:
            valu["gcups_at_2_cycles_per_instr"] = 1024 * 64 * sclk * 1e6 / (ipc * 2.0) / 1e9
            valu["frac_of_that"] = cells / (fill_ms * 1e-3) / 1e9 / valu["gcups_at_2_cycles_per_instr"]
    rec["valu"] = valu
    rec["valu_ceiling_gcups"] = VALU_PEAK_GCUPS
    rec["valu_frac"] = cells / (fill_ms * 1e-3) / 1e9 / VALU_PEAK_GCUPS
    rec["note"] = ("frac is priced against the HBM roof as SURVEY.md 8(d) defines it (1 B/cell of traceback state); the kernel is NOT "
                   "HBM-bound (counter traffic is about half the algorithmic bytes: checkpoints are kept in a band around each pair's "
                   "diagonals only).  What binds is vector instruction issue -- `valu`: instructions per cell and cycles per instruction "
                   "from the recorded SQ counters of this build, clock and board power from this run (extra.power carries the driver's throttle "
                   "report taken during the launches: whether the clock below 2.4 GHz is the power limit's doing is what THAT says, not this "
                   "note); valu_ceiling_gcups is the register-only replay of the 15-instruction cell at 2.35 GHz "
                   "(tools/ubench/gen_step.py), a third reference point.  DESIGN.md 4.1, 5.6")
    return rec


def roofline_by_kernel(algo_bytes, fill_ms, extras):
    """One map with the roofline fraction of every kernel the line reports, so that nobody has to dig through `extra`."""
    out = {"viterbi_ck (configs[1]: 10 000 x 1 kb, fill + traceback)": {"bound": "hbm", "bytes_per_cell": 1.0, "achieved_GBps": algo_bytes / (fill_ms * 1e-3) / 1e9,
                                                                          "frac": algo_bytes / (fill_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "binding": "valu_issue"}}
    smp = extras.get("sample") or {}
    ff = (smp.get("forward_fill") or {}).get("exact")
    if ff:
        out["forward_l1_exact_narrow (configs[3] fill, 6 144 x 1 kb, bit-exact)"] = {
            "bound": "hbm", "bytes_per_cell": 12.0, "achieved_GBps": ff["gcups"] * 12.0, "frac": ff["hbm_frac_12B_per_cell"],
            "binding": "valu_issue (glibc expf / log1pf restated: 417 instructions per cell, a third of them 4-8-cycle classes)",
            "profile": recorded_kernel_profile("forward_l1_exact")}
    ft = (smp.get("forward_fill") or {}).get("tolerance")
    if ft:
        out["forward_l1_fast_wide (configs[3] fill, 6 144 x 1 kb, tolerance mode: log-weights within 1e-5)"] = {
            "bound": "hbm", "bytes_per_cell": 12.0, "achieved_GBps": ft["gcups"] * 12.0, "frac": ft["hbm_frac_12B_per_cell"],
            "binding": "transcendental issue (8 v_exp_f32 / v_log_f32 per cell at a quarter of the vector rate) next to 5.1 TB/s of "
                       "writes (a plain store stream reaches ~6.2 on this part)",
            "profile": recorded_kernel_profile("forward_l1_tolerance")}
    lp = extras.get("long_pair") or {}
    if lp.get("gcups"):
        out["viterbi_lp (configs[2]: one 160 002 x 160 002 pair)"] = {"bound": "hbm", "bytes_per_cell": 1.0, "achieved_GBps": lp["gcups"],
                                                                      "frac": lp["hbm_frac_1B_per_cell"],
                                                                      "binding": "latency: a chain of ~213 000 dependent wavefront steps (834 strips of 3 columns per lane, "
                                                                                 "one wavefront per SIMD), then the spliced traceback",
                                                                      "profile": recorded_kernel_profile("viterbi_lp")}
    s16 = smp.get("sample_16x1000_exact_stream")
    if s16:
        out["configs[3] as stated: 16 pairs, forward + 16 000 samples"] = {"forward_ms": s16["forward_ms"], "sampleback_first_call_ms": s16["sampleback_first_call_ms"],
                                                                         "sampleback_warm_ms": s16["sampleback_ms"],
                                                                         "binding": "latency: 63 quad strips of 16 columns in a row per pair (two log-sums per step); ~13 speculation rounds"}
    return out


def measure_extras(hip, host, model, table, consts, a_cat, a_off, b_cat, b_off, cells, args):
    """Informational records next to the headline (never `value`): the PCIe-inclusive streamed call
    (H2D -> kernels -> D2H), the CLI batch mode end to end, and BASELINE configs[2] and [3].
    Each is bounded to a few seconds; a failure of one is recorded, not raised."""
    import subprocess
    import tempfile
    import zlib

    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # an extra must never fail the bench line
            out[name] = {"error": repr(exc)}

    def streamed():
        res = {}
        for n_pairs in (len(a_off) - 1, 40000):
            if n_pairs == len(a_off) - 1:
                ac, ao, bc, bo, cl = a_cat, a_off, b_cat, b_off, cells
            else:
                ac, ao, bc, bo = host.synth_encoded(0, n_pairs)
                cl = int((np.diff(ao).astype(np.float64) * np.diff(bo)).sum())
            pa, pb = hip.pinned_copy(ac), hip.pinned_copy(bc)
            # a model's FIRST call allocates the call's workspaces (GBs of HBM, page-locked staging); an embedder that knows
            # a call is coming asks for them ahead of it -- coati_hip_model_prepare, e.g. on a helper thread while it reads its
            # input (coati-alignpair --batch does): measured on a model of its own, prepare and the first call after it apart
            prep = None
            if n_pairs == len(a_off) - 1:
                mp = hip.Model(table, consts, 1)
                t0 = time.perf_counter()
                mp.prepare(n_pairs, int(np.diff(ao).max()), int(np.diff(bo).mean()))
                t_prep = time.perf_counter() - t0
                outw = (hip.pinned_empty(n_pairs, np.float32), hip.pinned_empty(int(ao[-1] + bo[-1]), np.uint8), hip.pinned_empty(n_pairs, np.uint64),
                        hip.pinned_empty(n_pairs, np.uint32))
                for arr in outw:
                    arr[...] = 0  # (the result arrays exist: page-locking them is the caller's own bring-up)
                t0 = time.perf_counter()
                mp.viterbi(pa, ao, pb, bo, out=outw, pinned=True)
                prep = {"prepare_ms": t_prep * 1e3, "first_call_after_prepare_ms": (time.perf_counter() - t0) * 1e3}
                mp.close()
            calls, outp = [], None
            for _ in range(7):
                t0 = time.perf_counter()
                outp = model.viterbi(pa, ao, pb, bo, out=outp, pinned=True)
                calls.append(time.perf_counter() - t0)
            # (the first call of a model allocates slots and first-touches the result arrays: reported, not mixed in)
            first_call, best, median = calls[0], min(calls[1:]), float(np.median(calls[1:]))
            pcalls, outq = [], None
            for _ in range(5):
                t0 = time.perf_counter()
                outq = model.viterbi(ac, ao, bc, bo, out=outq)
                pcalls.append(time.perf_counter() - t0)
            pageable = float(np.median(pcalls[1:]))
            assert np.isfinite(outp[0]).all() and (outp[3] == outq[3]).all()
            # the same pairs resident in HBM (what `value` measures), in this process on this GPU: kernel time by HIP events
            bt = hip.Batch(model, ac, ao, bc, bo)
            ks = []
            for _ in range(4):
                bt.viterbi_launch()
                bt.sync()
                ks.append(sum(bt.viterbi_timing()))
            resident = float(np.median(ks[1:])) * 1e-3
            want = bt.viterbi_fetch()
            bt.close()
            model.trim()
            same = bool((outp[0].view(np.uint32) == want[0].view(np.uint32)).all() and (outp[3] == want[3]).all())
            res[f"{n_pairs}_pairs"] = {"gcups": cl / median / 1e9, "pairs_per_s": n_pairs / median, "ms": median * 1e3,
                                       "best_ms": best * 1e3, "first_call_ms": first_call * 1e3, "calls": len(calls),
                                       "pageable_ms": pageable * 1e3, "pageable_gcups": cl / pageable / 1e9,
                                       "resident_kernel_ms": resident * 1e3, "inclusive_over_resident": resident / median,
                                       "best_over_resident": resident / best,
                                       "pageable_over_resident": resident / pageable, "scores_and_lengths_equal_resident": same}
            if prep is not None:
                res[f"{n_pairs}_pairs"].update(prep)
        res["what"] = ("coati_hip_viterbi_batch, wall time of ONE call from Python: plan + H2D of the encoded pairs + kernel + D2H of "
                       "scores/ops; from 4 096 pairs of >= 250 x 250 cells ONE persistent kernel (viterbi_ck_stream) fed chunk by chunk "
                       "over 12 slots (HBM workspace + page-locked staging), uploads on another stream, results stored by the kernel straight into host memory "
                       "(the caller's page-locked arrays, or the slot's staging block for pageable ones: round 6, no downloads); caller arrays page-locked "
                       "(coati_hip_host_alloc) resp. pageable; `ms` / `gcups` = MEDIAN of calls 2..7 (4 of the pageable form), the first "
                       "call -- slot allocation, first touch of the result arrays -- and the best one reported beside it; "
                       "inclusive_over_resident = kernel time of the same pairs as one resident batch / the median wall time")
        return res

    def long_pair():
        from tests import util  # (fixture decoder only)

        a, b, case, doc = util.load_long_pair("160k")
        tab = np.load(ROOT / "tests" / "golden" / doc["table"])
        m = hip.Model(tab, host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)
        bt = hip.Batch(m, *hip.pack_pairs([(a, b)]))
        ts = []
        for _ in range(4):
            bt.viterbi_launch()
            bt.sync()
            ts.append(bt.viterbi_timing()[0])
        sc, ops, off, ln = bt.viterbi_fetch()
        got = ops[int(off[0]):int(off[0]) + int(ln[0])]
        ok = int(np.float32(sc[0]).view(np.uint32)) == int(case["score_bits"], 16) and "%08x" % zlib.crc32(got.tobytes()) == case["ops_crc32"]
        cl = len(a) * len(b)
        ms = float(np.median(ts[1:]))
        r = {"pair": "tests/golden/long_pairs.npz 160k (sampledata/example-160k.fasta, sanitised): %d x %d nt" % (len(a), len(b)),
             "ms": ms, "gcups": cl / ms / 1e6, "hbm_frac_1B_per_cell": cl / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "bit_exact_vs_golden": bool(ok), "device_bytes": bt.device_bytes}
        bt.close()
        # the same pair with viterbi_ck forced (lean 15-instruction cell + checkpoint recompute, 4-column strips) next to the
        # planner's choice (viterbi_lp: 25-instruction bit cell, 2-column strips): which of the two a lone wavefront prefers
        forced = {}
        for name, env in (("viterbi_ck_4_columns", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "4"}),
                          ("viterbi_ck_8_columns", {"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "8"})):
            try:
                os.environ.update(env)
                hip.reload_env()
                b2 = hip.Batch(m, *hip.pack_pairs([(a, b)]))
                t2 = []
                for _ in range(3):
                    b2.viterbi_launch()
                    b2.sync()
                    t2.append(b2.viterbi_timing()[0])
                sc2, ops2, off2, ln2 = b2.viterbi_fetch()
                got2 = ops2[int(off2[0]):int(off2[0]) + int(ln2[0])]
                forced[name] = {"ms": float(np.median(t2[1:])), "gcups": cl / float(np.median(t2[1:])) / 1e6,
                                "bit_exact_vs_golden": bool(int(np.float32(sc2[0]).view(np.uint32)) == int(case["score_bits"], 16)
                                                            and "%08x" % zlib.crc32(got2.tobytes()) == case["ops_crc32"])}
                b2.close()
            except Exception as exc:
                forced[name] = {"error": repr(exc)[:200]}
            finally:
                for k in env:
                    os.environ.pop(k, None)
                hip.reload_env()
        r["forced"] = forced
        m.close()
        return r

    def sample():
        from oracle import pyoracle as orc  # checker (the error bound of the tolerance mode), outside every timed region

        n_fwd = 6144
        r = {"forward_fill": {}}
        # Both Forward modes of the C ABI (COATI_HIP_OPT_FORWARD_MODE, per model, in this process): `exact` -- glibc's expf /
        # log1pf restated, the reference's bits, the library's default -- and `tolerance` -- hardware exp2 / log2, what
        # north_star's "log-weights within 1e-5" allows.  Per mode: the bulk fill rate against the 12 B/cell roof, and BASELINE
        # configs[3] as stated (16 pairs, forward + 1 000 samples each) with the largest relative log-weight error of the
        # 16 000 samples against the oracle's evaluation of the SAME paths.
        enc16 = host.synth_encoded(0, 16)
        fills = None
        for mode_name, mode in (("exact", hip.FORWARD_EXACT), ("tolerance", hip.FORWARD_TOLERANCE)):
            m = hip.Model(table, consts, 1, forward_mode=mode)
            bt = hip.Batch(m, *host.synth_encoded(0, n_fwd))
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                bt.forward_launch()
                bt.sync()
                ts.append(time.perf_counter() - t0)
            t = min(ts[1:])
            r["forward_fill"][mode_name] = {"pairs": n_fwd, "ms": t * 1e3, "gcups": bt.cells / t / 1e9,
                                            "hbm_frac_12B_per_cell": bt.cells * 12 / t / 1e9 / HBM_PEAK_GBS}
            # the bulk kernel's values (16-column strips: not the quad strips the 16-pair case below runs on): terminal M, D, I of
            # every 96th pair against the oracle's Forward fill
            fin = bt.forward_final()
            a_c, a_o, b_c, b_o = host.synth_encoded(0, n_fwd)
            worst_f = 0.0
            for p in range(0, n_fwd, 96):
                a, b = a_c[int(a_o[p]):int(a_o[p + 1])], b_c[int(b_o[p]):int(b_o[p + 1])]
                M, D, I = orc.fill(orc.LOG, table, consts, 1, a, b)
                want = np.array([M[-1, -1], D[-1, -1], I[-1, -1]], np.float64)
                worst_f = max(worst_f, float(np.max(np.abs(fin[p].astype(np.float64) - want) / np.maximum(1.0, np.abs(want)))))
            r["forward_fill"][mode_name]["max_rel_error_of_terminal_mdi_vs_oracle_64_pairs"] = worst_f
            bt.close()
            m.trim()
            # configs[3]: ONE call per model, as `coati sample` makes it (first call), then the warm repeats
            bt = hip.Batch(m, *enc16)
            t0 = time.perf_counter()
            bt.forward_launch()
            bt.sync()
            t_f = time.perf_counter() - t0
            states = np.array([host.rng_seed(["42"]) for _ in range(16)], np.uint64)
            # the result arrays are the caller's.  `coati-sample` hands in zero-FILLED vectors (host/align.cc marg_sample: pages that
            # exist); numpy's zeros are lazily mapped and would be first touched page by page under the download (+1.7 ms for these
            # 32 MB, tools/sample_first_call.py) -- so: arrays written once before the first call, as the CLI's are
            total_ops = int(1000 * bt.lens.sum())
            res = (np.ones((16, 1000), np.float32), np.ones(max(total_ops, 1), np.uint8), np.ones((16, 1000), np.uint64),
                   np.ones((16, 1000), np.uint32), np.ones((16, 2), np.uint64))
            best, first = 1e30, None
            for _ in range(4):  # (first call of a fresh model: workspace + staging allocations, 18 rounds enqueued where 13 are needed)
                t0 = time.perf_counter()
                res = bt.sampleback(1000, states, independent=False, out=res)
                dt = time.perf_counter() - t0
                first = dt if first is None else first
                best = min(best, dt)
            lw, ops, off, ln = res[0], res[1], res[2], res[3]
            if fills is None:
                a_c, a_o, b_c, b_o = enc16
                fills = [orc.fill(orc.LOG, table, consts, 1, a_c[int(a_o[p]):int(a_o[p + 1])], b_c[int(b_o[p]):int(b_o[p + 1])]) for p in range(16)]
            worst = 0.0
            a_c, a_o, b_c, b_o = enc16
            for p in range(16):
                a, b = a_c[int(a_o[p]):int(a_o[p + 1])], b_c[int(b_o[p]):int(b_o[p + 1])]
                M, D, I = fills[p]
                for sidx in range(0, 1000, 1 if mode_name == "tolerance" else 10):
                    path = ops[int(off[p, sidx]):int(off[p, sidx]) + int(ln[p, sidx])]
                    want = float(orc.path_logweight(M, D, I, table, consts, 1, a, b, path))
                    worst = max(worst, abs(float(lw[p, sidx]) - want) / max(1.0, abs(want)))
            # ... and as `coati sample` runs it (host/align.cc marg_sample): a fresh model, the forward launch ENQUEUED, the sampler's
            # allocations (coati_hip_sampleback_prepare) under the Forward kernel, then the one sampleback call
            m2 = hip.Model(table, consts, 1, forward_mode=mode)
            b2 = hip.Batch(m2, *enc16)
            res2 = tuple(np.ones_like(x) for x in res)
            t0 = time.perf_counter()
            b2.forward_launch()
            b2.sampleback_prepare(1000)
            res2 = b2.sampleback(1000, states, independent=False, out=res2)
            t_cli = time.perf_counter() - t0
            same_cli = bool((res2[0].view(np.uint32) == lw.view(np.uint32)).all())
            b2.close()
            m2.close()
            r["forward_fill"][mode_name]["config3_16x1000"] = {
                "forward_ms": t_f * 1e3, "sampleback_first_call_ms": first * 1e3, "sampleback_warm_ms": best * 1e3,
                "as_the_cli_pays_it_ms": t_cli * 1e3, "as_the_cli_pays_it_same_log_weights": same_cli,
                "forward_plus_first_sampleback_timed_apart_ms": t_f * 1e3 + first * 1e3, "max_rel_logweight_error_vs_oracle": worst,
                "samples_checked": 16000 if mode_name == "tolerance" else 1600, "finite": bool(np.isfinite(lw).all())}
            if mode_name == "exact":
                r["sample_16x1000_exact_stream"] = {"forward_ms": t_f * 1e3, "sampleback_ms": best * 1e3, "sampleback_first_call_ms": first * 1e3,
                                                    "samples_per_s": 16000 / best, "finite": bool(np.isfinite(lw).all()),
                                                    "note": "sampleback_ms = the warm repeat; `coati sample` makes ONE call per process and pays "
                                                            "forward_ms + sampleback_first_call_ms (result arrays that exist, as its zero-filled vectors do; "
                                                            "into lazily mapped arrays the download first-touches 32 MB: +1.7 ms)"}
                lw_exact = lw
            bt.close()
            m.close()
        r["forward_fill"]["what"] = ("per Forward mode (COATI_HIP_OPT_FORWARD_MODE through coati_hip_model_set_option): fill of 6 144 pairs of 1 kb "
                                     "(74 GB of M/D/I), and configs[3] with the largest relative error of the samples' log-weights against the oracle")
        lw = lw_exact
        # the CPU beside it (comparison leg, outside every timed GPU region): the same 16 pairs, forward + 1 000 samples each,
        # through the unmodified reference engine where oracle/_ref travelled (align_pair.cc:149,401-458 via oracle/ref_shim.cc),
        # else the bit-identical port; one thread, and one pair per thread
        try:
            from concurrent.futures import ThreadPoolExecutor

            from oracle import pyoracle as orc  # checker / baseline only

            g, e = np.float32(0.001), np.float32(1.0) - np.float32(1.0) / np.float32(6.0)
            raw = [host.synth_raw(i) for i in range(16)]
            enc = [host.encode(anc, des) for anc, des in raw]
            use_ref = orc.ref_available()

            def one(i):
                (anc, des), (a, b) = raw[i], enc[i]
                if use_ref:
                    _, _, sc = orc.ref_forward_sample(table, g, e, 1, anc, des, a, b, ["42"], 1000, want_matrices=False)
                    return float(sc[0])
                M, D, I, E = orc.fill(orc.LOG, table, consts, 1, a, b, edges=True)
                mats, rng = np.concatenate([np.stack([M, D, I]), E]), orc.rng_seed(["42"])
                return float([orc.sampleback(mats, 1, rng)[1] for _ in range(1000)][0])

            t0 = time.perf_counter()
            first = [one(i) for i in range(4)]
            t1 = (time.perf_counter() - t0) / 4 * 16
            threads = min(16, os.cpu_count() or 1)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                allp = list(ex.map(one, range(16)))
            tn = time.perf_counter() - t0
            gpu_ms = r["sample_16x1000_exact_stream"]["forward_ms"] + r["sample_16x1000_exact_stream"]["sampleback_ms"]
            r["cpu_baseline"] = {"kind": "reference" if use_ref else "port", "what": "forward + 1 000 samples for each of the same 16 pairs",
                                 "one_thread_ms": t1 * 1e3, "one_thread_note": "4 pairs timed, scaled to 16",
                                 "one_pair_per_thread_ms": tn * 1e3, "threads": threads, "gpu_ms": gpu_ms,
                                 "gpu_over_one_thread": t1 * 1e3 / gpu_ms, "gpu_over_threads": tn * 1e3 / gpu_ms,
                                 "first_log_weight_equal": bool(np.float32(lw.reshape(16, -1)[0, 0]).view(np.uint32) == np.float32(allp[0]).view(np.uint32))}
        except Exception as exc:  # (never fails the line)
            r["cpu_baseline"] = {"error": repr(exc)}
        return r

    def cli_batch():
        exe = ROOT / "coati_amd" / "_build" / "coati-alignpair"
        n = 10000
        with tempfile.TemporaryDirectory() as td:
            fa, js = Path(td) / "pairs.fasta", Path(td) / "out.json"
            with open(fa, "w") as f:
                for i in range(n):
                    anc, des = host.synth_raw(i)
                    f.write(f">a{i}\n{anc}\n>d{i}\n{des}\n")
            env = dict(os.environ, COATI_HOST_TIMING="1")
            best, stages = 1e30, ""
            for _ in range(2):
                t0 = time.perf_counter()
                pr = subprocess.run([str(exe), "--batch", str(fa), "-o", str(js)], capture_output=True, text=True, env=env, timeout=300)
                dt = time.perf_counter() - t0
                if pr.returncode != 0:
                    raise RuntimeError(pr.stderr[-500:])
                if dt < best:
                    best, stages = dt, pr.stderr
            cl = sum(len(host.synth_raw(i)[0]) * len(host.synth_raw(i)[1]) for i in range(0, n, 100)) * 100
            st = {}
            for line in stages.splitlines():
                if ": " in line and " ms" in line:
                    name = line.split(": ", 1)[1].rsplit(" (total", 1)[0]
                    st[name.rsplit(" ", 2)[0]] = float(name.rsplit(" ", 2)[1])
            return {"what": "coati-alignpair --batch on a 10 000-pair FASTA -> JSON file, whole process wall time "
                            "(the analogue of the reference's benchmark/benchmark_main.cc.in:33-53: read + model + DP + write)",
                    "s": best, "pairs_per_s": n / best, "gcups_approx": cl / best / 1e9, "stage_ms": st,
                    "output_bytes": js.stat().st_size}

    def reference_suite():
        """The reference's own benchmark harness (benchmark/benchmark_main.cc.in:56-76, libcoati-benchmark-tests.txt:1-7):
        the seven BM_marg_alignment inputs.  Per case: the GPU's kernel time for the pair alone and for a batch of 64
        copies (HIP events, median of 3 launches; results checked against the fixture), next to the unmodified
        reference engine on ONE host thread of this box (as its benchmark runs it), where it fits the time budget."""
        from tests import util  # (fixture decoder only)
        from oracle import pyoracle as orc  # the CPU side of the comparison -- never on the product path

        keys = ["156", "1k", "2k", "4k", "8k", "16k", "32k"]
        doc = util.load_bench_pair("156")[3]
        tab = np.load(ROOT / "tests" / "golden" / doc["table"])
        g, e = doc["gap_open"], doc["gap_extend"]
        m = hip.Model(tab, host.gap_consts(g, e), 1)
        rows, ref_ns_per_cell = [], None
        for k in keys:
            a, b, case, _ = util.load_bench_pair(k)
            cl = len(a) * len(b)
            row = {"case": "bm_" + k, "len_a": len(a), "len_b": len(b)}
            for copies, tag in ((1, "pair"), (64, "batch64")):
                bt = hip.Batch(m, *hip.pack_pairs([(a, b)] * copies))
                ts = []
                for _ in range(4):
                    bt.viterbi_launch()
                    bt.sync()
                    ts.append(sum(bt.viterbi_timing()))
                sc, ops, off, ln = bt.viterbi_fetch()
                ok = all(int(np.float32(sc[p]).view(np.uint32)) == int(case["score_bits"], 16) and
                         "%08x" % zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes()) == case["ops_crc32"] for p in (0, copies - 1))
                ms = float(np.median(ts[1:]))
                row[tag + "_ms"] = ms
                row[tag + "_gcups"] = cl * copies / ms / 1e6
                row[tag + "_bit_exact"] = bool(ok)
                bt.close()
            m.trim()
            if orc.ref_available() and (ref_ns_per_cell is None or cl * ref_ns_per_cell * 2e-9 < 45.0):
                letters = np.array(list("ACGT"))
                des = "".join(letters[b])
                codons = util.SENSE64  # encoded ancestor -> nucleotides: code = codon61 * 3 + phase
                anc = "".join(util.codon_str(codons[int(c) // 3]) for c in a[0::3])
                t0 = time.perf_counter()
                orc.ref_viterbi(tab, np.float32(g), np.float32(e), 1, anc, des, a, b, want_matrices=False)
                dt = time.perf_counter() - t0
                ref_ns_per_cell = dt / cl * 1e9
                row["reference_engine_ms_1_thread"] = dt * 1e3
                row["pair_speedup_vs_reference"] = dt * 1e3 / row["pair_ms"]
            rows.append(row)
        m.close()
        return {"what": "benchmark/benchmark_main.cc.in BM_marg_alignment inputs: kernel ms on the GPU (pair alone / 64 copies in one "
                        "batch) and the unmodified reference engine (oracle/_ref, 1 host thread)", "cases": rows}

    def band_sensitivity():
        """The banded checkpoints on inputs that are NOT the benign synthetic set: 10 000 synthetic 1 kb pairs of which 5 % /
        25 % carry one 90-300 nt deletion or insertion in the descendant (a path that leaves the kept band is filled
        twice), and the reference's own benchmark pair bm_1k (benchmark/data/benchmark_1k.fasta: 85 deletion and 46
        insertion columns) x 10 000.  Per bag: kernel ms / GCUPS with the default band and with everything kept, the
        number of pairs filled twice, and that both runs give the same bits."""
        from tests import util  # (fixture decoder only)

        rng = np.random.default_rng(2024)
        n = 10000

        def with_indels(frac):
            a_parts, b_parts = [], []
            for p in range(n):
                a = a_cat[int(a_off[p]):int(a_off[p + 1])]
                b = b_cat[int(b_off[p]):int(b_off[p + 1])]
                if rng.random() < frac:
                    size = int(rng.integers(90, 301))
                    at = int(rng.integers(0, max(1, len(b) - size)))
                    if rng.random() < 0.5:
                        b = np.concatenate([b[:at], b[at + size:]])
                    else:
                        b = np.concatenate([b[:at], rng.integers(0, 4, size).astype(np.uint8), b[at:]])[:1020]
                a_parts.append(a)
                b_parts.append(b)
            return hip.pack_pairs(list(zip(a_parts, b_parts)))

        def run(packed, mdl):
            bt = hip.Batch(mdl, *packed)
            res = {}
            crc = {}
            for tag, band in (("band_default", None), ("band_off", 0)):
                if band is not None:
                    mdl.set_option(hip.OPT_CK_BAND, band)
                ts = []
                for i in range(6):
                    bt.viterbi_launch()
                    bt.sync()
                    if i >= 2:
                        ts.append(bt.viterbi_timing()[0])
                steps, twice = bt.band_stats()
                sc, ops, off, ln = bt.viterbi_fetch()
                h = zlib.crc32(sc.tobytes())
                h = zlib.crc32(ln.tobytes(), h)
                for p in range(0, len(ln), 7):
                    h = zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes(), h)
                crc[tag] = h
                ms = float(np.median(ts))
                res[tag] = {"ms": ms, "gcups": bt.cells / ms / 1e6, "band_steps": steps, "pairs_filled_twice": twice}
            # (back to what the model ran with before this bag: the first run's own record, not a guess)
            mdl.set_option(hip.OPT_CK_BAND, res["band_default"]["band_steps"])
            res["same_bits"] = crc["band_default"] == crc["band_off"]
            bt.close()
            return res

        out_b = {"what": "kernel time of 10 000-pair bags with the default checkpoint band and with everything kept; pairs filled "
                         "twice = paths that left the band"}
        out_b["indel_5pct"] = run(with_indels(0.05), model)
        out_b["default_band_steps"] = out_b["indel_5pct"]["band_default"]["band_steps"]
        out_b["indel_25pct"] = run(with_indels(0.25), model)
        a, b, case, doc = util.load_bench_pair("1k")
        tab = np.load(ROOT / "tests" / "golden" / doc["table"])
        m2 = hip.Model(tab, host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)
        out_b["bm_1k_x10000"] = run(hip.pack_pairs([(a, b)] * n), m2)
        m2.close()
        return out_b

    def power():
        """Board power (rocm-smi) while the headline batch is launched back to back for ~2.5 s: the fill runs at the package
        power cap.  Sampled from a thread of this process; rocm-smi is a child process."""
        import re
        import shutil
        import threading

        smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
        stop = threading.Event()

        def burn():
            while not stop.is_set():
                for _ in range(20):
                    batch.viterbi_launch()
                batch.sync()

        th = threading.Thread(target=burn)
        th.start()
        watts, cap, sclk = [], None, []
        throttle = None
        try:
            time.sleep(0.8)
            # why the clock is where it is WHILE the launches run: the driver's own throttle / limit report (whichever tool this
            # image has; recorded verbatim, trimmed) -- "power-bound" is a claim only this can carry
            for cmd in (["amd-smi", "metric", "-g", "0", "--throttle", "--json"], ["amd-smi", "metric", "-g", "0", "--throttle"],
                        [smi, "--showperflevel", "--showclkfrq", "--showvoltage"]):
                exe = shutil.which(cmd[0]) or (cmd[0] if os.path.exists(cmd[0]) else None)
                if exe is None:
                    continue
                try:
                    pr = subprocess.run([exe] + cmd[1:], capture_output=True, text=True, timeout=25)
                except Exception:
                    continue
                if pr.returncode == 0 and pr.stdout.strip():
                    throttle = {"command": " ".join(cmd)}
                    try:  # (amd-smi --json: the violation states and the accumulated counters, without the per-XCP lists)
                        t = json.loads(pr.stdout)["gpu_data"][0]["throttle"]
                        throttle.update({k: v for k, v in t.items() if not isinstance(v, (dict, list))})
                    except Exception:
                        throttle["output"] = pr.stdout.strip()[:1500]
                    break
            for _ in range(3):
                pr = subprocess.run([smi, "--showpower", "--showmaxpower", "--showclocks"], capture_output=True, text=True, timeout=20)
                m = re.search(r"Current Socket Graphics Package Power \(W\): ([0-9.]+)", pr.stdout)
                if m:
                    watts.append(float(m.group(1)))
                m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", pr.stdout)
                if m:
                    cap = float(m.group(1))
                m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", pr.stdout)
                if m:
                    sclk.append(int(m.group(1)))
        finally:
            stop.set()
            th.join()
            batch.close()
        if not watts:
            raise RuntimeError("rocm-smi reported no power")
        return {"what": "rocm-smi while the headline batch is launched back to back (GPU 0 of this box)", "package_power_w": watts,
                "package_power_cap_w": cap, "sclk_mhz": sclk, "frac_of_cap": (sum(watts) / len(watts) / cap) if cap else None,
                "throttle_report_during_launches": throttle}

    guarded("pcie_inclusive", streamed)
    guarded("band_sensitivity", band_sensitivity)
    guarded("power", power)
    guarded("reference_suite", reference_suite)
    guarded("cli_batch", cli_batch)
    guarded("long_pair", long_pair)
    guarded("sample", sample)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed launches (default: ~1 s of timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=10000, help="pairs per GPU per step")
    ap.add_argument("--model", default="mar-mg", choices=["mar-mg", "mar-ecm"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong-total", type=int, default=0,
                    help="BASELINE configs[4] style: a FIXED total of pairs split over the ranks by the native "
                         "partitioner (coati_hip_shard_bounds), scaling 'strong'; default 0 = --pairs per GPU (weak)")
    ap.add_argument("--strong-pairs", type=int, default=0,
                    help="N > 1 only: size of the `strong_1M` job (default 1 000 000; 20 000 in the one-rank self-test)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational two_stream_pipeline and pcie_inclusive measurements (their extra "
                         "launches would be mixed into a rocprofv3 --stats summary of this command)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: consecutive steps alternate between two resident copies of the shard on two library "
                         "streams, so the ragged end of one launch overlaps the start of the next (DESIGN.md 4.1)")
    args = ap.parse_args()

    import torch

    from coati_amd import hip, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # COATI_BENCH_SELFTEST_GATHER=1 runs the N > 1 code path (rendezvous through a TCP store, native communicator,
    # model broadcast, second slot, per-step gather, the sharded strong job) with a one-rank RCCL communicator:
    # a functional check on a 1-GPU box, not a benchmark configuration
    selftest = world == 1 and os.environ.get("COATI_BENCH_SELFTEST_GATHER") == "1"
    multi = world > 1 or selftest
    comm = None
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from coati_amd import dist as nd

        # ---- rendezvous BEFORE any GPU call: rank 0 makes the ncclUniqueId, the launcher's TCP store carries it
        # (coati_amd/dist.py: rendezvous_id; under torch.distributed.run the agent serves the store, by hand rank 0 does)
        uid = nd.rendezvous_id(world, rank)
        comm = nd.Comm(uid, world, rank, device=local_rank)
    if not torch.cuda.is_available() or hip.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no gfx950 device visible (there is no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    # ---- model: built on rank 0, broadcast over RCCL (coati_hip_dist_broadcast_model) ----------------
    if rank == 0:
        table, consts, gap_len = host.set_subst(args.model), host.gap_consts(), 1
    else:
        table = consts = gap_len = None
    if multi:
        table, consts, gap_len = comm.broadcast_model(table, consts, gap_len, root=0)
        table = table[0]
    model = hip.Model(table, consts, gap_len, device=local_rank)

    # ---- this rank's shard of the synthetic workload, uploaded once ------------
    if args.strong_total > 0:
        # every rank derives the same shard plan from the lengths alone (the generator is a function of the
        # pair index, so lengths of the whole set are cheap: only offsets are generated here)
        la_all, lb_all = host.synth_lengths(0, args.strong_total)
        bounds = hip.shard_bounds(np.concatenate([[0], np.cumsum(la_all)]), np.concatenate([[0], np.cumsum(lb_all)]), world)
        first, n_mine = int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
    else:
        first, n_mine = rank * args.pairs, args.pairs
    a_cat, a_off, b_cat, b_off = host.synth_encoded(first, n_mine)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    cells = batch.cells
    seq_bytes = int(a_off[-1] + b_off[-1])

    # N > 1: every step's results go to rank 0's HBM (coati_hip_dist_gather without the download: an all-gather of
    # the counts, then one group of ncclSend / ncclRecv out of the ranks' result arrays).  Two resident copies of
    # the shard alternate, so that the gather of step i runs on the communicator's stream while the kernel of step
    # i+1 runs: nothing is skipped, K launches and K gathers happen inside the timed region.  By default both
    # copies use ONE library stream (kernels never overlap each other; the gather waits for its own launch only,
    # coati_hip_viterbi_wait); --streams 2 gives each copy its own stream.
    slots = [(model, batch)]
    if multi or args.streams == 2:
        model2 = hip.Model(table, consts, gap_len, device=local_rank) if args.streams == 2 else model
        slots.append((model2, hip.Batch(model2, a_cat, a_off, b_cat, b_off)))
    state = {"i": 0, "pending": None, "counts": None}

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
            state["counts"] = comm.gather_device(slots[pend][1], root=0)  # (collective; waits for that launch only)
            state["pending"] = None

    def fence():
        if multi:
            drain()
            comm.barrier()
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
    if multi:
        elapsed = float(comm.allreduce([elapsed], "max")[0])
        total_cells = float(comm.allreduce([float(cells)], "sum")[0])
    else:
        total_cells = float(cells)
    strong = strong_job(comm, hip, host, torch, world, rank, local_rank, args) if multi else None

    global_pairs = args.strong_total if args.strong_total > 0 else args.pairs * world
    # sanity: results are real (score of pair 0 finite, ops consume both sequences)
    scores, ops, ops_off, ops_len = batch.viterbi_fetch()
    assert np.isfinite(scores).all() and (len(ops_len) == 0 or int(ops_len.min()) >= 900)

    if rank == 0:
        fill = float(np.mean(fill_ms))
        kernel_ms = {"viterbi_ck (fill + fused traceback)": fill}
        if args.streams == 2:
            # launches overlap: the event pair around one kernel also spans its wait behind the other
            # stream's kernel, so the average launch duration is taken as region time / launches
            kernel_ms["viterbi_ck event span incl. queueing behind the other stream"] = fill
            fill = elapsed / args.steps * 1e3
            kernel_ms["viterbi_ck (fill + fused traceback)"] = fill
        algo_bytes = cells * ALGO_BYTES_PER_CELL + seq_bytes
        # HBM bytes per launch from the PMC counters: a recorded measurement of THIS kernel build (tools/profile.sh
        # stores the hash of the kernel sources next to it); null when the sources have changed since
        traffic = traffic_rec = None
        if n_mine == 10000 and args.model == "mar-mg":
            traffic_rec = recorded_profile("traffic_latest.json")
            traffic = traffic_rec.get("viterbi_ck_bytes_per_launch_10000_pairs") if traffic_rec else None
        sq_rec = recorded_profile("sq_latest.json") if n_mine == 10000 and args.model == "mar-mg" else None
        # The headline kernel keeps its traceback checkpoints in a BAND around each pair's straight line; a pair whose
        # path leaves it is filled twice (same bits).  What that did on THIS workload, and what the same launches cost
        # with everything kept (COATI_HIP_OPT_CK_BAND = 0): part of the roofline record, measured here, never `value`.
        band_rec = None
        try:
            band_steps, refilled = batch.band_stats()
            if band_steps and args.no_extras:  # (the profiled command: headline launches only, so that rocprofv3's averages are the headline's)
                band_rec = {"steps": band_steps, "pairs_filled_twice": refilled, "pairs": n_mine}
            elif band_steps:
                model.set_option(hip.OPT_CK_BAND, 0)
                off_ms = []
                for i in range(8):
                    batch.viterbi_launch()
                    batch.sync()
                    if i >= 2:
                        off_ms.append(batch.viterbi_timing()[0])
                assert batch.band_stats() == (0, 0)
                model.set_option(hip.OPT_CK_BAND, band_steps)
                batch.viterbi_launch()
                batch.sync()
                band_rec = {"steps": band_steps, "pairs_filled_twice": refilled, "pairs": n_mine,
                            "band_off_ms": float(np.median(off_ms)), "band_off_gcups": cells / float(np.median(off_ms)) / 1e6,
                            "note": "the synthetic generator's indels (Poisson(2) per pair, mean 6 nt) keep every path inside the band; "
                                    "extra.band_sensitivity runs bags with long indels and a real pair"}
            else:
                band_rec = {"steps": 0, "pairs_filled_twice": 0, "pairs": n_mine}
        except Exception as exc:  # (never fails the line)
            band_rec = {"error": repr(exc)}
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
        extras = {} if args.no_extras or world > 1 else measure_extras(hip, host, model, table, consts, a_cat, a_off, b_cat, b_off,
                                                                         cells, args)
        out = {
            "metric": f"GCUPS (DP cell updates/s), marginal Viterbi fill+traceback, {'mar-ecm' if args.model == 'mar-ecm' else 'mar-mg94'} 1kb x 1kb pairs",
            "value": total_cells * args.steps / elapsed / 1e9,
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong_total > 0 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"{args.strong_total} synthetic 1 kb x 1 kb pairs split over the GPUs by DP cells, {args.model} "
                                    "(BASELINE.json configs[4] shape)" if args.strong_total > 0 else
                                    f"{args.pairs} synthetic 1 kb x 1 kb pairs per GPU, {args.model} "
                                    "(BASELINE.json configs[1]; generator SURVEY.md §8(d))"),
                       "streams": args.streams, "pairs_per_gpu": n_mine, "global_pairs": global_pairs, "gap_len": 1,
                       "parallelism": f"pairs sharded over {world} GPU(s); model broadcast + per-step result gather to rank 0's HBM "
                                      "through libcoati_hip_dist.so (RCCL linked directly)" if multi else "1 GPU"},
            "pairs_per_s": global_pairs * args.steps / elapsed,
            # SURVEY.md 8(d) defines the metric's wall time as H2D -> kernels -> D2H and asks for kernel-only beside it: `value` is
            # the kernel-only figure (inputs resident in HBM, as the bench contract prescribes); this is the same batch through ONE
            # coati_hip_viterbi_batch call from page-locked host arrays (median of calls 2..7; pcie_inclusive has the details)
            "value_inclusive": ((extras.get("pcie_inclusive") or {}).get(f"{n_mine}_pairs") or {}).get("gcups"),
            "kernel_ms": kernel_ms,
            "roofline": roofline_record(algo_bytes, fill, cells, traffic, traffic_rec, sq_rec, band_rec, extras.get("power")),
            "roofline_by_kernel": roofline_by_kernel(algo_bytes, fill, extras),
            "strong_1M": strong,
            "two_stream_pipeline": pipelined,
            "pcie_inclusive": extras.get("pcie_inclusive"),
            "extra": {k: v for k, v in extras.items() if k != "pcie_inclusive"} or None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table, consts, a_cat, a_off, b_cat, b_off)
        print(json.dumps(out), flush=True)
    if multi:
        # (untimed) one gather WITH the download: rank 0's part of the gathered arrays equals a direct fetch of its batch,
        # the counts are every rank's pairs / op bytes
        last = slots[(state["i"] - 1) % len(slots)][1]
        got = comm.gather(last, root=0)
        if rank == 0:
            counts, sc, ops, off, ln = got
            f_sc, f_ops, f_off, f_ln = last.viterbi_fetch()
            n0 = int(counts[0, 0])
            assert n0 == n_mine and int(counts[:, 0].sum()) == len(sc)
            assert (sc[:n0].view(np.uint32) == f_sc.view(np.uint32)).all() and (ln[:n0] == f_ln).all() and (off[:n0] == f_off).all()
            assert (ops[:int(counts[0, 1])] == f_ops[:int(counts[0, 1])]).all()
            print("gathered results of rank 0 identical to a direct fetch", file=sys.stderr)
    for md, bt in slots:
        bt.close()
    for md in {id(m): m for m, _ in slots}.values():
        md.close()
    if comm is not None:
        comm.close()


def strong_job(comm, hip, host, torch, world, rank, local_rank, args):
    """BASELINE configs[4] as ONE sharded job through coati_hip_dist_viterbi_shard: `--strong-pairs` synthetic 1 kb
    pairs (1 000 000 by default; 20 000 in the one-rank self-test), mar-ecm, split by coati_hip_shard_bounds; every
    rank generates and encodes ONLY its shard (all ranks know all LENGTHS: they are the plan); the ECM model is built on
    rank 0 and broadcast; results (scores, ops, offsets, lengths) end in rank 0's host arrays.  Wall time between two
    barriers, max over ranks.  Returns the record for the JSON line (rank 0), None elsewhere."""
    import zlib

    total = args.strong_pairs if args.strong_pairs > 0 else (20000 if world == 1 else 1000000)
    t_gen = time.perf_counter()
    la, lb = host.synth_lengths(0, total)
    a_off = np.concatenate([[0], np.cumsum(la)]).astype(np.uint64)
    b_off = np.concatenate([[0], np.cumsum(lb)]).astype(np.uint64)
    bounds = hip.shard_bounds(a_off, b_off, world)
    first, n_mine = int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
    a_cat, a_loc, b_cat, b_loc = host.synth_encoded(first, n_mine)
    assert int(a_loc[-1]) == int(a_off[first + n_mine] - a_off[first]) and int(b_loc[-1]) == int(b_off[first + n_mine] - b_off[first])
    t_gen = time.perf_counter() - t_gen
    if rank == 0:
        table, consts, gap_len = host.set_subst("mar-ecm"), host.gap_consts(), 1
    else:
        table = consts = gap_len = None
    table, consts, gap_len = comm.broadcast_model(table, consts, gap_len, root=0)
    model = hip.Model(table[0], consts, gap_len, device=local_rank)
    times = []
    res = None
    for _ in range(2):  # (the first job pays the first-touch allocation of workspaces and result arrays)
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = comm.viterbi_shard(model, a_cat, int(a_off[first]), a_off, b_cat, int(b_off[first]), b_off, root=0, reuse=res)
        comm.barrier()
        torch.cuda.synchronize()
        times.append(float(comm.allreduce([time.perf_counter() - t0], "max")[0]))
    # the same job with the results LEFT ON THE RANKS that computed them (coati_hip_dist_viterbi_shard_local: every rank
    # downloads its shard over its own PCIe link into page-locked arrays; scores + op lengths of all pairs are gathered
    # to rank 0 over RCCL) -- what `coati-alignpair --batch --devices` builds on: every rank formats its own slice
    times_local = []
    loc = None
    for _ in range(3):  # (the model's first streamed call allocates its slots, the second the last chunk's workspace)
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loc, summary = comm.viterbi_shard_local(model, a_cat, int(a_off[first]), a_off, b_cat, int(b_off[first]), b_off, root=0, reuse=loc)
        comm.barrier()
        torch.cuda.synchronize()
        times_local.append(float(comm.allreduce([time.perf_counter() - t0], "max")[0]))
    model.close()
    # every rank's local results must be the gathered ones (rank 0 checks its own shard against the gather-all job)
    if rank == 0:
        sc, ops, off, ln = res
        lsc, lops, loff, lln = loc
        n0 = len(lsc)
        same = bool((lsc.view(np.uint32) == sc[:n0].view(np.uint32)).all() and (lln == ln[:n0]).all() and (loff == off[:n0]).all()
                    and (summary[0].view(np.uint32) == sc.view(np.uint32)).all() and (summary[1] == ln).all())
    if rank != 0:
        return None
    cells = float((la.astype(np.float64) * lb).sum())
    assert np.isfinite(sc).all() and int(ln.min()) >= 900 and (off.astype(np.int64) + ln <= np.cumsum(la + lb)).all()
    return {"what": "BASELINE configs[4]: ONE sharded job (coati_hip_dist_viterbi_shard), shards by DP cells, each rank generated only "
                    "its shard, results in rank 0's host memory; wall time between barriers, max over ranks, second of two jobs",
            "pairs": total, "model": "mar-ecm", "n_gpus": world, "seconds": times[-1], "first_job_seconds": times[0],
            "gcups": cells / times[-1] / 1e9, "pairs_per_s": total / times[-1], "scaling": "strong",
            "shard_generation_seconds": t_gen, "scores_crc32": "%08x" % zlib.crc32(sc.tobytes()), "columns": int(ln.sum()),
            "local_results": {"what": "the same job (third of three) through coati_hip_dist_viterbi_shard_local: ops stay with the rank that computed them "
                                      "(own PCIe link, page-locked arrays), scores + op lengths of all pairs gathered to rank 0 over RCCL",
                              "seconds": times_local[-1], "first_job_seconds": times_local[0], "gcups": cells / times_local[-1] / 1e9,
                              "pairs_per_s": total / times_local[-1], "equal_to_gathered": same}}


if __name__ == "__main__":
    main()
