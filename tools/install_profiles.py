#!/usr/bin/env python3
"""Copy the summaries tools/profile.sh left under gpurun_out/ into profiles/<round>/ (tracked)."""
import json
import shutil

import numpy as np
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = ROOT / "gpurun_out" / "prof_n1", ROOT / "profiles" / rnd
dst.mkdir(parents=True, exist_ok=True)
for a, b in (("kernel_stats.csv", "bench_n1_kernel_stats.csv"), ("pmc_summary.csv", "bench_n1_pmc_summary.csv"),
             ("traffic.json", "bench_n1_traffic.json"), ("bench.json", "bench_n1_under_rocprof.json")):
    shutil.copy(src / a, dst / b)
# the streamed one-shot call: kernel statistics + the probe's wall times
strm = ROOT / "gpurun_out" / "prof_stream"
if (strm / "kernel_stats.csv").exists():
    shutil.copy(strm / "kernel_stats.csv", dst / "stream_kernel_stats.csv")
    shutil.copy(strm / "stream_probe.txt", dst / "stream_probe_under_rocprof.txt")
# the exact-stream sampler: kernel statistics of tools/sample_bench.py
smp = ROOT / "gpurun_out" / "prof_sample"
if (smp / "kernel_stats.csv").exists():
    shutil.copy(smp / "kernel_stats.csv", dst / "sample_kernel_stats.csv")
# (steady.txt and short_pairs.txt are hand-kept records of several tool runs: not overwritten here)
for extra in ("bench_n1.json",
              ):
    f = ROOT / "gpurun_out" / extra
    if f.exists():
        shutil.copy(f, dst / extra)
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (kernel_sources_sha16: ties the figure to the kernel build it was measured on)

# the REAL 160 kb pair and the Forward fill in both modes (tools/profile_long.sh): kernel statistics + counters, and ONE record per
# kernel family tied to its sources (bench.py: recorded_kernel_profile -- the entry goes null when the sources change)
import csv  # noqa: E402


def kernel_record(directory, kernel_prefix, family, files_prefix, workload):
    d = ROOT / "gpurun_out" / directory
    if not (d / "kernel_stats.csv").exists():
        return None
    shutil.copy(d / "kernel_stats.csv", dst / f"{files_prefix}_kernel_stats.csv")
    shutil.copy(d / "pmc_summary.csv", dst / f"{files_prefix}_pmc_summary.csv")
    for extra in ("times.txt", "fwd.txt"):
        if (d / extra).exists():
            shutil.copy(d / extra, dst / f"{files_prefix}_{extra}")
    rec = {"family": family, "sources_sha16": bench.sources_sha16(family), "round": int(rnd[1:]), "workload": workload,
           "source": [f"profiles/{rnd}/{files_prefix}_kernel_stats.csv", f"profiles/{rnd}/{files_prefix}_pmc_summary.csv"]}
    for r in csv.DictReader(open(d / "kernel_stats.csv")):
        if r["Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1].split("<")[0].startswith(kernel_prefix):
            rec.update({"kernel": r["Name"][:120], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6, "min_ms": float(r["MinNs"]) * 1e-6,
                        "max_ms": float(r["MaxNs"]) * 1e-6})
            break
    for r in csv.DictReader(open(d / "pmc_summary.csv")):
        if r["kernel"].startswith(kernel_prefix):
            rec[r["counter"]] = float(r["mean_value"])
    if "WRITE_SIZE" in rec and "FETCH_SIZE" in rec:
        rec["hbm_bytes_fetch_x2"] = (rec["WRITE_SIZE"] + 2.0 * rec["FETCH_SIZE"]) * 1024.0
        rec["hbm_bytes_fetch_raw"] = (rec["WRITE_SIZE"] + rec["FETCH_SIZE"]) * 1024.0
    return rec


sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

records = {}
for key, args in (("viterbi_lp", ("prof_long", "viterbi_lp", "viterbi_lp", "long_pair", "tests/golden/long_pairs.npz 160k: 160 002 x 160 002 nt, one pair")),
                  ("forward_l1_exact", ("prof_fwd_exact", "forward_l1_exact", "forward_l1", "forward_exact", "6 144 synthetic pairs of 1 kb, exact mode")),
                  ("forward_l1_tolerance", ("prof_fwd_tolerance", "forward_l1_fast", "forward_l1", "forward_tolerance", "6 144 synthetic pairs of 1 kb, tolerance mode"))):
    rec = kernel_record(*args)
    if rec is not None:
        records[key] = rec
if records:
    (ROOT / "profiles" / "kernel_profiles_latest.json").write_text(json.dumps(records, indent=1))
    print(json.dumps(records, indent=1))

t = json.loads((dst / "bench_n1_traffic.json").read_text())["viterbi_ck"]
(ROOT / "profiles" / "traffic_latest.json").write_text(json.dumps({
    "viterbi_ck_bytes_per_launch_10000_pairs": t["bytes_per_launch"],
    "kernel_sources_sha16": bench.kernel_sources_sha16(),
    "how": "tools/profile.sh: rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes of `bench.py --steps 3 "
           "--warmup 1`; bytes = (WRITE_SIZE + 2*FETCH_SIZE)*1024 (gfx950: FETCH_SIZE reports half of a coalesced "
           "stream, MI355X_MICROARCH.md section HBM)",
    "WRITE_SIZE_KB": t["WRITE_SIZE_KB"], "FETCH_SIZE_KB_raw": t["FETCH_SIZE_KB_raw"], "round": int(rnd[1:]),
    "source": f"profiles/{rnd}/bench_n1_pmc_summary.csv"}, indent=1))
print(t)
# SQ counters of the headline kernel (tools/pmc_fill.sh <tag> 10000 libcoati_hip.so -> gpurun_out/pmc_<tag>/libcoati_hip.txt):
# the record bench.py's roofline.valu is computed from, tied to the kernel sources like the traffic record
sq_src = ROOT / "gpurun_out" / (sys.argv[2] if len(sys.argv) > 2 else "pmc_headline") / "libcoati_hip.txt"
if sq_src.exists():
    shutil.copy(sq_src, dst / "pmc_fill_sq_counters.txt")
    rec = {}
    for line in sq_src.read_text().splitlines():
        f = line.rsplit(None, 3)  # (the kernel's name may hold blanks: "viterbi_ck<true, true>")
        if len(f) == 4 and f[0].startswith("viterbi_ck"):
            rec[f[1]] = float(f[3])
    from coati_amd import host  # noqa: E402

    la, lb = host.synth_lengths(0, 10000)
    ns = rec.get("KERNEL_NS_group3") or rec.get("KERNEL_NS_group1")
    out = {k: rec[k] for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                               "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_SALU",
                               "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD") if k in rec}
    out.update({"cells": float((la.astype(np.float64) * lb).sum()), "pairs": 10000, "kernel_ms": ns * 1e-6 if ns else None,
                "kernel_sources_sha16": bench.kernel_sources_sha16(), "round": int(rnd[1:]),
                "how": "tools/pmc_fill.sh: one rocprofv3 --pmc pass per counter group over 4 launches of the resident 10 000-pair batch "
                       "(tools/fill_loop.py); means over the dispatches; kernel_ms = the kernel's duration in the GRBM_GUI_ACTIVE pass",
                "source": f"profiles/{rnd}/pmc_fill_sq_counters.txt"})
    (ROOT / "profiles" / "sq_latest.json").write_text(json.dumps(out, indent=1))
    print(out)
