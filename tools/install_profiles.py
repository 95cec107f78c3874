#!/usr/bin/env python3
"""Copy the summaries tools/profile.sh left under gpurun_out/ into profiles/<round>/ (tracked)."""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = ROOT / "gpurun_out" / "prof_n1", ROOT / "profiles" / rnd
dst.mkdir(parents=True, exist_ok=True)
for a, b in (("kernel_stats.csv", "bench_n1_kernel_stats.csv"), ("pmc_summary.csv", "bench_n1_pmc_summary.csv"),
             ("traffic.json", "bench_n1_traffic.json"), ("bench.json", "bench_n1_under_rocprof.json")):
    shutil.copy(src / a, dst / b)
# the exact Forward fill: kernel statistics + PMC counters (tools/profile.sh, second half)
fwd = ROOT / "gpurun_out" / "prof_fwd"
if (fwd / "pmc_summary.csv").exists():
    shutil.copy(fwd / "kernel_stats.csv", dst / "forward_kernel_stats.csv")
    shutil.copy(fwd / "pmc_summary.csv", dst / "forward_pmc_summary.csv")
    shutil.copy(fwd / "fwd.json", dst / "forward_fwd_time.txt")
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
