#!/bin/bash
# GPU-box recipe behind profiles/rNN/: kernel-trace stats, then the two HBM PMC counters in
# separate passes (MI355X_MICROARCH.md, HBM/rocprofv3 section), all of the SAME bench command.
# usage (from the repo root on the GPU box): bash tools/profile.sh [tag] [extra bench.py args...]
# Results: gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_summary.csv,traffic.json,bench.json}
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:-n1}"; shift || true
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$OUT/bench.json" 2> "$OUT/bench.stderr"
for C in WRITE_SIZE FETCH_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o bench -- \
        python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_pmc_$C.json" 2> "$OUT/bench_pmc_$C.stderr"
done
python3 "$ROOT/tools/summarize_profile.py" "$OUT"
# the streamed one-shot call (one persistent viterbi_ck_stream launch per call): kernel statistics of
# tools/stream_probe.py (40 000 pairs, page-locked and pageable arrays, both pipeline forms)
STR="$ROOT/gpurun_out/prof_stream"
mkdir -p "$STR"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$STR/stats" -o stream -- \
    python3 "$ROOT/tools/stream_probe.py" 40000 4 > "$STR/stream_probe.txt" 2> "$STR/stream_probe.stderr"
cp "$(find "$STR/stats" -name '*kernel_stats.csv' | head -1)" "$STR/kernel_stats.csv" 2>/dev/null
# the exact-stream sampler (configs[3]: 16 pairs x 1 000 samples): kernel statistics of tools/sample_bench.py
SMP="$ROOT/gpurun_out/prof_sample"
mkdir -p "$SMP"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$SMP/stats" -o sample -- \
    python3 "$ROOT/tools/sample_bench.py" --pairs 64 > "$SMP/sample_bench.json" 2> "$SMP/sample_bench.stderr"
cp "$(find "$SMP/stats" -name '*kernel_stats.csv' | head -1)" "$SMP/kernel_stats.csv" 2>/dev/null
# the exact Forward fill (configs[3]): kernel statistics and the counters that say what its ~440 VALU slots per
# cell and its spills cost (separate passes, same command)
FWD="$ROOT/gpurun_out/prof_fwd"
mkdir -p "$FWD"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$FWD/stats" -o fwd -- \
    python3 "$ROOT/tools/fwd_time.py" 6144 > "$FWD/fwd.json" 2> "$FWD/fwd.stderr"
for C in SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES WRITE_SIZE FETCH_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$FWD/pmc_$C" -o fwd -- \
        python3 "$ROOT/tools/fwd_time.py" 6144 > /dev/null 2> "$FWD/pmc_$C.stderr"
done
python3 - "$FWD" <<'PY'
import csv, glob, sys, shutil
from collections import defaultdict
from pathlib import Path
out = Path(sys.argv[1])
st = glob.glob(str(out / "stats" / "**" / "*kernel_stats.csv"), recursive=True)
if st: shutil.copy(st[0], out / "kernel_stats.csv")
acc = defaultdict(list)
for f in glob.glob(str(out / "pmc_*" / "**" / "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(out / "pmc_summary.csv", "w") as fh:
    fh.write("kernel,counter,dispatches,mean_value\n")
    for (name, c), v in sorted(acc.items()):
        fh.write(f"{name},{c},{len(v)},{sum(v)/len(v):.3f}\n")
PY
