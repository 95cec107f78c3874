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
