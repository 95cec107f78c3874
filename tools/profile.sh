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
# (the Forward fill in both modes and the real long pair: tools/profile_long.sh)
