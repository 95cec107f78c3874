#!/bin/bash
# GPU-box recipe behind profiles/rNN/long_pair_*: the synthetic 160 kb pair (BASELINE configs[2]) through viterbi_lp --
# kernel-trace stats, the SQ instruction counters (separate passes), the per-strip timeline of the trace build.
# usage (from the repo root on the GPU box): bash tools/profile_long.sh
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_long"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o lp -- \
    python3 "$ROOT/tools/long_pair.py" --reps 5 > "$OUT/long_pair.json" 2> "$OUT/long_pair.stderr"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
for C in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE WRITE_SIZE FETCH_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o lp -- \
        python3 "$ROOT/tools/long_pair.py" --reps 2 > /dev/null 2> "$OUT/pmc_$C.stderr"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
from pathlib import Path
out = Path(sys.argv[1])
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
cd "$ROOT"
COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/trace_long.py > "$OUT/timeline.txt" 2>&1
python3 tools/long_golden_time.py 10k 20k 40k 80k 160k > "$OUT/golden_pairs.txt" 2>&1
