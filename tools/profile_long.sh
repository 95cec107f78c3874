#!/bin/bash
# GPU-box recipe behind profiles/rNN/long_pair_* and forward_*: the REAL 160 kb pair (BASELINE configs[2]:
# tests/golden/long_pairs.npz 160k = sampledata/example-160k.fasta, sanitised) through the planner's kernel (viterbi_lp) and
# the Forward fill in both modes -- kernel-trace stats, then the counters in SEPARATE passes (program directly after `--`).
# usage (from the repo root on the GPU box): bash tools/profile_long.sh
# Results: gpurun_out/prof_long/{kernel_stats.csv,pmc_summary.csv,times.txt}, gpurun_out/prof_fwd_{exact,tolerance}/...
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
summarise() {  # $1 = output directory with stats/ and pmc_*/
python3 - "$1" <<'PY'
import csv, glob, shutil, sys
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
}
OUT="$ROOT/gpurun_out/prof_long"; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o lp -- \
    python3 "$ROOT/tools/long_golden_time.py" 160k 160k > "$OUT/times.txt" 2> "$OUT/times.stderr"
for C in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE WRITE_SIZE FETCH_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o lp -- \
        python3 "$ROOT/tools/long_golden_time.py" 160k > /dev/null 2> "$OUT/pmc_$C.stderr"
done
summarise "$OUT"
for MODE in exact tolerance; do
    FWD="$ROOT/gpurun_out/prof_fwd_$MODE"; rm -rf "$FWD"; mkdir -p "$FWD"
    if [ "$MODE" = tolerance ]; then export COATI_HIP_FORWARD_FAST=1; else unset COATI_HIP_FORWARD_FAST; fi
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$FWD/stats" -o fwd -- \
        python3 "$ROOT/tools/fwd_time.py" 6144 > "$FWD/fwd.txt" 2> "$FWD/fwd.stderr"
    for C in SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE WRITE_SIZE FETCH_SIZE; do
        timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$FWD/pmc_$C" -o fwd -- \
            python3 "$ROOT/tools/fwd_time.py" 6144 > /dev/null 2> "$FWD/pmc_$C.stderr"
    done
    summarise "$FWD"
done
unset COATI_HIP_FORWARD_FAST
