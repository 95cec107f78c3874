#!/bin/bash
# SQ counters of the Viterbi kernel on a fixed workload (tools/fill_loop.py), one rocprofv3 pass per counter group,
# for one or more builds of the library.  usage (GPU box, repo root): bash tools/pmc_fill.sh <tag> <pairs> lib1.so [lib2.so ...]
# PMC_WORKLOAD=fwd: the exact Forward fill instead (tools/fwd_time.py <pairs>)
# -> gpurun_out/pmc_<tag>/<libname>.txt : kernel, counter, mean over the dispatches
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; PAIRS="$2"; shift 2
OUT="$ROOT/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "${PMC_WORKLOAD:-fill}" = "fwd" ]; then WORKLOAD_PY=fwd_time.py; WORKLOAD_ARG=1; else WORKLOAD_PY=fill_loop.py; WORKLOAD_ARG=4; fi
CGROUPS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
        "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH")
for LIB in "$@"; do
    export COATI_HIP_LIB="$(cd "$ROOT" && realpath "$LIB")"
    NAME="$(basename "$LIB" .so)"
    : > "$OUT/$NAME.txt"
    g=0
    for G in "${CGROUPS[@]}"; do
        g=$((g+1))
        rm -rf "$OUT/raw_${NAME}_$g"
        timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/raw_${NAME}_$g" -o x -- python3 "$ROOT/tools/${WORKLOAD_PY}" "$PAIRS" ${WORKLOAD_ARG} >> "$OUT/$NAME.log" 2>&1
        python3 - "$OUT/raw_${NAME}_$g" >> "$OUT/$NAME.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
        if "viterbi" in name or "forward" in name:
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, c), v in sorted(acc.items()):
    print(f"{name:24s} {c:28s} {len(v):3d} {sum(v)/len(v):18.1f}")
# the kernels' durations under this counter group (--kernel-trace): what GRBM_GUI_ACTIVE / 8 is divided by for the clock
dur = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
        if "viterbi" in name or "forward" in name:
            dur[name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for name, v in sorted(dur.items()):
    print(f"{name:24s} {'KERNEL_NS_group' + sys.argv[1][-1]:28s} {len(v):3d} {sum(v)/len(v):18.1f}")
PY
        rm -rf "$OUT/raw_${NAME}_$g"
    done
    echo "== $NAME"; cat "$OUT/$NAME.txt"
done
