# tile-major colin: parity of the Viterbi paths, then the same-box A/B against the step-major build, then HBM traffic
# usage: r5_tile.sh <tag> lib1 lib2 ...   (libraries under coati_amd/_build/ab/, without .so)
set -u
ROOT=$(pwd)
TAG=$1; shift
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_band.py tests/test_gpu_long.py -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
LIBS=""; for L in "$@"; do LIBS="$LIBS coati_amd/_build/ab/$L.so"; done
bash tools/experiments/r5_ab.sh $TAG $LIBS
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
 for PAIRS in 10000 40000; do
  export COATI_HIP_LIB=$ROOT/coati_amd/_build/ab/$L.so
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/raw_${L}_$C
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/raw_${L}_$C -o x -- python3 $ROOT/tools/fill_loop.py $PAIRS 4 >> $O/pmc.log 2>&1
    python3 - $O/raw_${L}_$C $L $PAIRS >> $O/traffic.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "viterbi_ck" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"{sys.argv[2]:8s} {sys.argv[3]:>6s} pairs {c:12s} {len(v):3d} launches, mean {sum(v)/len(v)/1e6:10.4f} GB (raw KB counter / 1e6)")
PY
    rm -rf $O/raw_${L}_$C
  done
 done
done
