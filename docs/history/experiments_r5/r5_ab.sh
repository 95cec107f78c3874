# same-box A/B of builds of libcoati_hip.so: usage r5_ab.sh <tag> lib1 lib2 ...   (alternating, three passes)
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for pass in 1 2 3; do
  for P in ${AB_PAIRS:-10000 40000}; do
    echo "== pass $pass pairs $P" >> $O/ab.txt
    timeout 600 python3 tools/ab_fill.py --pairs $P --rounds 8 "$@" >> $O/ab.txt 2>&1
  done
done
