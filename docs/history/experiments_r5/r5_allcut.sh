set -u
O=gpurun_out/r5allcut; mkdir -p $O
for i in 1 2 3; do
  for WI in 0 1; do
    echo "== walk_items $WI" >> $O/split.txt
    COATI_HIP_CK_WALK_ITEMS=$WI python3 tools/split_ab.py 10000 5904,3,s3 10000,3,s3 8000,3,s3 >> $O/split.txt 2>&1
  done
done
COATI_HIP_CK_SPLIT=10000,3,s3 COATI_HIP_CK_WALK_ITEMS=1 COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/experiments/tail_trace.py 10000 > $O/tail_allcut_walk1.txt 2>&1
