set -u
O=gpurun_out/r5walk2; mkdir -p $O
( COATI_HIP_CK_WALK_ITEMS=1 timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_golden.py tests/test_gpu_band.py -x -q 2>&1 | tail -5 ) > $O/pytest.txt
for pass in 1 2 3; do
  for P in 10000 40000 16000; do
    for WI in 0 1; do
      echo -n "pass $pass pairs $P walk_items $WI: " >> $O/ab.txt
      COATI_HIP_CK_WALK_ITEMS=$WI timeout 300 python3 tools/ab_fill.py --pairs $P --rounds 8 coati_amd/_build/libcoati_hip.so | cut -c34-120 >> $O/ab.txt 2>&1
    done
  done
done
COATI_HIP_CK_WALK_ITEMS=1 COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/experiments/tail_trace.py 10000 > $O/tail_walk1.txt 2>&1
COATI_HIP_CK_WALK_ITEMS=0 COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/experiments/tail_trace.py 10000 > $O/tail_walk0.txt 2>&1
