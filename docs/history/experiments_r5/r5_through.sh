set -u
O=gpurun_out/r5through; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_golden.py tests/test_gpu_band.py -x -q 2>&1 | tail -5 ) > $O/pytest.txt
for pass in 1 2 3; do
  for P in 10000 40000; do
    echo "== pass $pass pairs $P" >> $O/ab.txt
    timeout 600 python3 tools/ab_fill.py --pairs $P --rounds 8 coati_amd/_build/ab/libcoati_hip_head.so coati_amd/_build/libcoati_hip.so >> $O/ab.txt 2>&1
  done
done
