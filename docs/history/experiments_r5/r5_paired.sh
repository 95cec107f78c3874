set -u
O=gpurun_out/r5paired; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_golden.py tests/test_gpu_band.py tests/test_gpu_long.py -x -q 2>&1 | tail -8 ) > $O/pytest.txt
for P in 10000 40000; do
  for D in 0 2; do
    echo "== pairs $P CK_DEBUG $D" >> $O/ab.txt
    COATI_HIP_CK_DEBUG=$D timeout 300 python3 tools/fill_loop.py $P 6 >> $O/ab.txt 2>&1
  done
done
