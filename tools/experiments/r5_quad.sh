set -u
O=gpurun_out/r5quad; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_generic.py tests/test_gpu_sample.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.txt
for Q in 0 1; do
  echo "== COATI_HIP_FWD_QUAD=$Q" >> $O/sample_bench.txt
  COATI_HIP_FWD_QUAD=$Q timeout 600 python3 tools/sample_bench.py >> $O/sample_bench.txt 2>&1
done
