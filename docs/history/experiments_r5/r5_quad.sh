set -u
O=gpurun_out/r5quad; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_generic.py tests/test_gpu_sample.py tests/test_gpu_golden.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest.txt
timeout 400 python3 tools/fuzz.py 240 7301 > $O/fuzz.txt 2>&1
timeout 300 python3 tools/fuzz_sample.py 120 7302 >> $O/fuzz.txt 2>&1
for Q in 0 1; do
  echo "== COATI_HIP_FWD_QUAD=$Q" >> $O/sample_bench.txt
  COATI_HIP_FWD_QUAD=$Q timeout 600 python3 tools/sample_bench.py >> $O/sample_bench.txt 2>&1
done
python3 tools/experiments/quad_sweep.py 1 4 16 32 48 64 > $O/sweep_final.txt 2>&1
