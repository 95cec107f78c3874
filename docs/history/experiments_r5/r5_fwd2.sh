set -u
O=gpurun_out/r5fwd2; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_generic.py tests/test_gpu_sample.py tests/test_gpu_math.py tests/test_gpu_cli.py -x -q 2>&1 | tail -5 ) > $O/pytest.txt
for pass in 1 2 3; do
  for L in coati_amd/_build/ab/libcoati_hip_head.so coati_amd/_build/libcoati_hip.so; do
    echo "== $L" >> $O/fwd.txt
    COATI_HIP_LIB=$L timeout 300 python3 tools/fwd_time.py 16 >> $O/fwd.txt 2>&1
    COATI_HIP_LIB=$L timeout 300 python3 tools/fwd_time.py 64 >> $O/fwd.txt 2>&1
    COATI_HIP_LIB=$L timeout 300 python3 tools/fwd_time.py 6144 >> $O/fwd.txt 2>&1
  done
done
for i in 1 2; do python3 tools/split_ab.py 10000 5904,3,s3 5904,4,s3 5904,4,s2 5904,5,s3 7500,3,s3 >> $O/split.txt 2>&1; done
