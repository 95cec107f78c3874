set -u
O=gpurun_out/r5fsub; mkdir -p $O
for pass in 1 2 3; do
  for L in coati_amd/_build/libcoati_hip.so coati_amd/_build/ab/libcoati_hip_fsub8.so coati_amd/_build/ab/libcoati_hip_fsub4.so; do
    echo "== $L" >> $O/fwd.txt
    COATI_HIP_LIB=$L timeout 300 python3 tools/fwd_time.py 16 >> $O/fwd.txt 2>&1
    COATI_HIP_LIB=$L timeout 300 python3 tools/fwd_time.py 1 >> $O/fwd.txt 2>&1
  done
done
COATI_HIP_LIB=coati_amd/_build/ab/libcoati_hip_fsub8.so timeout 600 python -m pytest tests/test_gpu_generic.py tests/test_gpu_sample.py -x -q 2>&1 | tail -3 > $O/pytest8.txt
