set -u
O=gpurun_out/r5streamwalk; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_golden.py tests/test_gpu_band.py tests/test_gpu_cli.py tests/test_gpu_dist.py -x -q 2>&1 | tail -5 ) > $O/pytest.txt
for pass in 1 2 3; do
  for WI in 0 1; do
    echo "== pass $pass walk_items $WI" >> $O/probe.txt
    COATI_HIP_CK_WALK_ITEMS=$WI python3 tools/stream_probe.py 10000 2>&1 | grep "resident\|stream pinned\|stream pageable" >> $O/probe.txt
  done
done
python3 tools/stream_check.py > $O/stream_check.txt 2>&1
