set -u
mkdir -p gpurun_out/r5base
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r5base/pytest.txt
for P in 10000 40000; do
  for D in 0 1 2; do
    echo "== pairs $P CK_DEBUG $D" >> gpurun_out/r5base/ab.txt
    COATI_HIP_CK_DEBUG=$D timeout 300 python3 tools/ab_fill.py --pairs $P --rounds 8 coati_amd/_build/libcoati_hip.so >> gpurun_out/r5base/ab.txt 2>&1
  done
done
bash tools/pmc_fill.sh r5base 10000 coati_amd/_build/libcoati_hip.so > gpurun_out/r5base/pmc.log 2>&1
cp gpurun_out/pmc_r5base/libcoati_hip.txt gpurun_out/r5base/pmc_fill_sq_counters.txt
timeout 600 python3 bench.py > gpurun_out/r5base/bench.json 2> gpurun_out/r5base/bench.err
