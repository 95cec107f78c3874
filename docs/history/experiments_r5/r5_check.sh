set -u
O=gpurun_out/r5check; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.txt
for pass in 1 2 3; do
  for P in 10000 40000; do
    echo "== pass $pass pairs $P" >> $O/ab.txt
    timeout 600 python3 tools/ab_fill.py --pairs $P --rounds 8 coati_amd/_build/ab/libcoati_hip_base.so coati_amd/_build/libcoati_hip.so >> $O/ab.txt 2>&1
  done
done
python3 tools/stream_probe.py 10000 > $O/stream_probe_10000.txt 2>&1
python3 tools/dist_sim_bench.py 1000000 $O/dist_simulate_1M.json > $O/dist_sim.log 2>&1
