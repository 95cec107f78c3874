set -u
O=gpurun_out/r5dist; mkdir -p $O
python3 tools/dist_sim_bench.py 1000000 $O/dist_simulate_1M.json > $O/dist_sim.log 2>&1
COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/experiments/tail_trace.py 10000 > $O/tail_10000.txt 2>&1
COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/experiments/tail_trace.py 40000 > $O/tail_40000.txt 2>&1
