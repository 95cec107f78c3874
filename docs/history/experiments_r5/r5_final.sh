set -u
O=gpurun_out/r5final; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.txt
bash tools/profile.sh n1 > $O/profile.log 2>&1
bash tools/pmc_fill.sh headline 10000 coati_amd/_build/libcoati_hip.so > $O/pmc.log 2>&1
PMC_WORKLOAD=fwd bash tools/pmc_fill.sh fwd 6144 coati_amd/_build/libcoati_hip.so > $O/pmc_fwd.log 2>&1
timeout 900 python3 bench.py > gpurun_out/bench_n1.json 2> $O/bench.err
python3 tools/dist_sim_bench.py 1000000 $O/dist_simulate_1M.json > $O/dist_sim.log 2>&1
