set -u
O=gpurun_out/r5full; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.txt
python3 tools/stream_probe.py 10000 > $O/stream_probe_10000.txt 2>&1
timeout 600 python3 tools/ab_fill.py --pairs 10000 --rounds 8 coati_amd/_build/ab/libcoati_hip_base.so coati_amd/_build/ab/libcoati_hip_head.so coati_amd/_build/libcoati_hip.so > $O/ab.txt 2>&1
