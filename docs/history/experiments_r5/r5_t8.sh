set -u
O=gpurun_out/r5tile8; mkdir -p $O
cp coati_amd/_build/ab/tile8.so coati_amd/_build/libcoati_hip.so
( timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_band.py tests/test_gpu_long.py -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
AB_PAIRS="10000 40000" bash tools/experiments/r5_ab.sh r5tile8 coati_amd/_build/ab/tile6.so coati_amd/_build/ab/tile8.so coati_amd/_build/ab/tile8_2.so
