set -u
O=gpurun_out/r5tile6; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_band.py -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
AB_PAIRS="10000" bash tools/experiments/r5_ab.sh r5tile6 coati_amd/_build/ab/tile5.so coati_amd/_build/ab/tile6.so
COATI_HIP_CK_SPLIT=10000,4 AB_PAIRS="10000" bash tools/experiments/r5_ab.sh r5tile6_allcut coati_amd/_build/ab/tile5.so coati_amd/_build/ab/tile6.so
COATI_HIP_CK_SPLIT=0 AB_PAIRS="10000" bash tools/experiments/r5_ab.sh r5tile6_nocut coati_amd/_build/ab/tile5.so
