set -u
O=gpurun_out/r5lay; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_band.py tests/test_gpu_long.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
AB_PAIRS="10000 40000" bash tools/experiments/r5_ab.sh r5lay coati_amd/_build/ab/tile6.so coati_amd/_build/ab/lay.so
COATI_HIP_CK_BAND=0 AB_PAIRS="10000" bash tools/experiments/r5_ab.sh r5lay_bandoff coati_amd/_build/ab/tile6.so coati_amd/_build/ab/lay.so
