set -u
O=gpurun_out/r5tile7; mkdir -p $O
( COATI_HIP_CK_DEBUG=4 timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_band.py -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
for pass in 1 2 3; do
for D in 0 4; do
echo "== dbg $D" >> $O/ab.txt
COATI_HIP_CK_DEBUG=$D timeout 600 python3 tools/ab_fill.py --pairs 10000 --rounds 8 coati_amd/_build/ab/tile7.so >> $O/ab.txt 2>&1
done
done
for D in 0 4; do
echo "== all cut, dbg $D" >> $O/ab.txt
COATI_HIP_CK_SPLIT=10000,4 COATI_HIP_CK_DEBUG=$D timeout 600 python3 tools/ab_fill.py --pairs 10000 --rounds 8 coati_amd/_build/ab/tile7.so >> $O/ab.txt 2>&1
done
