set -u
O=gpurun_out/${1:-r5rot}; mkdir -p $O
for pass in 1 2; do
for P in ${ROT_PAIRS:-1500 2000 3000 4000 4096 4500 5000 8000}; do
for D in 0 8; do
echo -n "pairs $P dbg $D: " >> $O/ab.txt
COATI_HIP_CK_DEBUG=$D timeout 600 python3 tools/ab_fill.py --pairs $P --rounds 8 coati_amd/_build/ab/rot.so | cut -c34-100 >> $O/ab.txt 2>&1
done
done
done
