set -u
O=gpurun_out/${1:-r5band}; mkdir -p $O
for rep in 1 2; do
for band in 32 48 64 80; do
  for n in 10000 40000; do
    echo -n "BAND=$band pairs=$n  " >> $O/band.txt; COATI_HIP_CK_BAND=$band python3 tools/ab_fill.py --pairs $n --rounds 10 coati_amd/_build/libcoati_hip.so | cut -c34-110 >> $O/band.txt
  done
  COATI_HIP_CK_BAND=$band COATI_HIP_CK_DEBUG=2 python3 tools/fill_loop.py 10000 2 2>&1 | grep "pairs filled twice" | tail -1 | sed 's/.*walker iterations.pair, //' >> $O/band.txt
done
done
