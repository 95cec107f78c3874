set -u
O=gpurun_out/r5band; mkdir -p $O
tools/ubench/_build/fwd_rates > $O/ubench_fwd_rates3.txt 2>&1
for band in 0 32 48 64 80 96 128; do
  for n in 10000 40000; do
    echo -n "BAND=$band pairs=$n  " >> $O/band.txt; COATI_HIP_CK_BAND=$band python3 tools/ab_fill.py --pairs $n --rounds 10 coati_amd/_build/libcoati_hip.so | cut -c34-110 >> $O/band.txt
  done
  COATI_HIP_CK_BAND=$band COATI_HIP_CK_DEBUG=2 python3 tools/fill_loop.py 10000 2 2>&1 | grep "pairs filled twice" | tail -1 | sed 's/.*walker iterations.pair, //' >> $O/band.txt
done
