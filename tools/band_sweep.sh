# banded checkpoints: kernel time for several half widths (0 = everything kept), 10 000 and 40 000 pairs
for band in 0 48 64 96 128 192; do
  for n in 10000 40000; do
    echo -n "BAND=$band pairs=$n  "; COATI_HIP_CK_BAND=$band python tools/ab_fill.py --pairs $n --rounds 12 coati_amd/_build/libcoati_hip.so | cut -c34-110
  done
done
COATI_HIP_CK_BAND=96 COATI_HIP_CK_DEBUG=2 python tools/fill_loop.py 10000 2 2>&1 | tail -3
COATI_HIP_CK_BAND=16 COATI_HIP_CK_DEBUG=2 python tools/fill_loop.py 10000 2 2>&1 | tail -3
