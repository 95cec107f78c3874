set -u
O=gpurun_out/r5prio; mkdir -p $O
for pass in 1 2 3; do
  for D in 0 16 32; do
    echo -n "pass $pass dbg $D: " >> $O/ab.txt
    COATI_HIP_CK_DEBUG=$D timeout 300 python3 tools/ab_fill.py --pairs 10000 --rounds 8 coati_amd/_build/libcoati_hip.so | cut -c34-140 >> $O/ab.txt 2>&1
  done
done
for D in 0 16; do
  echo -n "40000 dbg $D: " >> $O/ab.txt
  COATI_HIP_CK_DEBUG=$D timeout 300 python3 tools/ab_fill.py --pairs 40000 --rounds 8 coati_amd/_build/libcoati_hip.so | cut -c34-140 >> $O/ab.txt 2>&1
done
COATI_HIP_CK_DEBUG=16 COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python3 tools/experiments/tail_trace.py 10000 > $O/tail_prio16.txt 2>&1
