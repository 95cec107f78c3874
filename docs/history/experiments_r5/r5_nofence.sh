set -u
O=gpurun_out/r5nofence; mkdir -p $O
for pass in 1 2 3; do
  for D in 0 8 64 72; do
    echo -n "pass $pass dbg $D: " >> $O/ab.txt
    COATI_HIP_CK_DEBUG=$D timeout 300 python3 tools/ab_fill.py --pairs 10000 --rounds 8 coati_amd/_build/ab/libcoati_hip_nofence.so | cut -c34-140 >> $O/ab.txt 2>&1
  done
done
