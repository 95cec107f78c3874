set -u
O=gpurun_out/r5pace; mkdir -p $O
for pass in 1 2 3; do
  for E in "0,0" "105,62" "95,70" "120,55" "90,0"; do
    echo -n "pass $pass exp $E: " >> $O/ab.txt
    COATI_HIP_CK_EXP=$E timeout 300 python3 tools/ab_fill.py --pairs 10000 --rounds 8 coati_amd/_build/ab/libcoati_hip_prioexp.so | cut -c34-110 >> $O/ab.txt 2>&1
  done
done
COATI_HIP_CK_EXP=105,62 COATI_HIP_LIB=coati_amd/_build/ab/libcoati_hip_prioexp_trace.so python3 tools/experiments/tail_trace.py 10000 > $O/tail_pace.txt 2>&1
