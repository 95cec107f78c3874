set -u
O=gpurun_out/r5rot3; mkdir -p $O
cp coati_amd/_build/ab/rot.so coati_amd/_build/libcoati_hip.so
for rep in 1 2; do
for D in 0 8; do
  for n in 3000 6000 20000; do
    echo -n "dbg $D pairs $n: " >> $O/out.txt
    COATI_HIP_CK_DEBUG=$D python3 tools/stream_probe.py $n 8 2>&1 | grep "stream pinned\|resident" | cut -d: -f2 | tr '\n' '|' >> $O/out.txt
    echo >> $O/out.txt
  done
done
done
