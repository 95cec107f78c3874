# the streamed call's last chunk: size (tenths of a unit of 1e9 cells) and row parts, through tools/stream_probe.py (one model, warm)
set -u
O=gpurun_out/r5streamtail; mkdir -p $O
for rep in 1 2; do
for cfg in "26 23" "36 23" "46 23" "46 24" "56 23"; do
  set -- $cfg
  for n in 10000 20000; do
    echo -n "tail $1 parts $2 pairs $n: " >> $O/out.txt
    COATI_HIP_STREAM_TAIL_UNITS=$1 COATI_HIP_STREAM_PARTS=$2 python3 tools/stream_probe.py $n 8 2>&1 | grep "stream pinned\|resident" | cut -d: -f2 | tr '\n' '|' >> $O/out.txt
    echo >> $O/out.txt
  done
done
done
