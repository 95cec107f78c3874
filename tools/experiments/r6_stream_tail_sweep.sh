for rep in 1 2; do
for P in -1 24 25 26; do
  echo "== parts $P"; COATI_HIP_PIPE=stream COATI_HIP_STREAM_PARTS=$P python tools/stream_probe.py 10000 10 2>&1 | grep -E "stream pinned"
done
for U in 20 32 40; do
  echo "== tail units $U"; COATI_HIP_PIPE=stream COATI_HIP_STREAM_TAIL_UNITS=$U python tools/stream_probe.py 10000 10 2>&1 | grep -E "stream pinned"
done
done
