set -u
O=gpurun_out/r5spec2; mkdir -p $O
for C in 131072 196608 262144 393216 524288; do
 for Z in 2.0; do
  echo -n "cands $C z $Z: " >> $O/out.txt
  COATI_HIP_SPEC_CANDS=$C COATI_HIP_SPEC_Z=$Z timeout 300 python3 tools/sample_bench.py --pairs 256 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sampleback ms %.3f  forward16 %.3f' % (d['sampleback_exact_stream']['ms'], d['forward_fill_16_pairs_ms']))" >> $O/out.txt
 done
done
for Z in 1.5 1.75 2.25 2.5; do
  echo -n "cands 196608 z $Z: " >> $O/out.txt
  COATI_HIP_SPEC_Z=$Z timeout 300 python3 tools/sample_bench.py --pairs 256 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sampleback ms %.3f  forward16 %.3f' % (d['sampleback_exact_stream']['ms'], d['forward_fill_16_pairs_ms']))" >> $O/out.txt
done
