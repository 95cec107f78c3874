set -u
O=gpurun_out/r5spec; mkdir -p $O
for pass in 1 2; do
for C in 131072 196608 262144 393216 524288 786432; do
  for Z in 2.0 1.5; do
    echo "cands=$C z=$Z" >> $O/spec.txt
    COATI_HIP_SPEC_CANDS=$C COATI_HIP_SPEC_Z=$Z timeout 300 python3 tools/sample_bench.py --pairs 64 2>&1 | grep -o '"sampleback_exact_stream": {[^}]*}' >> $O/spec.txt
  done
done
done
