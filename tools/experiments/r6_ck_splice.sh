#!/bin/bash
# round 6: viterbi_ck's spliced traceback for multi-strip pairs -- the four levels on the reference's mid-size batches
mkdir -p gpurun_out/r6ck
for lv in 0 1 nobridge miss; do
  echo "== COATI_HIP_CK_SPLICE=$lv"
  COATI_HIP_CK_SPLICE=$lv python tools/mid_batches.py 64 8k 16k 32k
  COATI_HIP_CK_SPLICE=$lv COATI_HIP_VITERBI_CK=1 COATI_HIP_STRIP_W=8 python tools/mid_batches.py 64 4k 8k
  COATI_HIP_CK_SPLICE=$lv COATI_HIP_CK_DEBUG=2 python tools/mid_batches.py 64 16k 2>&1 | grep -v "^$" | tail -4
done
echo "== fill only (COATI_HIP_CK_DEBUG=1; results not checked)"
COATI_HIP_CK_DEBUG=1 python tools/mid_batches.py 64 8k 16k 32k | sed 's/WRONG/(fill only)/g'
COATI_HIP_CK_DEBUG=1 COATI_HIP_VITERBI_CK=1 COATI_HIP_STRIP_W=8 python tools/mid_batches.py 64 4k 8k | sed 's/WRONG/(fill only)/g'
