#!/bin/bash
# fused two-strip pairs (COATI_HIP_CK_FUSE, viterbi_ck.hip) against two items per pair over launch sizes, alternating, same box
# usage (GPU box, repo root): bash tools/experiments/r6_fuse_sizes.sh
for N in 4500 6000 8000 10000 20000 40000; do
  for rep in 1 2; do
    for F in 0 1; do
      echo -n "pairs $N fuse $F: "
      COATI_HIP_CK_FUSE=$F python tools/ab_fill.py --pairs $N --rounds 8 coati_amd/_build/libcoati_hip.so 2>&1 | tail -1
    done
  done
done
