#!/usr/bin/env python3
"""Pairwise step of `coati msa` at scale: one 1 kb reference against N leaves with D distinct branch
lengths (host align_leafs: D tables, one batched launch), next to the per-leaf CPU port."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import host
from oracle import pyoracle as orc
from tests import util

n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_dist = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(3)
ref = util.random_anc(rng, 334)
leaves = [util.mutate(rng, ref, sub=0.02 + 0.1 * rng.random(), n_indel=2) for _ in range(n_leaves)]
br = [0.01 + 0.02 * int(k) for k in rng.integers(0, n_dist, n_leaves)]
host.align_leafs(ref, leaves[:4], br[:4])  # warm up (HIP runtime)
t0 = time.perf_counter()
res = host.align_leafs(ref, leaves, br)
t_gpu = time.perf_counter() - t0
cells = sum(len(ref) * len(x) for x in leaves)
# CPU port on a sample, one thread
consts = host.gap_consts()
n_cpu = 20
t0 = time.perf_counter()
for leaf, t in zip(leaves[:n_cpu], br[:n_cpu]):
    table = host.set_subst("mar-mg", br_len=t)
    orc.viterbi(table, consts, 1, util.encode_anc(ref), util.encode_des(leaf))
t_cpu = (time.perf_counter() - t0) / n_cpu * n_leaves
print(f"{n_leaves} leaves x 1 kb, {n_dist} branch lengths: align_leafs {t_gpu*1e3:.1f} ms end to end "
      f"({cells/t_gpu/1e9:.1f} GCUPS incl. host model/encode/strings); CPU port, 1 thread (extrapolated): {t_cpu:.1f} s")
