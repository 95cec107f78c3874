#!/usr/bin/env python3
"""Debug aid: per-cell decision bits of the GPU fill against the oracle's decision bytes for a few
small pairs (this is the script that exposed the SLP-vectoriser v_pk_add_f32 miscompute)."""
import sys, numpy as np
sys.path.insert(0, '.')
from coati_amd import hip
from oracle import pyoracle as orc
from tests import util
rng = np.random.default_rng(11)
table = util.random_table(rng); consts = orc.gap_consts()
pairs = util.make_pairs(rng, 6, 1, 30)
enc = util.encode_pairs(pairs)
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, *hip.pack_pairs(enc))
batch.viterbi_launch()
scores, ops, ops_off, ops_len = batch.viterbi_fetch()
for p,(a,b) in enumerate(enc):
    if len(a)*len(b)==0: continue
    M,D,I = orc.fill(0, table, consts, 1, a, b)
    want = orc.tb_flags(M,D,I,consts)[1:,1:]
    got = batch.debug_flags(p)
    diff = got != want
    print('pair',p,len(a),len(b),'ndiff',diff.sum(), 'of', diff.size)
    for plane,(sh,mask) in enumerate([(0,3),(2,3),(4,1)]):
        d = ((got>>sh)&mask) != ((want>>sh)&mask)
        print('  field',plane,'diff',d.sum(), 'first', np.argwhere(d)[:6].tolist())
    if diff.sum():
        ij = np.argwhere(diff)[0]; print('  got',got[ij[0],ij[1]],'want',want[ij[0],ij[1]])
        print(got[:4,:20]); print(want[:4,:20])
