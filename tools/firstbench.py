import sys, time, numpy as np
sys.path.insert(0, '.')
from coati_amd import hip
from oracle import pyoracle as orc
from tests import util
rng = np.random.default_rng(5)
table = util.random_table(rng); consts = orc.gap_consts()
base = []
for _ in range(64):
    a = util.random_anc(rng, 334); base.append((a, util.mutate(rng, a)))
enc = util.encode_pairs(base)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
enc = [enc[i % 64] for i in range(N)]
a_cat, a_off, b_cat, b_off = hip.pack_pairs(enc)
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
print('cells', batch.cells, 'device GB', batch.device_bytes / 1e9)
for it in range(5):
    t0 = time.time(); batch.viterbi_launch(); batch.sync(); t1 = time.time()
    f, w = batch.viterbi_timing()
    print(f'iter {it}: wall {1e3*(t1-t0):.2f} ms fill {f:.3f} ms walk {w:.3f} ms  GCUPS(fill) {batch.cells/f/1e6:.1f} GCUPS(total) {batch.cells/(f+w)/1e6:.1f}')
scores, ops, ops_off, ops_len = batch.viterbi_fetch()
for p in range(3):
    wo, ws = orc.viterbi(table, consts, 1, *enc[p])
    got = ops[int(ops_off[p]):int(ops_off[p]) + int(ops_len[p])]
    print(p, scores[p], ws, (got == wo).all() if len(got) == len(wo) else 'LEN')
