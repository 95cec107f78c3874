import sys, numpy as np
sys.path.insert(0, '.')
from coati_amd import hip
from oracle import pyoracle as orc
from tests import util
rng = np.random.default_rng(3)
table = util.random_table(rng); consts = orc.gap_consts()
anc = util.random_anc(rng, 2); des = "".join(rng.choice(list("ACGT"), 40))
a, b = util.encode_anc(anc), util.encode_des(des)
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, *hip.pack_pairs([(a, b)]))
batch.viterbi_launch(); batch.sync()
got = batch.debug_flags(0)
M, D, I = orc.fill(0, table, consts, 1, a, b)
want = orc.tb_flags(M, D, I, consts)[1:, 1:]
np.set_printoptions(linewidth=250)
print('got'); print(got)
print('want'); print(want)
print('xor'); print(got ^ want)
