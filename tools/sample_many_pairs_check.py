"""Exact-stream sampling with MANY pairs (a share of the candidate array is a few dozen candidates, pairs wait for later
rounds): the device rounds against the host rounds and the one-walker-per-pair loop, bit for bit.
usage: python tools/sample_many_pairs_check.py [n_pairs] [n_samples]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coati_amd import hip, host
from tests import util

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000  # (above 1 024 pairs the library walks sequentially whatever the switches say)
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(11)
pairs = util.make_pairs(rng, n, 10, 60, L=1)  # 30 .. 180 nt
enc = util.encode_pairs(pairs)
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *hip.pack_pairs(enc))
batch.forward_launch()
states = np.stack([host.rng_seed(["7", str(p)]) for p in range(len(enc))])
outs = {}
for name, env in (("device rounds", {}), ("device rounds, 4 096 candidates", {"COATI_HIP_SPEC_CANDS": "4096"}), ("host rounds", {"COATI_HIP_SPEC_HOST_ROUNDS": "1"}),
                  ("sequential", {"COATI_HIP_SAMPLE_SEQUENTIAL": "1"})):
    for k in ("COATI_HIP_SPEC_CANDS", "COATI_HIP_SPEC_HOST_ROUNDS", "COATI_HIP_SAMPLE_SEQUENTIAL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    lw, ops, off, ln, st = batch.sampleback(ns, states, independent=False)
    dt = time.perf_counter() - t0
    crc = 0
    for p in range(0, len(enc), 7):
        for s in range(ns):
            crc = zlib.crc32(ops[int(off[p, s]):int(off[p, s]) + int(ln[p, s])].tobytes(), crc)
    outs[name] = (crc, zlib.crc32(lw.tobytes()), int(ln.sum()), zlib.crc32(st.tobytes()))
    print(f"{name}: {dt * 1e3:.1f} ms  {outs[name]}", flush=True)
assert len(set(outs.values())) == 1, outs
print(f"ok: {n} pairs x {ns} samples identical in all variants")
