"""A/B of the row-part split of the last pairs of a viterbi_ck launch (COATI_HIP_CK_SPLIT="pairs,parts", 0 = off):
each variant in its own process on the same resident pairs, results compared through a checksum.
usage: python tools/split_ab.py [n_pairs] [variants ...]     e.g.  split_ab.py 10000 0 3072,2 2048,4"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, zlib, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip, host
n = %d
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, *host.synth_encoded(0, n))
ts = []
for r in range(11):
    batch.viterbi_launch(); batch.sync()
    if r >= 2: ts.append(sum(batch.viterbi_timing()))
sc, ops, off, ln = batch.viterbi_fetch()
h = zlib.crc32(sc.tobytes())
for p in range(n):
    h = zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes(), h)
print(json.dumps({"ms_med": round(float(np.median(ts)), 3), "ms_min": round(float(min(ts)), 3), "gcups": round(batch.cells / float(np.median(ts)) / 1e6),
                  "crc": h, "nan": int(np.isnan(sc).sum()), "device_GB": round(batch.device_bytes / 1e9, 2)}))
'''
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
variants = sys.argv[2:] or ["0", "3072,2"]
for rnd in range(2):
    for v in variants:
        env = dict(os.environ, COATI_HIP_CK_SPLIT=v)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, n)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("split", v, line[-1] if line else ("FAILED " + r.stderr[-600:]), flush=True)
