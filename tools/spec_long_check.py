#!/usr/bin/env python3
"""One-off robustness check of the speculative exact-stream sampler on a divergent pair (the
sanitised 10 kb sample pair: ~12.6 k draws per sample, sigma ~60): run once as is and once with
COATI_HIP_SAMPLE_SEQUENTIAL=1; the two JSON lines must agree in ops/lw/st."""
import sys, os, zlib, json, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from tests import util
a, b, case, doc = util.load_long_pair("10k")
table = np.load(ROOT / 'tests' / 'golden' / doc["table"])
model = hip.Model(table, host.gap_consts(), 1)
batch = hip.Batch(model, *hip.pack_pairs([(a, b)]))
batch.forward_launch(); batch.sync()
st = np.stack([host.rng_seed(["7"])])
t0 = time.perf_counter()
lw, ops, off, ln, so = batch.sampleback(200, st, independent=False)
dt = time.perf_counter() - t0
crc = 0
for s in range(200):
    crc = zlib.crc32(ops[int(off[0, s]):int(off[0, s]) + int(ln[0, s])].tobytes(), crc)
print(json.dumps({"seq": "COATI_HIP_SAMPLE_SEQUENTIAL" in os.environ, "ms": dt * 1e3, "ops": crc, "lw": zlib.crc32(lw.tobytes()), "st": so.tolist(), "len_mean": float(ln.mean()), "len_std": float(ln.std())}))
