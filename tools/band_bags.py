"""Checkpoint band half width against kernel time and refills on the bench's bags: the synthetic headline batch, the bags
where 5 % / 25 % of the pairs carry a 90-300 nt indel, and the reference's bm_1k pair x 10 000.
usage: python tools/band_bags.py [band ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coati_amd import hip, host
from tests import util

bands = [int(x) for x in sys.argv[1:]] or [32, 48, 64, 80, 96, 128]
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 10000)
rng = np.random.default_rng(2024)


def with_indels(frac):  # (bench.py: band_sensitivity)
    a_parts, b_parts = [], []
    for p in range(10000):
        a = a_cat[int(a_off[p]):int(a_off[p + 1])]
        b = b_cat[int(b_off[p]):int(b_off[p + 1])]
        if rng.random() < frac:
            size = int(rng.integers(90, 301))
            at = int(rng.integers(0, max(1, len(b) - size)))
            if rng.random() < 0.5:
                b = np.concatenate([b[:at], b[at + size:]])
            else:
                b = np.concatenate([b[:at], rng.integers(0, 4, size).astype(np.uint8), b[at:]])[:1020]
        a_parts.append(a)
        b_parts.append(b)
    return hip.pack_pairs(list(zip(a_parts, b_parts)))


bags = {"headline": (a_cat, a_off, b_cat, b_off), "indel_5pct": with_indels(0.05), "indel_25pct": with_indels(0.25)}
a1, b1, case, doc = util.load_bench_pair("1k")  # (the reference's benchmark pair: 85 deletion and 46 insertion columns)
m_1k = hip.Model(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", doc["table"])),
                 host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)
bags["bm_1k_x10000"] = hip.pack_pairs([(a1, b1)] * 10000)
for name, enc in bags.items():
    model = m_1k if name.startswith("bm_") else model
    batch = hip.Batch(model, *enc)
    row = []
    for band in bands:
        model.set_option(hip.OPT_CK_BAND, band)
        ts = []
        for r in range(7):
            batch.viterbi_launch(); batch.sync()
            ts.append(batch.viterbi_timing()[0])
        row.append(f"{band}: {np.median(ts[2:]):.2f} ms / {batch.band_stats()[1]} refilled")
    print(f"{name:14s}", "   ".join(row), flush=True)
    batch.close()
