#!/usr/bin/env python3
"""Build-container tool: expected results of the FULL BASELINE configs[1] workload (10 000 synthetic
1 kb pairs, generator SURVEY.md 8(d) = coati_amd/host/synth.cc) from the CPU oracle, reduced to
checksums (a CRC32 over every pair's ops in pair order, a CRC32 over the fp32 score bits, the total
number of alignment columns), stored in tests/golden/synth10k_checksums.json.  The oracle is pinned
bit-for-bit against the compiled reference (tests/test_oracle_vs_ref.py); 16 of these pairs are
additionally stored in full, from the reference itself, in viterbi_cases.json."""
import json
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import host  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

N = 10000
table, consts = host.set_subst("mar-mg"), host.gap_consts()
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, N)


def one(p):
    a = a_cat[a_off[p]:a_off[p + 1]]
    b = b_cat[b_off[p]:b_off[p + 1]]
    ops, sc = orc.viterbi(table, consts, 1, a, b, lowmem=True)
    return ops, sc


with ThreadPoolExecutor(8) as ex:
    res = list(ex.map(one, range(N)))
crc_ops, total = 0, 0
for ops, _ in res:
    crc_ops = zlib.crc32(ops.tobytes(), crc_ops)
    total += len(ops)
scores = np.array([sc for _, sc in res], np.float32)
doc = {"pairs": N, "model": "mar-mg defaults (table built by coati_amd/host set_subst)", "seed_base": "0xC0A71",
       "table_crc32": "%08x" % zlib.crc32(np.ascontiguousarray(table).tobytes()),
       "ops_crc32": "%08x" % crc_ops, "scores_crc32": "%08x" % zlib.crc32(scores.tobytes()), "columns": total,
       "score_sum_f64": float(np.float64(scores).sum())}
(ROOT / "tests" / "golden" / "synth10k_checksums.json").write_text(json.dumps(doc, indent=1))
print(doc)
