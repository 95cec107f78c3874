#!/usr/bin/env python3
"""Build-container tool: expected results of the FULL BASELINE configs[1] workload (10 000 synthetic
1 kb pairs, mar-mg94) and of a 2 000-pair slice of the configs[4] workload (the same generator,
mar-ecm), reduced to checksums: a CRC32 over every pair's ops in pair order, a CRC32 over the fp32
score bits, the total number of alignment columns.

Source of truth: the UNMODIFIED reference engine (oracle/_ref: viterbi_mem + traceback_viterbi of
/root/reference/src/lib/align_pair.cc).  The oracle port computes the same workload and must give the
same checksums (asserted here): the committed files are therefore pinned to the reference directly,
and the port -- which the GPU tests use at other sizes -- is pinned once more at full size.

    python tools/make_golden_synth.py        # writes tests/golden/synth10k_checksums.json, synth_ecm2k_checksums.json
"""
import json
import os
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import host  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

assert orc.ref_available(), "oracle/_ref is not built (make ref; needs /root/reference)"
G, E = 0.001, float(np.float32(1.0) - np.float32(1.0) / np.float32(6.0))  # the CLI defaults (gap open, extend)
THREADS = min(8, os.cpu_count() or 1)


def ops_of_strings(sa: str, sb: str) -> np.ndarray:
    """Alignment columns -> op bytes (coati_hip.h: 0 match, 1 deletion = gap in descendant, 2 insertion)."""
    a = np.frombuffer(sa.encode(), np.uint8)
    b = np.frombuffer(sb.encode(), np.uint8)
    ops = np.zeros(len(a), np.uint8)
    ops[b == ord("-")] = 1
    ops[a == ord("-")] = 2
    return ops


def checksums(model: str, n: int):
    table, consts = host.set_subst(model), host.gap_consts()
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n)

    def one(p):
        a = a_cat[a_off[p]:a_off[p + 1]]
        b = b_cat[b_off[p]:b_off[p + 1]]
        anc, des = host.synth_raw(p)
        _, _, _, sa, sb, sc_ref = orc.ref_viterbi(table, G, E, 1, anc, des, a, b, want_matrices=False)
        ops_port, sc_port = orc.viterbi(table, consts, 1, a, b, lowmem=True)
        ops_ref = ops_of_strings(sa, sb)
        assert len(ops_ref) == len(ops_port) and (ops_ref == ops_port).all(), (model, p)
        assert np.float32(sc_ref).view(np.uint32) == np.float32(sc_port).view(np.uint32), (model, p, sc_ref, sc_port)
        return ops_ref, sc_ref

    with ThreadPoolExecutor(THREADS) as ex:
        res = list(ex.map(one, range(n)))
    crc_ops, total = 0, 0
    for ops, _ in res:
        crc_ops = zlib.crc32(ops.tobytes(), crc_ops)
        total += len(ops)
    scores = np.array([sc for _, sc in res], np.float32)
    return {"pairs": n, "model": f"{model} defaults (table built by coati_amd/host set_subst)", "seed_base": "0xC0A71",
            "source": "oracle/_ref (the unmodified reference engine); the oracle port gave identical ops and score bits for every pair",
            "table_crc32": "%08x" % zlib.crc32(np.ascontiguousarray(table).tobytes()),
            "ops_crc32": "%08x" % crc_ops, "scores_crc32": "%08x" % zlib.crc32(scores.tobytes()), "columns": total,
            "score_sum_f64": float(np.float64(scores).sum())}


for model, n, name in (("mar-mg", 10000, "synth10k_checksums.json"), ("mar-ecm", 2000, "synth_ecm2k_checksums.json")):
    doc = checksums(model, n)
    (ROOT / "tests" / "golden" / name).write_text(json.dumps(doc, indent=1))
    print(name, doc)
