#!/usr/bin/env python3
"""Build-container tool: fixtures for the reference's own benchmark suite
(/root/reference/benchmark/benchmark_main.cc.in:56-76, libcoati-benchmark-tests.txt:1-7): the seven
BM_marg_alignment inputs benchmark/data/benchmark_{156,1k,2k,4k,8k,16k,32k}.fasta, prepared as
marg_alignment prepares them (terminal stop codons trimmed, process_marginal / utils.cc:822-835).

tests/golden/benchmark_suite.npz: the prepared sequences, 2 bit/base (data).
tests/golden/benchmark_suite.json: per case fp32 score bits, alignment columns, CRC32 of the ops and the op counts
from the UNMODIFIED reference engine (oracle/_ref: viterbi_mem + traceback_viterbi; 32k needs 10.3 GB of fp32
matrices there), cross-checked against the low-memory oracle; plus the reference engine's wall time on the
machine that generated the fixture (informational).
"""
import json
import sys
import time
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import pyoracle as orc  # noqa: E402
from tests import util  # noqa: E402

REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"
G = np.float32(0.001)
E = np.float32(1.0) - np.float32(1.0) / np.float32(6.0)
table = np.load(OUT / "table_mg94_goldenP.npy")
consts = orc.gap_consts()


def read_fasta(path):
    seqs = []
    for line in Path(path).read_text().splitlines():
        if line.startswith(">"):
            seqs.append("")
        elif line and seqs:
            seqs[-1] += "".join(line.split())
    return seqs


def trim_stop(s):
    return s[:-3] if len(s) >= 3 and s[-3:] in ("TAA", "TAG", "TGA") else s


def pack2(s):
    v = np.array([util.NT.index(c) for c in s], np.uint8)
    v = np.concatenate([v, np.zeros((-len(v)) % 4, np.uint8)]).reshape(-1, 4)
    return (v[:, 0] | (v[:, 1] << 2) | (v[:, 2] << 4) | (v[:, 3] << 6)).astype(np.uint8)


def strings_to_ops(sa, sb):
    a = np.frombuffer(sa.encode(), np.uint8)
    b = np.frombuffer(sb.encode(), np.uint8)
    ops = np.zeros(len(a), np.uint8)
    ops[b == ord("-")] = 1
    ops[a == ord("-")] = 2
    return ops


arrays, meta = {}, []
for name in ("156", "1k", "2k", "4k", "8k", "16k", "32k"):
    anc, des = (s.upper() for s in read_fasta(REF / "benchmark" / "data" / f"benchmark_{name}.fasta"))
    anc, des = trim_stop(anc), trim_stop(des)
    assert len(anc) % 3 == 0 and set(anc) <= set("ACGT") and set(des) <= set("ACGT"), name
    a, b = util.encode_anc(anc), util.encode_des(des)
    t0 = time.time()
    _, _, _, sa, sb, rsc = orc.ref_viterbi(table, G, E, 1, anc, des, a, b, want_matrices=False)
    t_ref = time.time() - t0
    rops = strings_to_ops(sa, sb)
    ops, sc = orc.viterbi(table, consts, 1, a, b, lowmem=True)
    assert np.float32(rsc).view(np.uint32) == np.float32(sc).view(np.uint32), (name, rsc, sc)
    assert np.array_equal(rops, ops), name
    meta.append({"key": name, "name": f"benchmark/data/benchmark_{name}.fasta after stop trimming", "len_a": len(anc), "len_b": len(des),
                 "score_bits": "%08x" % int(np.float32(rsc).view(np.uint32)), "score": float(rsc), "columns": int(len(rops)),
                 "ops_crc32": "%08x" % zlib.crc32(rops.tobytes()), "n_match": int((rops == 0).sum()), "n_del": int((rops == 1).sum()),
                 "n_ins": int((rops == 2).sum()), "reference_engine_seconds_fixture_machine": round(t_ref, 4),
                 "source": "reference engine (oracle/_ref); low-memory oracle identical"})
    arrays[f"anc_{name}"] = pack2(anc)
    arrays[f"des_{name}"] = pack2(des)
    print(f"{name}: {len(anc)} x {len(des)}  reference {t_ref:.2f}s", flush=True)

np.savez_compressed(OUT / "benchmark_suite.npz", **arrays)
(OUT / "benchmark_suite.json").write_text(json.dumps({"gap_open": float(G), "gap_extend": float(E), "gap_len": 1,
                                                      "table": "table_mg94_goldenP.npy", "cases": meta}, indent=1))
print("wrote", OUT / "benchmark_suite.npz", (OUT / "benchmark_suite.npz").stat().st_size)
