#!/usr/bin/env python3
"""Build-container tool: long-pair fixtures (BASELINE configs[2]).

Inputs: sampledata/example-{10k,20k,40k,80k,160k}.fasta of the reference, SANITISED as
SURVEY.md §8(d) prescribes (the reference CLI rejects the raw files because the ancestor
holds in-frame stop codons, utils.cc:510-514): in the ancestor every in-frame
TAA->TAC, TAG->TAC, TGA->TGC; descendant unchanged; terminal stop of either trimmed as
process_marginal does.  The sanitised sequences are stored 2 bit/base in
tests/golden/long_pairs.npz (data, ~155 KB).

Expected outputs (tests/golden/long_pairs.json): fp32 score bits, number of alignment
columns, CRC32 of the ops (one byte per column, 0 M / 1 D / 2 I).  For 10k/20k/40k they
come from the UNMODIFIED reference engine (oracle/_ref: viterbi_mem + traceback_viterbi)
and the low-memory oracle is checked against them here; 80k and 160k need 77 GB / 307 GB
of fp32 matrices in the reference, so their expectations come from the low-memory oracle
that the three smaller files pin.
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


def sanitise(anc):
    fix = {"TAA": "TAC", "TAG": "TAC", "TGA": "TGC"}
    cod = [anc[i:i + 3] for i in range(0, len(anc), 3)]
    n = sum(c in fix for c in cod)
    return "".join(fix.get(c, c) for c in cod), n


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
for name in ("10k", "20k", "40k", "80k", "160k"):
    anc, des = (s.upper() for s in read_fasta(REF / "sampledata" / f"example-{name}.fasta"))
    anc, n_fixed = sanitise(anc)
    anc, des = trim_stop(anc), trim_stop(des)
    a, b = util.encode_anc(anc), util.encode_des(des)
    t0 = time.time()
    ops, sc = orc.viterbi(table, consts, 1, a, b, lowmem=True)
    t_low = time.time() - t0
    entry = {"name": f"example-{name}.fasta sanitised", "len_a": len(anc), "len_b": len(des), "stops_replaced": n_fixed,
             "score_bits": "%08x" % int(np.float32(sc).view(np.uint32)), "score": float(sc), "columns": int(len(ops)),
             "ops_crc32": "%08x" % zlib.crc32(ops.tobytes()), "n_match": int((ops == 0).sum()),
             "n_del": int((ops == 1).sum()), "n_ins": int((ops == 2).sum()), "source": "low-memory oracle"}
    if name in ("10k", "20k", "40k"):
        t0 = time.time()
        _, _, _, sa, sb, rsc = orc.ref_viterbi(table, G, E, 1, anc, des, a, b, want_matrices=False)
        rops = strings_to_ops(sa, sb)
        assert np.float32(rsc).view(np.uint32) == np.float32(sc).view(np.uint32), (name, rsc, sc)
        assert np.array_equal(rops, ops), name
        entry["source"] = "reference engine (oracle/_ref); low-memory oracle identical"
        print(f"{name}: reference {time.time()-t0:.1f}s == lowmem oracle {t_low:.1f}s", flush=True)
    else:
        print(f"{name}: lowmem oracle {t_low:.1f}s", flush=True)
    arrays[f"anc_{name}"] = pack2(anc)
    arrays[f"des_{name}"] = pack2(des)
    entry["key"] = name
    meta.append(entry)

np.savez_compressed(OUT / "long_pairs.npz", **arrays)
(OUT / "long_pairs.json").write_text(json.dumps({"gap_open": float(G), "gap_extend": float(E), "gap_len": 1,
                                                 "table": "table_mg94_goldenP.npy", "cases": meta}, indent=1))
print("wrote", OUT / "long_pairs.npz", (OUT / "long_pairs.npz").stat().st_size)
