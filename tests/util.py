"""Shared helpers for the tests: tiny pure-Python encoder and pair generators.

The encoder here is deliberately independent of the product's C++ host code
(coati_amd/host) so that it can be used to cross-check it.
"""
from __future__ import annotations

import numpy as np

NT = "ACGT"
STOPS64 = (48, 50, 56)  # TAA TAG TGA
SENSE64 = [c for c in range(64) if c not in STOPS64]
NT16 = "ACGTRYMKSWBDHVN"


def cod64_to_61(c: int) -> int:
    assert c not in STOPS64
    return c - sum(1 for s in STOPS64 if s < c)


def encode_anc(seq: str) -> np.ndarray:
    out = []
    for i in range(0, len(seq), 3):
        c = (NT.index(seq[i]) << 4) | (NT.index(seq[i + 1]) << 2) | NT.index(seq[i + 2])
        c61 = cod64_to_61(c)
        out += [c61 * 3, c61 * 3 + 1, c61 * 3 + 2]
    return np.array(out, np.uint8)


def encode_des(seq: str) -> np.ndarray:
    return np.array([NT16.index(ch) for ch in seq.upper().replace("U", "T")], np.uint8)


def codon_str(c64: int) -> str:
    return NT[(c64 >> 4) & 3] + NT[(c64 >> 2) & 3] + NT[c64 & 3]


def random_anc(rng, n_codons: int) -> str:
    return "".join(codon_str(int(c)) for c in rng.choice(SENSE64, n_codons))


def mutate(rng, anc: str, sub=0.05, n_indel=2, mean_len=6, amb=0.0) -> str:
    s = list(anc)
    for i in range(len(s)):
        if rng.random() < sub:
            s[i] = rng.choice([x for x in NT if x != s[i]])
    for _ in range(rng.poisson(n_indel)):
        ln = int(rng.geometric(1.0 / mean_len))
        pos = int(rng.integers(0, len(s) + 1))
        if rng.random() < 0.5:
            s[pos:pos] = list(rng.choice(list(NT), ln))
        else:
            del s[pos:pos + ln]
    if amb > 0:
        for i in range(len(s)):
            if rng.random() < amb:
                s[i] = rng.choice(list(NT16[4:]))
    return "".join(s)


def random_table(rng) -> np.ndarray:
    """A substitution table with realistic range; the DP does not care where it came from."""
    t = rng.uniform(-8.0, 2.0, size=(183, 15)).astype(np.float32)
    return np.ascontiguousarray(t)


def tie_table() -> np.ndarray:
    """Coarse table (few distinct values) to provoke exact ties in the recurrences."""
    rng = np.random.default_rng(7)
    return np.ascontiguousarray(rng.choice(np.array([-4.0, -2.0, 0.5, 1.0], np.float32), size=(183, 15)))


def make_pairs(rng, n, min_cod=1, max_cod=120, L=1, amb=0.02):
    """Mixed bag: related pairs, unrelated pairs, low complexity, extreme length ratios."""
    pairs = []
    for k in range(n):
        nc = int(rng.integers(min_cod, max_cod + 1))
        kind = k % 5
        if kind == 4:  # low complexity
            cod = codon_str(int(rng.choice(SENSE64)))
            anc = cod * nc
        else:
            anc = random_anc(rng, nc)
        if kind in (0, 1, 4):
            des = mutate(rng, anc, amb=amb if kind == 1 else 0.0)
        elif kind == 2:
            des = "".join(rng.choice(list(NT), int(rng.integers(0, 3 * max_cod))))
        else:
            des = mutate(rng, anc[: max(3, len(anc) // 4)])
        if L > 1:
            anc = anc[: len(anc) // (3 * L) * (3 * L)] if L % 3 else anc[: len(anc) // L * L]
            des = des[: len(des) // L * L]
        pairs.append((anc, des))
    return pairs


def encode_pairs(pairs):
    return [(encode_anc(a), encode_des(b)) for a, b in pairs]


def load_long_pair(key: str):
    """Encoded (a, b) of one sanitised long sample pair from tests/golden/long_pairs.npz (2 bit/base;
    see tools/make_golden_long.py) and its expectation record."""
    import json
    from pathlib import Path

    gold = Path(__file__).resolve().parent / "golden"
    doc = json.loads((gold / "long_pairs.json").read_text())
    case = next(c for c in doc["cases"] if c["key"] == key)
    z = np.load(gold / "long_pairs.npz")

    def unpack(v, n):
        nt = np.stack([v & 3, (v >> 2) & 3, (v >> 4) & 3, (v >> 6) & 3], axis=1).reshape(-1)[:n].astype(np.uint8)
        return nt

    anc_nt = unpack(z[f"anc_{key}"], case["len_a"])
    des_nt = unpack(z[f"des_{key}"], case["len_b"])
    # ancestor: codon61*3 + phase (utils.cc:496-528); descendant: nt16 code = 0..3 for ACGT
    cod = (anc_nt[0::3].astype(np.int32) << 4) | (anc_nt[1::3].astype(np.int32) << 2) | anc_nt[2::3]
    assert not np.isin(cod, STOPS64).any()
    c61 = cod - sum((cod > s).astype(np.int32) for s in STOPS64)
    a = (c61[:, None] * 3 + np.arange(3)[None, :]).reshape(-1).astype(np.uint8)
    return a, des_nt, case, doc


def load_bench_pair(key: str):
    """Encoded (a, b) of one input of the reference's benchmark suite (benchmark/data/benchmark_<key>.fasta, prepared
    as marg_alignment prepares it) from tests/golden/benchmark_suite.npz, its expectation record and the fixture
    document (tools/make_golden_bench.py)."""
    import json
    from pathlib import Path

    gold = Path(__file__).resolve().parent / "golden"
    doc = json.loads((gold / "benchmark_suite.json").read_text())
    case = next(c for c in doc["cases"] if c["key"] == key)
    z = np.load(gold / "benchmark_suite.npz")

    def unpack(v, n):
        return np.stack([v & 3, (v >> 2) & 3, (v >> 4) & 3, (v >> 6) & 3], axis=1).reshape(-1)[:n].astype(np.uint8)

    anc_nt = unpack(z[f"anc_{key}"], case["len_a"])
    des_nt = unpack(z[f"des_{key}"], case["len_b"])
    cod = (anc_nt[0::3].astype(np.int32) << 4) | (anc_nt[1::3].astype(np.int32) << 2) | anc_nt[2::3]
    assert not np.isin(cod, STOPS64).any()
    c61 = cod - sum((cod > s).astype(np.int32) for s in STOPS64)
    a = (c61[:, None] * 3 + np.arange(3)[None, :]).reshape(-1).astype(np.uint8)
    return a, des_nt, case, doc


def forward_exact() -> bool:
    """The Forward / sampling kernels run the bit-exact libm restatement unless the
    fast log-plus was asked for (COATI_HIP_FORWARD_FAST=1, 1e-5 relative)."""
    import os

    from coati_amd import hip

    if hip.DEFAULT_FORWARD_MODE is not None:  # (the mode the test process put its models into: tests/test_gpu_sample.py)
        return hip.DEFAULT_FORWARD_MODE == hip.FORWARD_EXACT
    v = os.environ.get("COATI_HIP_FORWARD_FAST", "")
    return v in ("", "0")


def same_bits(got, want) -> bool:
    got = np.ascontiguousarray(got, np.float32)
    want = np.ascontiguousarray(want, np.float32)
    return got.shape == want.shape and bool((got.view(np.uint32) == want.view(np.uint32)).all())
