"""The CPU oracle against the committed golden vectors (generated from the
compiled reference by tools/make_golden.py).  Runs anywhere."""
import json
from pathlib import Path

import numpy as np
import pytest

from tests import util

GOLD = Path(__file__).resolve().parent / "golden"


def f32(hexbits):
    return np.array([int(hexbits, 16)], np.uint32).view(np.float32)[0]


def bits(x):
    return int(np.float32(x).view(np.uint32))


@pytest.fixture(scope="module")
def table():
    return np.load(GOLD / "table_mg94_goldenP.npy")


def test_viterbi_cases(oracle, table):
    doc = json.loads((GOLD / "viterbi_cases.json").read_text())
    consts = oracle.gap_consts(doc["gap_open"], doc["gap_extend"])
    assert len(doc["cases"]) > 200
    for c in doc["cases"]:
        a, b = util.encode_anc(c["anc"]), util.encode_des(c["des"])
        L = c["gap_len"]
        big = (len(a) + L) * (len(b) + L) > 3_000_000
        ops, score = oracle.viterbi(table, consts, L, a, b, lowmem=big)
        assert oracle.ops_to_strings(ops, c["anc"], c["des"]) == (c["aln_anc"], c["aln_des"]), c["name"]
        assert bits(score) == int(c["score_bits"], 16), c["name"]
        if "M_bits" in c:
            M, D, I = oracle.fill(oracle.TROPICAL, table, consts, L, a, b)
            for mat, key in ((M, "M_bits"), (D, "D_bits"), (I, "I_bits")):
                want = np.array([int(h, 16) for h in c[key]], np.uint32)
                assert (mat.ravel().view(np.uint32) == want).all(), (c["name"], key)


def test_reference_example_scores(oracle, table):
    """Oracle outputs quoted in SURVEY.md Appendix A for the reference's sample data."""
    consts = oracle.gap_consts()
    for anc, des, L, aln, score in (("CTCTGGATAGTG", "CTATAGTG", 1, "CT----ATAGTG", 1.50913),
                                    ("GCGATTGCTGTT", "GCGACTGTT", 1, "GCGA---CTGTT", 3.79779),
                                    ("ACGTTAAGGGGT", "ACGAAT", 1, "ACG--AA----T", -10.1664),
                                    ("ACGTTAAGGGGT", "ACGAAT", 3, "AC------GAAT", -14.0144),
                                    ("CTCTGGATAGTG", "CTATAGTR", 1, "CT----ATAGTR", 1.51577)):
        ops, sc = oracle.viterbi(table, consts, L, util.encode_anc(anc), util.encode_des(des))
        assert oracle.ops_to_strings(ops, anc, des) == (anc, aln)
        assert sc == pytest.approx(score, rel=1e-5)


def test_forward_and_sample_cases(oracle, table):
    doc = json.loads((GOLD / "sample_cases.json").read_text())
    consts = oracle.gap_consts(doc["gap_open"], doc["gap_extend"])
    for c in doc["cases"]:
        a, b = util.encode_anc(c["anc"]), util.encode_des(c["des"])
        L = c["gap_len"]
        M, D, I = oracle.fill(oracle.LOG, table, consts, L, a, b)
        assert bits(M[-1, -1]) == int(c["final_M_bits"], 16), c["name"]
        assert bits(D[-1, -1]) == int(c["final_D_bits"], 16), c["name"]
        assert bits(I[-1, -1]) == int(c["final_I_bits"], 16), c["name"]
        rng = oracle.rng_seed(c["seeds"])
        for k, s in enumerate(c["samples"]):
            ops, sc = oracle.sampleback_mdi(M, D, I, table, consts, L, a, b, rng)
            assert oracle.ops_to_strings(ops, c["anc"], c["des"]) == (s["anc"], s["des"]), (c["name"], k)
            assert bits(sc) == int(s["score_bits"], 16), (c["name"], k)


def test_marg_sample_known_answers(oracle, table):
    """Reference doctest marg_sample (align_marginal.cc:653-672): exact strings; the
    17-digit scores depend on Eigen's fp32 expm rounding, so 1e-6 relative here."""
    known = json.loads((GOLD / "reference_known_answers.json").read_text())["marg_sample"]
    consts = oracle.gap_consts()
    for case in known:
        anc, des = case["seqs"]
        a, b = util.encode_anc(anc), util.encode_des(des)
        M, D, I = oracle.fill(oracle.LOG, table, consts, 1, a, b)
        rng = oracle.rng_seed(["42"])
        for (want_a, want_b), want_s in zip(case["out"], case["scores"]):
            ops, sc = oracle.sampleback_mdi(M, D, I, table, consts, 1, a, b, rng)
            assert oracle.ops_to_strings(ops, anc, des) == (want_a, want_b)
            assert float(sc) == pytest.approx(float(want_s), rel=1e-6)


def test_rng_streams(oracle):
    for entry in json.loads((GOLD / "rng_streams.json").read_text()):
        r = oracle.rng_seed(entry["seeds"])
        got = [bits(oracle.rng_f24(r)) for _ in entry["f24_bits"]]
        assert got == [int(h, 16) for h in entry["f24_bits"]], entry["seeds"]


def test_flag_walk_equals_matrix_walk(oracle, table):
    rng = np.random.default_rng(5)
    consts = oracle.gap_consts()
    for L in (1, 3):
        for anc, des in util.make_pairs(rng, 40, 1, 50, L=L):
            a, b = util.encode_anc(anc), util.encode_des(des)
            o1, s1 = oracle.viterbi(table, consts, L, a, b)
            o2, s2 = oracle.viterbi(table, consts, L, a, b, lowmem=True)
            assert (o1 == o2).all() and bits(s1) == bits(s2)


@pytest.mark.parametrize("key", ["10k", "20k"])
def test_long_sample_pairs_lowmem_oracle(oracle, key):
    """Sanitised sampledata/example-{10k,20k}.fasta: the low-memory oracle reproduces what the
    compiled reference produced (score bits, column count, CRC32 of the ops)."""
    import zlib

    a, b, case, doc = util.load_long_pair(key)
    assert "reference engine" in case["source"]
    table = np.load(GOLD / doc["table"])
    consts = oracle.gap_consts(doc["gap_open"], doc["gap_extend"])
    ops, sc = oracle.viterbi(table, consts, 1, a, b, lowmem=True)
    assert bits(sc) == int(case["score_bits"], 16)
    assert len(ops) == case["columns"]
    assert "%08x" % zlib.crc32(ops.tobytes()) == case["ops_crc32"]
    # and the O(n) path re-scoring used for pairs too large for any oracle
    assert bits(oracle.path_score(table, consts, 1, a, b, ops)) == int(case["score_bits"], 16)
