"""HIP Viterbi path against the committed golden vectors of the compiled
reference (tests/golden/viterbi_cases.json), through the C ABI, with the table
the C++ host layer builds -- i.e. the whole product path."""
import json
from pathlib import Path

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def test_golden_viterbi_cases_gap_len_1():
    from coati_amd import hip, host

    doc = json.loads((GOLD / "viterbi_cases.json").read_text())
    table = np.load(GOLD / "table_mg94_goldenP.npy")
    consts = host.gap_consts(doc["gap_open"], doc["gap_extend"])
    cases = [c for c in doc["cases"] if c["gap_len"] == 1]
    assert len(cases) > 150
    enc = [host.encode(c["anc"], c["des"]) for c in cases]
    model = hip.Model(table, consts, 1)
    scores, ops, ops_off, ops_len = model.viterbi(*hip.pack_pairs(enc))
    for p, c in enumerate(cases):
        got = ops[int(ops_off[p]):int(ops_off[p]) + int(ops_len[p])]
        sa = "".join("-" if o == 2 else ch for o, ch in zip(got, _expand(got, c["anc"], 2)))
        sb = "".join("-" if o == 1 else ch for o, ch in zip(got, _expand(got, c["des"], 1)))
        assert (sa, sb) == (c["aln_anc"], c["aln_des"]), c["name"]
        assert int(np.float32(scores[p]).view(np.uint32)) == int(c["score_bits"], 16), c["name"]


def test_golden_viterbi_cases_gap_len_3():
    """The 51 gap_len == 3 cases of the compiled reference (`coati alignpair -k 3`; fixture from
    oracle/_ref, tools/make_golden.py): aligned strings and score bits through the planner's kernel
    (viterbi_k; dp_generic when COATI_HIP_FORCE_GENERIC is set -- see the next test)."""
    from coati_amd import hip, host

    doc = json.loads((GOLD / "viterbi_cases.json").read_text())
    table = np.load(GOLD / "table_mg94_goldenP.npy")
    consts = host.gap_consts(doc["gap_open"], doc["gap_extend"])
    cases = [c for c in doc["cases"] if c["gap_len"] == 3]
    assert len(cases) >= 50
    enc = [host.encode(c["anc"], c["des"]) for c in cases]
    model = hip.Model(table, consts, 3)
    scores, ops, ops_off, ops_len = model.viterbi(*hip.pack_pairs(enc))
    for p, c in enumerate(cases):
        got = ops[int(ops_off[p]):int(ops_off[p]) + int(ops_len[p])]
        sa = "".join("-" if o == 2 else ch for o, ch in zip(got, _expand(got, c["anc"], 2)))
        sb = "".join("-" if o == 1 else ch for o, ch in zip(got, _expand(got, c["des"], 1)))
        assert (sa, sb) == (c["aln_anc"], c["aln_des"]), c["name"]
        assert int(np.float32(scores[p]).view(np.uint32)) == int(c["score_bits"], 16), c["name"]


def test_golden_viterbi_cases_gap_len_3_generic_kernel():
    """The same fixtures through dp_generic (the library reads COATI_HIP_FORCE_GENERIC once: child process)."""
    import os
    import subprocess
    import sys

    if os.environ.get("COATI_HIP_FORCE_GENERIC"):
        pytest.skip("already inside the forced run")
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, COATI_HIP_FORCE_GENERIC="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", str(Path(__file__).resolve()),
                          "-k", "test_golden_viterbi_cases_gap_len_3 and not generic"], env=env, capture_output=True,
                         text=True, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:]


def _expand(ops, seq, gap_op):
    """Characters of `seq` laid out along the alignment columns (placeholder where the op is a gap in seq)."""
    it = iter(seq)
    return [(" " if o == gap_op else next(it)) for o in ops]


def test_reference_doctest_alignments_default_model():
    """marg_alignment known answers (align_marginal.cc:149-240), model built by the host layer."""
    from coati_amd import hip, host

    known = json.loads((GOLD / "reference_known_answers.json").read_text())["marg_alignment"]
    for case in known:
        if case.get("gap_len", 1) != 1:
            continue
        table = host.set_subst(case["model"], amb_best=case.get("amb") == "BEST")
        seqs = list(case["seqs"])
        names = case.get("names", ["1", "2"])
        if case.get("rev") or (case.get("refs") and case["refs"] == names[1]):
            seqs.reverse()
        a, b = host.encode(*seqs)
        model = hip.Model(table, host.gap_consts(), 1)
        scores, ops, ops_off, ops_len = model.viterbi(*hip.pack_pairs([(a, b)]))
        got = ops[int(ops_off[0]):int(ops_off[0]) + int(ops_len[0])]
        sa = "".join("-" if o == 2 else ch for o, ch in zip(got, _expand(got, seqs[0], 2)))
        sb = "".join("-" if o == 1 else ch for o, ch in zip(got, _expand(got, seqs[1], 1)))
        assert [sa, sb] == case["out"], case


@pytest.mark.parametrize("kernel", ["auto", "ck", "lp4", "lp3", "nosplice", "miss"])
@pytest.mark.parametrize("key", ["10k", "20k", "40k", "80k", "160k"])
def test_long_sample_pairs(key, kernel, monkeypatch):
    """BASELINE configs[2]: the reference's long sample pairs (sanitised)
    up to 160 002 x 160 002 nt.  Their strips run on different wavefronts, pipelined through HBM
    boundary columns.  Expected score bits / columns / CRC32(ops) come from the compiled reference
    (10k-40k) and from the low-memory oracle those pin (80k, 160k): bit-exact."""
    import zlib

    from coati_amd import hip, host

    # (a lone long pair is the planner's case for viterbi_l1; "ck" forces the checkpoint kernel through the
    # same narrow pipelined strips)
    monkeypatch.delenv("COATI_HIP_VITERBI_BITS", raising=False)
    monkeypatch.delenv("COATI_HIP_STRIP_W", raising=False)
    if kernel == "ck":
        monkeypatch.setenv("COATI_HIP_VITERBI_CK", "1")
    else:
        monkeypatch.delenv("COATI_HIP_VITERBI_CK", raising=False)
    # (round 6) viterbi_lp's shapes forced: 4 columns per lane (the round-5 plan for the 160 kb pair) and 3 (per-column
    # decision words; the planner's choice for the 160 kb pair only -- the shorter ones take 2 columns)
    if kernel in ("lp4", "lp3"):
        monkeypatch.setenv("COATI_HIP_STRIP_W", kernel[2])
    # (round 6) the planner's shape with the spliced traceback off, and with strip records that never match
    monkeypatch.delenv("COATI_HIP_LP_SPLICE", raising=False)
    if kernel in ("nosplice", "miss"):
        monkeypatch.setenv("COATI_HIP_LP_SPLICE", "0" if kernel == "nosplice" else "miss")
    a, b, case, doc = util.load_long_pair(key)
    table = np.load(GOLD / doc["table"])
    consts = host.gap_consts(doc["gap_open"], doc["gap_extend"])
    model = hip.Model(table, consts, 1)
    scores, ops, ops_off, ops_len = model.viterbi(*hip.pack_pairs([(a, b)]))
    got = ops[int(ops_off[0]):int(ops_off[0]) + int(ops_len[0])]
    assert int(np.float32(scores[0]).view(np.uint32)) == int(case["score_bits"], 16)
    assert len(got) == case["columns"]
    assert (int((got == 0).sum()), int((got == 1).sum()), int((got == 2).sum())) == (case["n_match"], case["n_del"],
                                                                                     case["n_ins"])
    assert "%08x" % zlib.crc32(got.tobytes()) == case["ops_crc32"]


@pytest.mark.parametrize("kernel", ["auto", "ck", "bits"])
def test_reference_benchmark_suite(kernel, monkeypatch):
    """The reference's own benchmark inputs (benchmark/benchmark_main.cc.in:56-76: bm_156 ... bm_32k, seven
    marginal alignments up to 29 394 x 29 295 nt): score bits, columns, op counts and CRC32 of the ops from the
    unmodified reference engine (tests/golden/benchmark_suite.json) -- each pair alone (a lone long pair: narrowed
    strips pipelined over wavefronts) and all seven in ONE batch with four copies each (multi-strip pairs of very
    different sizes sharing the queue), under the planner's kernel and under each gap_len-1 kernel forced."""
    import zlib

    from coati_amd import hip, host

    monkeypatch.delenv("COATI_HIP_VITERBI_BITS", raising=False)
    monkeypatch.delenv("COATI_HIP_VITERBI_CK", raising=False)
    if kernel == "ck":
        monkeypatch.setenv("COATI_HIP_VITERBI_CK", "1")
    elif kernel == "bits":
        monkeypatch.setenv("COATI_HIP_VITERBI_BITS", "1")
    keys = ["156", "1k", "2k", "4k", "8k", "16k", "32k"]
    loaded = {k: util.load_bench_pair(k) for k in keys}
    doc = loaded["156"][3]
    model = hip.Model(np.load(GOLD / doc["table"]), host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)

    def check(case, score, got):
        assert int(np.float32(score).view(np.uint32)) == int(case["score_bits"], 16), case["key"]
        assert len(got) == case["columns"], case["key"]
        assert (int((got == 0).sum()), int((got == 1).sum()), int((got == 2).sum())) == (case["n_match"], case["n_del"], case["n_ins"])
        assert "%08x" % zlib.crc32(got.tobytes()) == case["ops_crc32"], case["key"]

    for k in keys:
        a, b, case, _ = loaded[k]
        scores, ops, off, ln = model.viterbi(*hip.pack_pairs([(a, b)]))
        check(case, scores[0], ops[int(off[0]):int(off[0]) + int(ln[0])])
    order = [k for k in keys for _ in range(4)]
    scores, ops, off, ln = model.viterbi(*hip.pack_pairs([(loaded[k][0], loaded[k][1]) for k in order]))
    for p, k in enumerate(order):
        check(loaded[k][2], scores[p], ops[int(off[p]):int(off[p]) + int(ln[p])])
    model.close()


def test_full_baseline_workload_checksums():
    """BASELINE configs[1] at FULL size: all 10 000 synthetic 1 kb pairs through the C ABI; the CRC32
    of every op of every pair, the CRC32 of the fp32 score bits and the column total equal the CPU
    oracle's (tools/make_golden_synth.py).  Bit-exact or it fails."""
    import zlib

    from coati_amd import hip, host

    want = json.loads((GOLD / "synth10k_checksums.json").read_text())
    table = host.set_subst("mar-mg")
    assert "%08x" % zlib.crc32(np.ascontiguousarray(table).tobytes()) == want["table_crc32"], "host model drifted"
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, want["pairs"])
    model = hip.Model(table, host.gap_consts(), 1)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    for _ in range(2):  # a relaunch on the same batch gives the same answer
        batch.viterbi_launch()
    scores, ops, off, ln = batch.viterbi_fetch()
    assert "%08x" % zlib.crc32(scores.tobytes()) == want["scores_crc32"]
    assert int(ln.sum()) == want["columns"]
    crc = 0
    for p in range(want["pairs"]):
        crc = zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes(), crc)
    assert "%08x" % crc == want["ops_crc32"]


def test_config4_mar_ecm_workload_checksums():
    """BASELINE configs[4]'s workload (synthetic 1 kb pairs, mar-ecm) on one GPU: 2 000 pairs through the
    resident batch AND through the pipelined one-shot call; CRC32 of every op of every pair, of the fp32
    score bits and the column total equal those of the unmodified reference engine
    (tools/make_golden_synth.py -> tests/golden/synth_ecm2k_checksums.json).  Also pins host::ecm_p /
    set_subst("mar-ecm"): the table checksum must be the one the fixture was generated with."""
    import zlib

    from coati_amd import hip, host

    want = json.loads((GOLD / "synth_ecm2k_checksums.json").read_text())
    table = host.set_subst("mar-ecm")
    assert "%08x" % zlib.crc32(np.ascontiguousarray(table).tobytes()) == want["table_crc32"], "host ECM model drifted"
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, want["pairs"])
    model = hip.Model(table, host.gap_consts(), 1)

    def check(scores, ops, off, ln):
        assert "%08x" % zlib.crc32(np.ascontiguousarray(scores).tobytes()) == want["scores_crc32"]
        assert int(ln.sum()) == want["columns"]
        crc = 0
        for p in range(want["pairs"]):
            crc = zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes(), crc)
        assert "%08x" % crc == want["ops_crc32"]

    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    batch.viterbi_launch()
    check(*batch.viterbi_fetch())
    batch.close()
    check(*model.viterbi(a_cat, a_off, b_cat, b_off))
    model.close()
