"""The general DP kernel (dp_generic): Viterbi for gap unit lengths > 1 (bit-exact
vs the oracle), the same kernel forced on gap_len 1 as a cross-check of the
hand-scheduled viterbi_l1, and the Forward fill (log semiring; device expf/log1pf
differ from glibc by ulps, so: 1e-5 relative -- the tolerance north_star states)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def viterbi_compare(hip, oracle, table, consts, L, pairs, check_flags=True):
    enc = util.encode_pairs(pairs)
    model = hip.Model(table, consts, L)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.viterbi_launch()
    scores, ops, ops_off, ops_len = batch.viterbi_fetch()
    for p, (a, b) in enumerate(enc):
        want_ops, want_score = oracle.viterbi(table, consts, L, a, b)
        got = ops[int(ops_off[p]):int(ops_off[p]) + int(ops_len[p])]
        assert bits(scores[p]) == bits(want_score), (L, p, len(a), len(b), scores[p], want_score)
        assert len(got) == len(want_ops) and (got == want_ops).all(), (L, p, len(a), len(b))
        if check_flags and 0 < len(a) * len(b) <= 200_000:
            M, D, I = oracle.fill(oracle.TROPICAL, table, consts, L, a, b)
            want = oracle.tb_flags(M, D, I, consts)[L:, L:].copy()
            got_f = batch.debug_flags(p)
            want[-1, -1] = got_f[-1, -1]  # oracle's last cell is terminal-adjusted
            # gap_len 2 and 3 store the live cells only ((i - j) % L == 0; the others are `lowest` in the
            # reference and can never be on a path): the kernel reports 0xff for the rest
            live = got_f != 0xFF
            if L in (2, 3) and "COATI_HIP_FORCE_GENERIC" not in os.environ:
                ii, jj = np.meshgrid(np.arange(len(a)), np.arange(len(b)), indexing="ij")
                assert (live == ((ii - jj) % L == 0)).all()
            else:
                assert live.all()
            assert (got_f[live] == want[live]).all(), (L, p, np.argwhere((got_f != want) & live)[:5])
    batch.close()
    model.close()


@pytest.mark.parametrize("L", [2, 3, 6])
def test_viterbi_gap_unit_lengths(oracle, L):
    from coati_amd import hip

    rng = np.random.default_rng(30 + L)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    unit = 3 * L if L % 3 else L
    pairs = util.make_pairs(rng, 60, 1, 70, L=L, amb=0.03) + [("", ""), ("ACG" * (unit // 3), ""), ("", "ACGTAC" * L)]
    # lengths around the lane (16) and strip (1024) boundaries
    for lb in (L, 16 // L * L, 32 // L * L + L, 1020 // L * L, 1026 // L * L + L, 2052 // L * L):
        nc = max(unit // 3, (lb // 3) // (unit // 3) * (unit // 3))
        a = util.random_anc(rng, nc)
        d = (util.mutate(rng, a) + "".join(rng.choice(list("ACGT"), lb)))[:lb]
        pairs.append((a, d))
    viterbi_compare(hip, oracle, table, consts, L, pairs)


def test_viterbi_gap_len_3_known_answer(oracle):
    """align_marginal.cc:211-221: gap.len = 3 -> AC------GAAT."""
    from coati_amd import hip, host

    table = host.set_subst("mar-mg")
    a, b = host.encode("ACGTTAAGGGGT", "ACGAAT")
    model = hip.Model(table, host.gap_consts(), 3)
    scores, ops, off, ln = model.viterbi(*hip.pack_pairs([(a, b)]))
    got = ops[int(off[0]):int(off[0]) + int(ln[0])]
    assert oracle.ops_to_strings(got, "ACGTTAAGGGGT", "ACGAAT") == ("ACGTTAAGGGGT", "AC------GAAT")


def test_generic_kernel_equals_l1_kernel():
    """COATI_HIP_FORCE_GENERIC=1 routes gap_len 1 through dp_generic: the whole GPU Viterbi suite must
    still pass bit-exactly (two independent kernels, one oracle)."""
    env = dict(os.environ, COATI_HIP_FORCE_GENERIC="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", str(ROOT / "tests" / "test_gpu_viterbi.py"),
                          "-k", "small_mixed or tie_heavy or edge_lengths"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]


def test_generic_kernel_still_serves_gap_len_2_and_3():
    """gap_len 2 and 3 normally run on viterbi_k / forward_k (live cells only).  Forced through
    dp_generic (which fills every cell like the reference) the same tests must pass: two
    independent implementations, one oracle."""
    if os.environ.get("COATI_HIP_FORCE_GENERIC"):
        pytest.skip("already inside the forced run")
    env = dict(os.environ, COATI_HIP_FORCE_GENERIC="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", str(ROOT / "tests" / "test_gpu_generic.py"),
                          "-k", "gap_unit_lengths or forward_matrices"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]


def test_fast_forward_mode_within_tolerance():
    """COATI_HIP_FORWARD_FAST=1 swaps the libm restatement for hardware exp2/log2 (3.5x faster fill);
    the Forward and sampling suites must then still hold north_star's 1e-5 relative bound."""
    if os.environ.get("COATI_HIP_FORWARD_FAST") or os.environ.get("COATI_HIP_FORCE_GENERIC"):
        pytest.skip("already inside a child run")
    env = dict(os.environ, COATI_HIP_FORWARD_FAST="1")
    # (the sampling suite runs in both modes in ITS process, through coati_hip_model_set_option: tests/test_gpu_sample.py; here
    # the environment's default -- what COATI_HIP_FORWARD_FAST=1 makes of every model -- on the Forward matrices)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", str(ROOT / "tests" / "test_gpu_generic.py"),
                          "-k", "forward_matrices or forward_golden"],
                         env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]


@pytest.mark.parametrize("w", [1, 2, 4, 8, 16])
def test_forward_strip_shapes(w):
    """forward_l1 narrows its strips (16 -> 8 -> 4 -> 2 -> 1 columns per lane) while a batch has few
    strips for the GPU's SIMDs, and a handful of pairs run QUAD strips (four lanes per column): what the small
    batches of this suite get.  COATI_HIP_FWD_W forces the others (1: the 1-column strips, quads off): same bits
    (matrices, final cells, samples)."""
    if os.environ.get("COATI_HIP_FWD_W") or os.environ.get("COATI_HIP_FORCE_GENERIC") or os.environ.get("COATI_HIP_FORWARD_FAST"):
        pytest.skip("already inside a child run")
    env = dict(os.environ, COATI_HIP_FWD_W=str(w), COATI_HIP_FWD_QUAD="0")
    # (the sampling suite runs in both modes in ITS process, through coati_hip_model_set_option: tests/test_gpu_sample.py; here
    # the environment's default -- what COATI_HIP_FORWARD_FAST=1 makes of every model -- on the Forward matrices)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", str(ROOT / "tests" / "test_gpu_generic.py"),
                          "-k", "forward_matrices or forward_golden"],
                         env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]


def rel_close(got, want, tol=1e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    finite = want > -1e30
    assert (got[~finite] < -1e30).all()
    return np.abs(got[finite] - want[finite]) <= tol * np.maximum(1.0, np.abs(want[finite]))


@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_forward_matrices_vs_oracle(oracle, L):
    from coati_amd import hip

    rng = np.random.default_rng(40 + L)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    unit = 3 * L if L % 3 else L  # ancestor lengths are multiples of 3 and of L
    pairs = util.make_pairs(rng, 30, 1, 60, L=L, amb=0.03) + [("", ""), ("ACG" * (unit // 3), "")]
    a_long = util.random_anc(rng, 120 * (1 if L == 1 else L))
    d_long = util.mutate(rng, a_long)
    pairs.append((a_long, d_long[: len(d_long) // L * L]))
    big = util.random_anc(rng, 348)
    pairs.append((big, (util.mutate(rng, big) + "ACGT" * 30)[: 1032 // L * L]))  # crosses the strip boundary
    enc = util.encode_pairs(pairs)
    model = hip.Model(table, consts, L)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.forward_launch()
    final = batch.forward_final()
    for p, (a, b) in enumerate(enc):
        M, D, I = oracle.fill(oracle.LOG, table, consts, L, a, b)
        want_final = np.array([M[-1, -1], D[-1, -1], I[-1, -1]])
        assert rel_close(final[p], want_final).all(), (L, p, final[p], want_final)
        if util.forward_exact():  # default mode: the reference's own libm arithmetic, bit for bit
            assert util.same_bits(final[p], want_final), (L, p, final[p], want_final)
        if len(a) * len(b) == 0:
            continue
        gM, gD, gI = batch.debug_forward_matrices(p)
        for got, want in ((gM, M), (gD, D), (gI, I)):
            w = want[L:, L:].copy()
            ok = rel_close(got.ravel()[:-1], w.ravel()[:-1])  # (the oracle's last cell is terminal-adjusted)
            assert ok.all(), (L, p, len(a), len(b))
            if util.forward_exact():
                live = w.ravel()[:-1] > -1e30
                assert util.same_bits(got.ravel()[:-1][live], w.ravel()[:-1][live]), (L, p, len(a), len(b))
    batch.close()
    model.close()


def test_forward_golden_final_cells():
    """Final M/D/I of the reference's Forward for the committed sample cases (tests/golden)."""
    from coati_amd import hip, host

    doc = json.loads((GOLD / "sample_cases.json").read_text())
    table = np.load(GOLD / "table_mg94_goldenP.npy")
    consts = host.gap_consts(doc["gap_open"], doc["gap_extend"])
    for L in (1, 3):
        cases = [c for c in doc["cases"] if c["gap_len"] == L]
        enc = [host.encode(c["anc"], c["des"]) for c in cases]
        model = hip.Model(table, consts, L)
        batch = hip.Batch(model, *hip.pack_pairs(enc))
        batch.forward_launch()
        final = batch.forward_final()
        for p, c in enumerate(cases):
            want = np.array([int(c[f"final_{m}_bits"], 16) for m in "MDI"], np.uint32).view(np.float32)
            assert rel_close(final[p], want).all(), (c["name"], final[p], want)
            if util.forward_exact():
                assert util.same_bits(final[p], want), (c["name"], final[p], want)
