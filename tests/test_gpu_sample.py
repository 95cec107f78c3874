"""Stochastic traceback on the GPU (`coati sample` path): device RNG stream, exact-
stream sampling vs the oracle / golden vectors, log-weights, independent streams."""
import json
from pathlib import Path

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(autouse=True, params=["exact", "tolerance"])
def forward_mode(request):
    """Every test of this module runs in both Forward modes of the C ABI (COATI_HIP_OPT_FORWARD_MODE, set per model through
    coati_hip_model_set_option -- in this process, no environment): `exact` must reproduce the reference's bits (draw for
    draw), `tolerance` its log-weights within north_star's 1e-5 relative."""
    from coati_amd import hip

    old = hip.DEFAULT_FORWARD_MODE
    hip.DEFAULT_FORWARD_MODE = hip.FORWARD_EXACT if request.param == "exact" else hip.FORWARD_TOLERANCE
    yield request.param
    hip.DEFAULT_FORWARD_MODE = old


def rel_close(got, want, tol=1e-5):
    got, want = np.float64(got), np.float64(want)
    return abs(got - want) <= tol * max(1.0, abs(want))


def test_device_rng_stream_bit_exact():
    from coati_amd import hip, host

    model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
    for entry in json.loads((GOLD / "rng_streams.json").read_text()):
        st = host.rng_seed(entry["seeds"])
        got = model.debug_rng_f24(st, len(entry["f24_bits"]))
        assert got.view(np.uint32).tolist() == [int(h, 16) for h in entry["f24_bits"]], entry["seeds"]


def run_exact(hip, host, oracle, table, consts, L, pairs, seeds, n_samples):
    enc = util.encode_pairs(pairs)
    model = hip.Model(table, consts, L)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.forward_launch()
    states = np.stack([host.rng_seed(seeds) for _ in enc])
    lw, ops, off, ln, st_out = batch.sampleback(n_samples, states, independent=False)
    identical = total = 0
    for p, (a, b) in enumerate(enc):
        M, D, I = oracle.fill(oracle.LOG, table, consts, L, a, b)
        rng = oracle.rng_seed(seeds)
        in_sync = True
        for s in range(n_samples):
            got = ops[int(off[p, s]):int(off[p, s]) + int(ln[p, s])]
            # every GPU sample is a valid path whose log-weight the oracle reproduces (1e-5 relative)
            assert (got == 1).sum() + (got == 0).sum() == len(a) and (got == 2).sum() + (got == 0).sum() == len(b)
            want_lw = oracle.path_logweight(M, D, I, table, consts, L, a, b, got)
            assert rel_close(lw[p, s], want_lw), (p, s, lw[p, s], want_lw)
            if in_sync:  # same draws as the oracle until a (rare) ulp-level flip shifts the stream
                w_ops, w_lw = oracle.sampleback_mdi(M, D, I, table, consts, L, a, b, rng)
                total += 1
                if len(w_ops) == len(got) and (w_ops == got).all():
                    identical += 1
                    assert rel_close(lw[p, s], w_lw)
                    if util.forward_exact():
                        assert util.same_bits(lw[p, s], np.float32(w_lw)), (p, s, lw[p, s], w_lw)
                else:
                    in_sync = False
        if in_sync:  # the stream was consumed draw for draw
            assert (int(st_out[p, 0]), int(st_out[p, 1])) == (rng.lo, rng.hi)
    batch.close()
    model.close()
    return identical, total


@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_exact_stream_matches_oracle(oracle, L):
    from coati_amd import hip, host

    rng = np.random.default_rng(50 + L)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    unit = 3 * L if L % 3 else L
    pairs = util.make_pairs(rng, 24, 1, 40, L=L) + [("", ""), ("ACG" * (unit // 3), "")]
    identical, total = run_exact(hip, host, oracle, table, consts, L, pairs, ["42"], 20)
    if util.forward_exact():  # same libm arithmetic as the reference: never a flipped draw
        assert identical == total, (identical, total)
    assert identical >= 0.98 * total, (identical, total)


def test_config3_16_pairs_x_1000_samples_match_oracle(oracle):
    """BASELINE configs[3] at its stated size: `coati sample -n 1000` on 16 synthetic 1 kb pairs, mar-mg94.
    The speculative exact-stream sampler against the ORACLE's sampleback (oracle sampleback_mdi, pinned to
    the compiled reference in tests/test_oracle_vs_ref.py), draw for draw: every one of the 16 000 samples
    has the oracle's ops and log-weight bits, and the generator ends in the oracle's state."""
    from coati_amd import hip, host

    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    pairs = [host.synth_raw(i) for i in range(16)]
    identical, total = run_exact(hip, host, oracle, table, consts, 1, pairs, ["42"], 1000)
    assert total == 16 * 1000
    if util.forward_exact():
        assert identical == total, (identical, total)
    assert identical >= 0.97 * total, (identical, total)


def test_reference_doctest_marg_sample():
    """marg_sample known answers (align_marginal.cc:653-672), seed "42", default model."""
    from coati_amd import hip, host

    known = json.loads((GOLD / "reference_known_answers.json").read_text())["marg_sample"]
    table = host.set_subst("mar-mg")
    for case in known:
        anc, des = case["seqs"]
        a, b = host.encode(anc, des)
        model = hip.Model(table, host.gap_consts(), 1)
        batch = hip.Batch(model, *hip.pack_pairs([(a, b)]))
        batch.forward_launch()
        lw, ops, off, ln, _ = batch.sampleback(len(case["out"]), host.rng_seed(["42"]).reshape(1, 2))
        for s, ((want_a, want_b), want_s) in enumerate(zip(case["out"], case["scores"])):
            got = ops[int(off[0, s]):int(off[0, s]) + int(ln[0, s])]
            ia, ib = iter(anc), iter(des)
            sa = "".join("-" if o == 2 else next(ia) for o in got)
            sb = "".join("-" if o == 1 else next(ib) for o in got)
            assert (sa, sb) == (want_a, want_b)
            assert rel_close(lw[0, s], float(want_s), 1e-5)


def test_golden_sample_cases_exact_stream(oracle):
    """Samples of the compiled reference (tests/golden/sample_cases.json): same seeds -> same alignments
    (up to rare ulp-level flips), log-weights within 1e-5."""
    from coati_amd import hip, host

    doc = json.loads((GOLD / "sample_cases.json").read_text())
    table = np.load(GOLD / "table_mg94_goldenP.npy")
    consts = host.gap_consts(doc["gap_open"], doc["gap_extend"])
    same = total = 0
    for c in doc["cases"]:
        a, b = host.encode(c["anc"], c["des"])
        model = hip.Model(table, consts, c["gap_len"])
        batch = hip.Batch(model, *hip.pack_pairs([(a, b)]))
        batch.forward_launch()
        n = len(c["samples"])
        lw, ops, off, ln, _ = batch.sampleback(n, host.rng_seed(c["seeds"]).reshape(1, 2))
        for s, want in enumerate(c["samples"]):
            got = ops[int(off[0, s]):int(off[0, s]) + int(ln[0, s])]
            sa, sb = oracle.ops_to_strings(got, c["anc"], c["des"])
            total += 1
            if (sa, sb) == (want["anc"], want["des"]):
                same += 1
                w = np.array([int(want["score_bits"], 16)], np.uint32).view(np.float32)[0]
                assert rel_close(lw[0, s], w), (c["name"], s)
                if util.forward_exact():
                    assert util.same_bits(lw[0, s], w), (c["name"], s, lw[0, s], w)
            else:
                break  # the stream shifted; later samples of this case are different draws
        batch.close()
        model.close()
    if util.forward_exact():
        assert same == total, (same, total)
    assert same >= 0.97 * total, (same, total)


def test_independent_streams(oracle):
    from coati_amd import hip, host

    rng = np.random.default_rng(60)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    pairs = util.make_pairs(rng, 6, 10, 60)
    enc = util.encode_pairs(pairs)
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.forward_launch()
    states = np.stack([host.rng_seed([f"s{p}"]) for p in range(len(enc))])
    n = 200
    lw_i, ops_i, off_i, ln_i, _ = batch.sampleback(n, states, independent=True)
    lw_e, ops_e, off_e, ln_e, _ = batch.sampleback(n, states, independent=False)
    for p, (a, b) in enumerate(enc):
        M, D, I = oracle.fill(oracle.LOG, table, consts, 1, a, b)
        # sample 0 of both modes is the same draw sequence
        g0 = ops_i[int(off_i[p, 0]):int(off_i[p, 0]) + int(ln_i[p, 0])]
        e0 = ops_e[int(off_e[p, 0]):int(off_e[p, 0]) + int(ln_e[p, 0])]
        assert len(g0) == len(e0) and (g0 == e0).all() and lw_i[p, 0] == lw_e[p, 0]
        for s in range(0, n, 17):
            got = ops_i[int(off_i[p, s]):int(off_i[p, s]) + int(ln_i[p, s])]
            assert rel_close(lw_i[p, s], oracle.path_logweight(M, D, I, table, consts, 1, a, b, got))
        # both modes sample the same distribution: mean log-weights agree within sampling error
        se = np.sqrt(lw_i[p].var() / n + lw_e[p].var() / n) + 1e-6
        assert abs(lw_i[p].mean() - lw_e[p].mean()) < 6 * se


def test_speculative_exact_stream_equals_sequential():
    """The parallel exact-stream sampler (speculated stream offsets) returns exactly what the
    one-walker-per-pair loop returns: same ops, same log-weight bits, same final RNG state -- with its rounds on the
    device (default) and on the host, and with a candidate budget small enough to need dozens of rounds.
    Each variant runs in a child process (the switches are COATI_HIP_* environment variables)."""
    import os
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys, zlib, json, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip, host
from tests import util
rng = np.random.default_rng(5)
pairs = util.make_pairs(rng, 10, 20, 150, L=1) + [("", ""), ("ACG", ""), ("", "ACGT")]
pairs += [host.synth_raw(i) for i in range(3)]
enc = util.encode_pairs(pairs)
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *hip.pack_pairs(enc))
batch.forward_launch()
states = np.stack([host.rng_seed(["42", str(p)]) for p in range(len(enc))])
lw, ops, off, ln, st = batch.sampleback(150, states, independent=False)
crc = 0
for p in range(len(enc)):
    for s in range(150):
        crc = zlib.crc32(ops[int(off[p, s]):int(off[p, s]) + int(ln[p, s])].tobytes(), crc)
flat_off, flat_len = off.reshape(-1).astype(np.int64), ln.reshape(-1).astype(np.int64)
packed = bool(flat_off[0] == 0 and (flat_off[1:] == flat_off[:-1] + flat_len[:-1]).all())  # the ops come back to back, in output order
print(json.dumps({"ops": crc, "lw": zlib.crc32(lw.tobytes()), "len": int(ln.sum()), "st": zlib.crc32(st.tobytes()), "packed": packed}))
''' % str(root)
    # the same call: device rounds (default), the one-walker-per-pair loop, the rounds planned and resolved on the host
    # (round 3's loop), and device rounds with so few candidates that a round resolves a handful of samples per pair and
    # most of a share is one window (many rounds, windows cut by the share, pairs that finish rounds apart)
    # ... and with a step table that holds three or five diagonals only: every gap column takes the walk to cells whose
    # entries the walkers compute themselves (the table's band is 129+ diagonals by default: these pairs never leave it)
    variants = [{}, {"COATI_HIP_SAMPLE_SEQUENTIAL": "1"}, {"COATI_HIP_SPEC_HOST_ROUNDS": "1"}, {"COATI_HIP_SPEC_CANDS": "1024"},
                {"COATI_HIP_SPEC_CANDS": "1024", "COATI_HIP_SPEC_Z": "0.5"}, {"COATI_HIP_SAMPLE_BAND": "1"}, {"COATI_HIP_SAMPLE_BAND": "2", "COATI_HIP_SPEC_HOST_ROUNDS": "1"}]
    outs = []
    for extra in variants:
        env = dict(os.environ)
        for k in ("COATI_HIP_SAMPLE_SEQUENTIAL", "COATI_HIP_SPEC_HOST_ROUNDS", "COATI_HIP_SPEC_CANDS", "COATI_HIP_SPEC_Z", "COATI_HIP_SAMPLE_BAND"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["packed"]
    for extra, out in zip(variants[1:], outs[1:]):
        assert out == outs[0], (extra, out, outs[0])


def test_sampleback_prepare_changes_nothing_but_the_first_call():
    """coati_hip_sampleback_prepare (round 6: the sampler's allocations behind forward_launch, under the Forward kernel -- what
    `coati-sample` does): the samples of a prepared call are those of an unprepared one, for the exact-stream, the
    independent-stream and the one-sample forms; a prepare without forward_launch is refused."""
    from coati_amd import hip, host

    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    enc = host.synth_encoded(0, 5, n_codons=120)
    states = np.array([host.rng_seed(["7"]) for _ in range(5)], np.uint64)
    for n, indep in ((40, False), (40, True), (1, False)):
        got = []
        for prepare in (False, True):
            model = hip.Model(table, consts, 1)
            batch = hip.Batch(model, *enc)
            if prepare:
                with pytest.raises(hip.CoatiHipError):
                    batch.sampleback_prepare(n, indep)
            batch.forward_launch()
            if prepare:
                batch.sampleback_prepare(n, indep)
            got.append(batch.sampleback(n, states, independent=indep))
            batch.close()
            model.close()
        for x, y in zip(got[0], got[1]):
            assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))
