"""Pin the CPU oracle bit-for-bit against the unmodified reference DP engine
(src/lib/align_pair.cc compiled into oracle/_ref by `make -C oracle ref`).
Only runs where /root/reference was available to build oracle/_ref."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    return oracle


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.mark.parametrize("L", [1, 3])
@pytest.mark.parametrize("table_kind", ["random", "ties"])
def test_viterbi_matrices_alignment_score(ref, L, table_kind):
    rng = np.random.default_rng(100 + L)
    table = util.random_table(rng) if table_kind == "random" else util.tie_table()
    g, e = np.float32(0.001), np.float32(1.0) - np.float32(1.0) / np.float32(6.0)
    consts = ref.gap_consts(g, e)
    pairs = util.make_pairs(rng, 60, 0, 40, L=L, amb=0.05) + [("", ""), ("AAA" * L, ""), ("", "ACG" * L)]
    for anc, des in pairs:
        a, b = util.encode_anc(anc), util.encode_des(des)
        M, D, I, sa, sb, sc = ref.ref_viterbi(table, g, e, L, anc, des, a, b)
        Mo, Do, Io = ref.fill(ref.TROPICAL, table, consts, L, a, b)
        assert (bits(M) == bits(Mo)).all() and (bits(D) == bits(Do)).all() and (bits(I) == bits(Io)).all()
        ops, score = ref.traceback(Mo, Do, Io, consts, L)
        assert ref.ops_to_strings(ops, anc, des) == (sa, sb)
        assert bits(score) == bits(sc)
        ops2, score2 = ref.viterbi(table, consts, L, a, b, lowmem=True)
        assert (ops2 == ops).all() and bits(score2) == bits(score)


@pytest.mark.parametrize("L", [1, 3])
def test_forward_and_sampleback(ref, L):
    rng = np.random.default_rng(200 + L)
    table = util.random_table(rng)
    g, e = np.float32(0.001), np.float32(1.0) - np.float32(1.0) / np.float32(6.0)
    consts = ref.gap_consts(g, e)
    for k, (anc, des) in enumerate(util.make_pairs(rng, 30, 0, 25, L=L) + [("", "")]):
        a, b = util.encode_anc(anc), util.encode_des(des)
        seeds = ["42"] if k % 2 else [f"s{k}", "7"]
        mats, alns, scores = ref.ref_forward_sample(table, g, e, L, anc, des, a, b, seeds, 12)
        M, D, I, E = ref.fill(ref.LOG, table, consts, L, a, b, edges=True)
        mine = np.concatenate([np.stack([M, D, I]), E])
        assert (bits(mine) == bits(mats)).all()
        r1, r2 = ref.rng_seed(seeds), ref.rng_seed(seeds)
        for (sa, sb), sc in zip(alns, scores):
            ops, s1 = ref.sampleback(mine, L, r1)
            assert ref.ops_to_strings(ops, anc, des) == (sa, sb)
            assert bits(s1) == bits(sc)
            ops2, s2 = ref.sampleback_mdi(M, D, I, table, consts, L, a, b, r2)
            assert (ops2 == ops).all() and bits(s2) == bits(s1)
            assert bits(ref.path_logweight(M, D, I, table, consts, L, a, b, ops)) == bits(s1)


def test_rng_stream(ref):
    for seeds in (["42"], [""], ["random42"], ["1", "2", "abc"], ["-17"], ["2147483648"], ["0042"], ["+5"]):
        want = ref.ref_rng_f24(seeds, 32)
        r = ref.rng_seed(seeds)
        got = np.array([ref.rng_f24(r) for _ in range(32)], np.float32)
        assert (bits(want) == bits(got)).all(), seeds
