"""viterbi_lp (round 3): the decision-bit fill for batches with fewer strips than the GPU has SIMDs -- a few long
pairs, and every small batch -- in 16-step blocks of hand-allocated instructions, 4, 3 or 2 columns per lane (3: round 6,
decision words per column).  It shares the
decision-bit layout and the traceback with viterbi_l1; results must be the reference's bits (oracle) and equal to
viterbi_l1's."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def hip():
    from coati_amd import hip as h

    assert h.device_count() > 0, "no gfx950 device: the HIP path cannot run"
    return h


@pytest.fixture(autouse=True)
def decision_bit_plan(monkeypatch):
    for v in ("COATI_HIP_VITERBI_CK", "COATI_HIP_VITERBI_BITS", "COATI_HIP_L1_LP", "COATI_HIP_STRIP_W", "COATI_HIP_FORCE_GENERIC", "COATI_HIP_LP_SPLICE", "COATI_HIP_CK_SPLICE"):
        monkeypatch.delenv(v, raising=False)


def run(hip, table, consts, enc, flags_of=()):
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.viterbi_launch()
    scores, ops, off, ln = batch.viterbi_fetch()
    flags = {p: batch.debug_flags(p) for p in flags_of}
    batch.close()
    model.close()
    return scores.copy(), [ops[int(off[p]):int(off[p]) + int(ln[p])].copy() for p in range(len(enc))], flags


@pytest.mark.parametrize("strip_w", ["2", "3", "4"])
def test_block_and_chunk_boundaries_against_the_oracle(hip, oracle, monkeypatch, strip_w):
    """Ancestor lengths around the 16-step blocks, the 64 steps in which the lanes start and the 16-row boundary chunks;
    descendant lengths around one, two and several strips of both shapes; ambiguous nucleotides; every pair against
    the oracle (score bits, ops, and the five decision bits of every cell where the matrix is small)."""
    monkeypatch.setenv("COATI_HIP_STRIP_W", strip_w)
    rng = np.random.default_rng(2024)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    pairs = []
    for la3 in (1, 5, 6, 15, 16, 17, 21, 22, 26, 27, 32, 43, 70):       # codons: 3 ... 210 rows
        for nb in (1, 2, 7, 127, 128, 129, 191, 192, 193, 255, 256, 257, 385, 520, 577, 700):
            anc = util.random_anc(rng, la3)
            des = "".join(rng.choice(list(util.NT16 if nb % 5 == 0 else util.NT), nb))
            pairs.append((anc, des))
    enc = util.encode_pairs(pairs)
    small = [p for p, (a, b) in enumerate(enc) if len(a) * len(b) <= 30_000][::3]
    scores, ops, flags = run(hip, table, consts, enc, flags_of=small)
    for p, (a, b) in enumerate(enc):
        want_ops, want_score = oracle.viterbi(table, consts, 1, a, b)
        assert bits(scores[p]) == bits(want_score), (p, len(a), len(b))
        assert len(ops[p]) == len(want_ops) and (ops[p] == want_ops).all(), (p, len(a), len(b))
    for p in small:
        a, b = enc[p]
        M, D, I = oracle.fill(oracle.TROPICAL, table, consts, 1, a, b)
        want = oracle.tb_flags(M, D, I, consts)[1:, 1:].copy()
        want[-1, -1] = flags[p][-1, -1]  # (the oracle's last cell is terminal-adjusted)
        assert (flags[p] == want).all(), (p, len(a), len(b), np.argwhere(flags[p] != want)[:5])


def test_long_pairs_equal_viterbi_l1_and_the_oracle(hip, oracle, monkeypatch):
    """Three related pairs of 9-31 kb (2-column strips, then 4- and 3-column strips forced) with long indels and a stretch of
    ambiguous nucleotides: identical to viterbi_l1 (COATI_HIP_L1_LP=0) and to the oracle's low-memory Viterbi."""
    rng = np.random.default_rng(31)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    pairs = []
    for codons in (3011, 6007, 10400):
        a = util.random_anc(rng, codons)
        d = list(util.mutate(rng, a, n_indel=30, mean_len=12))
        at = int(rng.integers(0, len(d) - 400))
        d[at:at + 40] = list(rng.choice(list(util.NT16[4:]), 40))
        del d[at + 200:at + 200 + 333]
        pairs.append((a, "".join(d)))
    enc = util.encode_pairs(pairs)
    got = {}
    # (round 6) the spliced traceback -- every strip's wavefront walks its strip speculatively, the true walk splices the records
    # it meets -- is on by default for these pairs; off, and with records that never match ("miss": the walk must get through
    # every strip on its own), the same ops
    for name, env in (("lp2", {}), ("lp4", {"COATI_HIP_STRIP_W": "4"}), ("lp3", {"COATI_HIP_STRIP_W": "3"}), ("l1", {"COATI_HIP_L1_LP": "0"}),
                      ("lp2_nosplice", {"COATI_HIP_LP_SPLICE": "0"}), ("lp3_nosplice", {"COATI_HIP_STRIP_W": "3", "COATI_HIP_LP_SPLICE": "0"}),
                      ("lp3_miss", {"COATI_HIP_STRIP_W": "3", "COATI_HIP_LP_SPLICE": "miss"}), ("lp4_miss", {"COATI_HIP_STRIP_W": "4", "COATI_HIP_LP_SPLICE": "miss"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got[name] = run(hip, table, consts, enc)
        for k in env:
            monkeypatch.delenv(k)
    for name in ("lp4", "lp3", "l1", "lp2_nosplice", "lp3_nosplice", "lp3_miss", "lp4_miss"):
        assert (bits(got[name][0]) == bits(got["lp2"][0])).all(), name
        for p in range(len(enc)):
            assert np.array_equal(got[name][1][p], got["lp2"][1][p]), (name, p)
    for p, (a, b) in enumerate(enc):
        want_ops, want_score = oracle.viterbi(table, consts, 1, a, b, lowmem=True)
        assert bits(got["lp2"][0][p]) == bits(want_score), p
        assert np.array_equal(got["lp2"][1][p], want_ops), p


def test_several_long_pairs_with_their_own_tables(hip, oracle):
    """Per-pair substitution tables (a wavefront reloads its LDS copy between strips of different pairs) and uneven
    lengths in one small batch."""
    rng = np.random.default_rng(5)
    consts = oracle.gap_consts()
    tables = np.stack([util.random_table(rng) for _ in range(3)])
    pairs = []
    for codons in (700, 1500, 90, 2600, 1):
        a = util.random_anc(rng, codons)
        pairs.append((a, util.mutate(rng, a, n_indel=8)))
    enc = util.encode_pairs(pairs)
    which = np.array([0, 2, 1, 2, 0], np.uint32)
    model = hip.Model(tables, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc), table_index=which)
    batch.viterbi_launch()
    scores, ops, off, ln = batch.viterbi_fetch()
    for p, (a, b) in enumerate(enc):
        want_ops, want_score = oracle.viterbi(tables[which[p]], consts, 1, a, b, lowmem=len(a) * len(b) > 4_000_000)
        assert bits(scores[p]) == bits(want_score), p
        assert np.array_equal(ops[int(off[p]):int(off[p]) + int(ln[p])], want_ops), p
    batch.close()
    model.close()


def test_spliced_traceback_on_lopsided_and_unrelated_pairs(hip, oracle, monkeypatch):
    """The spliced traceback (round 6) guesses where a strip's speculative walk should start from the pair's straight line: pairs
    whose lengths differ by a factor of six either way, a pair of UNRELATED sequences (its path wanders: records that do not
    match must simply not be used) and a pair with one 3 kb deletion -- splice on (default), off and forced to miss give the
    oracle's ops."""
    rng = np.random.default_rng(77)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    anc_long = util.random_anc(rng, 4000)
    pairs = [(util.random_anc(rng, 700), "".join(rng.choice(list(util.NT), 12500))),        # 2 100 x 12 500
             (anc_long, util.mutate(rng, anc_long[:2100], n_indel=6)),                        # 12 000 x ~2 100
             (util.random_anc(rng, 2300), "".join(rng.choice(list(util.NT), 7100))),         # unrelated, 6 900 x 7 100
             (anc_long, util.mutate(rng, anc_long[:4000] + anc_long[7000:], n_indel=10))]    # one 3 kb deletion
    enc = util.encode_pairs(pairs)
    want = [oracle.viterbi(table, consts, 1, a, b, lowmem=True) for a, b in enc]
    for env in ({}, {"COATI_HIP_LP_SPLICE": "0"}, {"COATI_HIP_LP_SPLICE": "miss"}, {"COATI_HIP_STRIP_W": "3"}, {"COATI_HIP_STRIP_W": "4"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        scores, ops, _ = run(hip, table, consts, enc)
        for k in env:
            monkeypatch.delenv(k)
        for p, (w_ops, w_sc) in enumerate(want):
            assert bits(scores[p]) == bits(w_sc), (env, p)
            assert np.array_equal(ops[p], w_ops), (env, p)


def test_viterbi_ck_spliced_traceback_of_multi_strip_pairs(hip, oracle, monkeypatch):
    """viterbi_ck's spliced traceback (round 6: every strip's wavefront walks its strip speculatively into a record and bridges from
    its right neighbour's exit to it; the pair's true walk copies what it meets): pairs of 5 ... 25 strips of 8 and 4 columns per
    lane -- lopsided either way, UNRELATED (records that match nothing), one 3 kb and one 5 kb deletion (a bridge and a record that
    overflow their areas), a deletion followed by an insertion of 2 kb (the path leaves the kept checkpoint band: strips are
    filled again and the walk repeated over the same records) -- with the splice on (default), off, without bridges and forced
    to miss: the oracle's score bits and ops."""
    rng = np.random.default_rng(78)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    anc_long = util.random_anc(rng, 4000)
    noise = "".join(rng.choice(list(util.NT), 2001))
    pairs = [(util.random_anc(rng, 700), "".join(rng.choice(list(util.NT), 12500))),        # 2 100 x 12 500
             (anc_long, util.mutate(rng, anc_long[:2700], n_indel=6)),                        # 12 000 x ~2 700
             (util.random_anc(rng, 2300), "".join(rng.choice(list(util.NT), 7100))),         # unrelated, 6 900 x 7 100
             (anc_long, util.mutate(rng, anc_long[:4000] + anc_long[7000:], n_indel=10)),    # one 3 kb deletion
             (anc_long, util.mutate(rng, anc_long[:3000] + anc_long[8000:], n_indel=6)),     # one 5 kb deletion
             (anc_long, anc_long[:3000] + anc_long[5001:9000] + noise + anc_long[9000:]),     # 2 kb out, 2 kb in: off the band
             (anc_long, util.mutate(rng, anc_long, n_indel=40, mean_len=9))]                  # related, many runs
    enc = util.encode_pairs(pairs)
    want = [oracle.viterbi(table, consts, 1, a, b, lowmem=True) for a, b in enc]
    for env in ({}, {"COATI_HIP_CK_SPLICE": "0"}, {"COATI_HIP_CK_SPLICE": "nobridge"}, {"COATI_HIP_CK_SPLICE": "miss"}, {"COATI_HIP_STRIP_W": "4"},
                {"COATI_HIP_STRIP_W": "4", "COATI_HIP_CK_SPLICE": "miss"}, {"COATI_HIP_STRIP_W": "16"}):
        env = dict({"COATI_HIP_VITERBI_CK": "1", "COATI_HIP_STRIP_W": "8"}, **env)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        scores, ops, _ = run(hip, table, consts, enc)
        for k in env:
            monkeypatch.delenv(k)
        for p, (w_ops, w_sc) in enumerate(want):
            assert bits(scores[p]) == bits(w_sc), (env, p)
            assert np.array_equal(ops[p], w_ops), (env, p)
