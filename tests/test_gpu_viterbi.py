"""Parity of the HIP Viterbi path (through the C ABI) with the CPU oracle.
Bit-exact: fp32 score bits, every alignment op, and every per-cell traceback
decision byte."""
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


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def hip():
    from coati_amd import hip as h

    assert h.device_count() > 0, "no gfx950 device: the HIP path cannot run"
    return h


@pytest.fixture(params=["auto", "ck", "bits"], autouse=True)
def kernel_choice(request, monkeypatch):
    """gap_len 1 has two kernels and a planner that picks by batch shape (abi.hip: viterbi_ck -- lean
    fill + checkpoint traceback -- except for batches of short pairs and narrow-strip long pairs, which
    keep viterbi_l1).  Every test of this file runs three times: planner's choice, viterbi_ck forced
    (so its 4- and 8-column shapes, multi-strip hand-off and tile recompute see the small and the
    ragged cases too), viterbi_l1 forced."""
    monkeypatch.delenv("COATI_HIP_VITERBI_CK", raising=False)
    monkeypatch.delenv("COATI_HIP_VITERBI_BITS", raising=False)
    if request.param == "ck":
        monkeypatch.setenv("COATI_HIP_VITERBI_CK", "1")
    elif request.param == "bits":
        monkeypatch.setenv("COATI_HIP_VITERBI_BITS", "1")
    return request.param


def run_and_compare(hip, oracle, table, consts, pairs, check_flags=True):
    enc = util.encode_pairs(pairs)
    a_cat, a_off, b_cat, b_off = hip.pack_pairs(enc)
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    batch.viterbi_launch()
    scores, ops, ops_off, ops_len = batch.viterbi_fetch()
    for p, (a, b) in enumerate(enc):
        want_ops, want_score = oracle.viterbi(table, consts, 1, a, b, lowmem=len(a) * len(b) > 4_000_000)
        got = ops[int(ops_off[p]):int(ops_off[p]) + int(ops_len[p])]
        assert bits(scores[p]) == bits(want_score), (p, len(a), len(b), scores[p], want_score)
        assert len(got) == len(want_ops) and (got == want_ops).all(), (p, len(a), len(b))
        slot0 = int(a_off[p] + b_off[p])
        assert slot0 <= int(ops_off[p]) and int(ops_off[p]) + int(ops_len[p]) <= slot0 + len(a) + len(b)
        if check_flags and 0 < len(a) * len(b) <= 400_000:
            M, D, I = oracle.fill(oracle.TROPICAL, table, consts, 1, a, b)
            # undo the terminal adjustment of the last cell so that flags are comparable there too
            want = oracle.tb_flags(M, D, I, consts)[1:, 1:].copy()
            got_f = batch.debug_flags(p)
            want[-1, -1] = got_f[-1, -1]  # last cell: oracle matrices are terminal-adjusted there
            assert (got_f == want).all(), (p, np.argwhere(got_f != want)[:5])
    batch.close()
    model.close()


def test_small_mixed_pairs(hip, oracle):
    rng = np.random.default_rng(11)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    pairs = util.make_pairs(rng, 200, 1, 60, amb=0.05)
    run_and_compare(hip, oracle, table, consts, pairs)


def test_tie_heavy_table(hip, oracle):
    rng = np.random.default_rng(12)
    consts = oracle.gap_consts()
    run_and_compare(hip, oracle, util.tie_table(), consts, util.make_pairs(rng, 120, 1, 50))


def test_edge_lengths(hip, oracle):
    rng = np.random.default_rng(13)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    anc = util.random_anc(rng, 12)
    pairs = [("", ""), ("", "ACGT"), (anc, ""), ("AAA", "A"), ("AAA", "C" * 40)]
    # descendant lengths around the lane (16) and strip (1024) boundaries
    for lb in (1, 15, 16, 17, 31, 32, 33, 1023, 1024, 1025, 1040, 2047, 2049):
        nc = max(1, lb // 3)
        a = util.random_anc(rng, nc)
        d = util.mutate(rng, a)
        d = (d + "".join(rng.choice(list("ACGT"), lb)))[:lb]
        pairs.append((a, d))
    # long ancestor vs short descendant and vice versa
    pairs.append((util.random_anc(rng, 700), "ACGTACGTAC"))
    pairs.append(("ATG", "".join(rng.choice(list("ACGT"), 2100))))
    run_and_compare(hip, oracle, table, consts, pairs)


def test_other_gap_parameters(hip, oracle):
    rng = np.random.default_rng(14)
    table = util.random_table(rng)
    for g, e in ((0.01, 0.5), (0.2, 0.9), (1e-6, 0.1)):
        consts = oracle.gap_consts(g, e)
        run_and_compare(hip, oracle, table, consts, util.make_pairs(rng, 40, 1, 40), check_flags=False)


def test_1kb_pairs(hip, oracle):
    rng = np.random.default_rng(15)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    pairs = []
    for _ in range(24):
        a = util.random_anc(rng, 334)
        pairs.append((a, util.mutate(rng, a)))
    run_and_compare(hip, oracle, table, consts, pairs, check_flags=False)


def test_one_shot_chunked_equals_resident(hip, oracle):
    rng = np.random.default_rng(16)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    enc = util.encode_pairs(util.make_pairs(rng, 50, 1, 80))
    a_cat, a_off, b_cat, b_off = hip.pack_pairs(enc)
    model = hip.Model(table, consts, 1)
    scores, ops, ops_off, ops_len = model.viterbi(a_cat, a_off, b_cat, b_off)
    for p, (a, b) in enumerate(enc):
        want_ops, want_score = oracle.viterbi(table, consts, 1, a, b)
        got = ops[int(ops_off[p]):int(ops_off[p]) + int(ops_len[p])]
        assert bits(scores[p]) == bits(want_score) and (got == want_ops).all()


def test_workspace_reuse_across_batches(hip, oracle):
    """A model hands the HBM workspace of a destroyed batch to the next one (coati_hip.h:
    coati_hip_model_trim).  Batches of different shapes created, run and destroyed in turn on ONE
    model -- larger after smaller, smaller into a larger leftover, Forward + samples in between --
    must give what a fresh model gives; the stale contents of a reused workspace must never show."""
    from coati_amd import host

    rng = np.random.default_rng(19)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    model = hip.Model(table, consts, 1)
    shapes = [(30, 60), (8, 20), (3, 400), (40, 90), (1, 5), (25, 60)]
    for round_no, (n, max_cod) in enumerate(shapes):
        enc = util.encode_pairs(util.make_pairs(rng, n, 1, max_cod) + [("", ""), ("ACG", "")])
        batch = hip.Batch(model, *hip.pack_pairs(enc))
        batch.viterbi_launch()
        scores, ops, off, ln = batch.viterbi_fetch()
        if round_no % 2 == 1:  # Forward on the same (reused) workspace
            batch.forward_launch()
            final = batch.forward_final()
        for p, (a, b) in enumerate(enc):
            want_ops, want_score = oracle.viterbi(table, consts, 1, a, b)
            got = ops[int(off[p]):int(off[p]) + int(ln[p])]
            assert bits(scores[p]) == bits(want_score) and len(got) == len(want_ops) and (got == want_ops).all(), (round_no, p)
            if round_no % 2 == 1 and util.forward_exact():
                M, D, I = oracle.fill(oracle.LOG, table, consts, 1, a, b)
                assert util.same_bits(final[p], np.array([M[-1, -1], D[-1, -1], I[-1, -1]], np.float32)), (round_no, p)
        batch.close()
        if round_no == 3:
            model.trim()  # drop the cache in the middle: the next batch allocates afresh
    model.trim()
    model.close()


def test_model_destroyed_before_its_batch(hip, oracle):
    """coati_hip.h: a batch keeps its model alive -- destroying the model handle first leaves the
    batch usable and nothing dangling when the batch goes last."""
    rng = np.random.default_rng(23)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    enc = util.encode_pairs(util.make_pairs(rng, 6, 5, 40))
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    model.close()
    batch.viterbi_launch()
    scores, ops, off, ln = batch.viterbi_fetch()
    for p, (a, b) in enumerate(enc):
        want_ops, want_score = oracle.viterbi(table, consts, 1, a, b)
        got = ops[int(off[p]):int(off[p]) + int(ln[p])]
        assert bits(scores[p]) == bits(want_score) and (got == want_ops).all()
    batch.close()


def test_one_shot_call_chunks_to_the_memory_budget():
    """coati_hip_viterbi_batch cuts its input into as many resident batches as the free HBM needs.
    COATI_HIP_MEM_BUDGET (bytes) forces that with a tiny budget in a child process: many chunks,
    same scores / ops / offsets as the unchunked call."""
    import json
    import os
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys, zlib, json, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip, host
from tests import util
rng = np.random.default_rng(31)
enc = util.encode_pairs(util.make_pairs(rng, 120, 1, 90) + [("", ""), ("ACG", ""), ("", "ACGT")])
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
scores, ops, off, ln = model.viterbi(*hip.pack_pairs(enc))
crc = 0
for p in range(len(enc)):
    crc = zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes(), crc)
print(json.dumps({"ops": crc, "scores": zlib.crc32(scores.tobytes()), "len": int(ln.sum()), "off": zlib.crc32(off.tobytes())}))
''' % str(root)
    outs = []
    for budget in (None, "200000"):  # ~200 kB: a handful of pairs per chunk
        env = dict(os.environ)
        env.pop("COATI_HIP_MEM_BUDGET", None)
        if budget:
            env["COATI_HIP_MEM_BUDGET"] = budget
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1], outs


def test_invalid_inputs_rejected(hip, oracle):
    table = util.random_table(np.random.default_rng(1))
    consts = oracle.gap_consts()
    model = hip.Model(table, consts, 1)
    a = np.array([0, 1, 2, 3], np.uint8)  # length not a multiple of 3
    with pytest.raises(hip.CoatiHipError):
        hip.Batch(model, *hip.pack_pairs([(a, np.array([0], np.uint8))]))
    with pytest.raises(hip.CoatiHipError):  # descendant code 15 ('-') is not a table column
        hip.Batch(model, *hip.pack_pairs([(np.array([0, 1, 2], np.uint8), np.array([15], np.uint8))]))
    with pytest.raises(hip.CoatiHipError):  # ancestor code out of range
        hip.Batch(model, *hip.pack_pairs([(np.array([183, 1, 2], np.uint8), np.array([1], np.uint8))]))
    with pytest.raises(hip.CoatiHipError):
        hip.Model(table, consts, 0)


def test_long_pair_strips_pipelined_across_wavefronts(hip, oracle):
    """A 21 kb x 20 kb pair = 20 strips handed from wavefront to wavefront, in a batch that also holds
    short pairs (uneven load), vs the oracle's low-memory Viterbi."""
    rng = np.random.default_rng(17)
    table = util.random_table(rng)
    consts = oracle.gap_consts()
    a = util.random_anc(rng, 7000)
    d = util.mutate(rng, a, n_indel=40, mean_len=9)[:20000]
    pairs = util.make_pairs(rng, 300, 1, 120) + [(a, d)] + util.make_pairs(rng, 100, 50, 400)
    a2 = util.random_anc(rng, 1500)
    pairs.insert(7, (a2, util.mutate(rng, a2, n_indel=10)))  # a second multi-strip pair (5 strips)
    run_and_compare(hip, oracle, table, consts, pairs, check_flags=False)


@pytest.mark.parametrize("strip_w", ["2", "4", "8", "16"])
def test_strip_shapes_forced(oracle, strip_w):
    """The three strip shapes (4, 8, 16 columns per lane) produce the same bit-exact results;
    COATI_HIP_STRIP_W forces the main shape, the last strip of a pair picks its own.  Runs in a
    child process because the library reads the variable when a batch is created."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip
from oracle import pyoracle as orc
from tests import util
rng = np.random.default_rng(77)
table = util.random_table(rng)
consts = orc.gap_consts()
pairs = util.make_pairs(rng, 40, 1, 420, L=1, amb=0.02)
# descendant lengths around every shape boundary, one pair spanning several strips of any shape
for nb in (1, 3, 4, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1030, 1290, 1540, 2049):
    anc = util.random_anc(rng, max(1, nb // 3 + int(rng.integers(0, 3))))
    des = "".join(rng.choice(list(util.NT), nb))
    pairs.append((anc, des))
pairs.append((util.random_anc(rng, 900), util.mutate(rng, util.random_anc(rng, 1100), n_indel=6)))
enc = util.encode_pairs(pairs)
model = hip.Model(table, consts, 1)
scores, ops, off, ln = model.viterbi(*hip.pack_pairs(enc))
bad = 0
for p, (a, b) in enumerate(enc):
    w_ops, w_sc = orc.viterbi(table, consts, 1, a, b)
    got = ops[int(off[p]):int(off[p]) + int(ln[p])]
    if not (len(got) == len(w_ops) and (got == w_ops).all() and np.float32(scores[p]).view(np.uint32) == np.float32(w_sc).view(np.uint32)):
        bad += 1
        print("MISMATCH pair", p, len(a), len(b))
print("checked", len(enc), "bad", bad)
sys.exit(1 if bad else 0)
''' % str(root)
    env = dict(os.environ, COATI_HIP_STRIP_W=strip_w)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]


def test_per_pair_tables(oracle):
    """A model with several substitution tables (per-leaf branch lengths of `coati msa`,
    align_msa.cc:285-318) and a batch whose pairs pick among them: every pair equals the oracle run
    with ITS table; Viterbi bit-exact, Forward final cell within 1e-5."""
    from coati_amd import hip

    rng = np.random.default_rng(21)
    tables = np.stack([util.random_table(rng) for _ in range(5)])
    consts = oracle.gap_consts()
    pairs = util.make_pairs(rng, 60, 1, 200, L=1, amb=0.02)
    enc = util.encode_pairs(pairs)
    tix = rng.integers(0, 5, len(enc)).astype(np.uint32)
    model = hip.Model(tables, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc), table_index=tix)
    batch.viterbi_launch()
    scores, ops, off, ln = batch.viterbi_fetch()
    batch.forward_launch()
    final = batch.forward_final()
    for p, (a, b) in enumerate(enc):
        w_ops, w_sc = oracle.viterbi(tables[tix[p]], consts, 1, a, b)
        got = ops[int(off[p]):int(off[p]) + int(ln[p])]
        assert len(got) == len(w_ops) and (got == w_ops).all(), p
        assert np.float32(scores[p]).view(np.uint32) == np.float32(w_sc).view(np.uint32), p
        M, D, I = oracle.fill(oracle.LOG, tables[tix[p]], consts, 1, a, b)
        want = np.array([M[-1, -1], D[-1, -1], I[-1, -1]], np.float64)
        assert (np.abs(final[p] - want) <= 1e-5 * np.maximum(1.0, np.abs(want))).all(), p
    with pytest.raises(hip.CoatiHipError):
        hip.Batch(model, *hip.pack_pairs(enc), table_index=np.full(len(enc), 5, np.uint32))
    batch.close()
    model.close()


def test_extremely_ragged_pairs():
    """tools/ragged_check.py: a few rows x 50 000 columns, 30 000 rows x 1 column, empty sides ...
    through Viterbi, Forward and exact-stream sampling, all bit-exact against the oracle."""
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "ragged_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ragged_check ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_pipelined_one_shot_pinned_and_pageable(hip, oracle):
    """coati_hip_viterbi_batch pipelines chunks over three slots (own stream, workspace, page-locked
    staging).  Page-locked caller arrays (coati_hip_host_alloc) are copied by DMA directly, pageable
    ones go through the staging block: both must give what one resident batch gives, also when the
    schedule has many more chunks than slots (slot reuse) and when a model is reused for a second call."""
    from coati_amd import host

    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 2600)  # 3 chunks by the ramp-up schedule
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    batch.viterbi_launch()
    want = batch.viterbi_fetch()
    batch.close()

    # (a pair's ops occupy the tail of its la+lb slot: only [off, off+len) is defined)
    valid = np.zeros(len(want[1]), bool)
    for p in range(len(want[0])):
        valid[int(want[2][p]):int(want[2][p]) + int(want[3][p])] = True

    def same(got):
        sc, ops, off, ln = got
        assert (bits(sc) == bits(want[0])).all() and (ln == want[3]).all() and (off == want[2]).all()
        assert (ops[:len(want[1])][valid] == want[1][valid]).all()

    same(model.viterbi(a_cat, a_off, b_cat, b_off))  # pageable in, pageable out
    pa, pb = hip.pinned_copy(a_cat), hip.pinned_copy(b_cat)
    same(model.viterbi(pa, a_off, pb, b_off, pinned=True))  # page-locked in and out
    same(model.viterbi(pa, a_off, pb, b_off))  # page-locked in, pageable out
    same(model.viterbi(a_cat, a_off, b_cat, b_off, pinned=True))
    model.close()
    # many chunks (tiny budget): slots are reused again and again
    import os
    os.environ["COATI_HIP_MEM_BUDGET"] = str(40 << 20)
    try:
        model = hip.Model(table, consts, 1)
        same(model.viterbi(pa, a_off, pb, b_off, pinned=True))
        same(model.viterbi(a_cat, a_off, b_cat, b_off))
        model.close()
    finally:
        del os.environ["COATI_HIP_MEM_BUDGET"]


@pytest.mark.parametrize("results", ["stored_by_the_kernel", "downloaded"])
def test_streamed_one_shot_persistent_kernel(hip, oracle, kernel_choice, monkeypatch, results):
    """The streamed form of coati_hip_viterbi_batch (one persistent viterbi_ck_stream launch fed chunk by chunk;
    chosen by itself from 4 096 pairs of >= 250 x 250 cells, forced here): bit-exact against one resident
    batch of the same pairs -- page-locked and pageable arrays; many more chunks than the 12 stream slots
    (every slot reused, small unit forced); pairs of two and three strips (own checkpoints, cross-wavefront
    hand-off inside the persistent kernel), empty and one-letter sides in the same call; a model with
    several tables; a second call on the same model.  A few pairs are checked against the oracle too.
    Round 6: the walks store their results straight into host memory -- the caller's page-locked arrays, or the slot's
    staging block for pageable ones -- and nothing is downloaded (the default); and the round-5 form, results downloaded."""
    from coati_amd import host

    if kernel_choice == "bits":
        pytest.skip("viterbi_l1 forced: the streamed form is viterbi_ck only")
    if results == "downloaded":
        monkeypatch.setenv("COATI_HIP_STREAM_HELPERS", "7")
    if kernel_choice == "ck":  # the last chunks cut into row parts (off by default since round 3: pipeline.hip)
        monkeypatch.setenv("COATI_HIP_STREAM_PARTS", "1")
    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    rng = np.random.default_rng(77)
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 1500)
    enc = [(a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]) for p in range(1500)]
    extra = util.encode_pairs(util.make_pairs(rng, 40, 0, 60, L=1, amb=0.05))  # short and empty sides
    for la, lb in [(900, 2300), (1200, 1025), (300, 3100), (2502, 700), (3, 1500), (1500, 1)]:
        extra.append((rng.integers(0, 183, la).astype(np.uint8), rng.integers(0, 4, lb).astype(np.uint8)))
    order = rng.permutation(len(enc) + len(extra))
    enc = [(enc + extra)[i] for i in order]
    a_cat, a_off, b_cat, b_off = hip.pack_pairs(enc)

    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    batch.viterbi_launch()
    want = batch.viterbi_fetch()
    batch.close()
    valid = np.zeros(len(want[1]), bool)
    for p in range(len(enc)):
        valid[int(want[2][p]):int(want[2][p]) + int(want[3][p])] = True
    for p in list(rng.choice(len(enc), 6, replace=False)):
        a, b = enc[p]
        w_ops, w_sc = oracle.viterbi(table, consts, 1, a, b, lowmem=len(a) * len(b) > 4_000_000)
        got = want[1][int(want[2][p]):int(want[2][p]) + int(want[3][p])]
        assert bits(want[0][p]) == bits(w_sc) and len(got) == len(w_ops) and (got == w_ops).all(), p

    def same(got):
        sc, ops, off, ln = got
        assert (bits(sc) == bits(want[0])).all() and (ln == want[3]).all() and (off == want[2]).all()
        assert (ops[:len(want[1])][valid] == want[1][valid]).all()

    monkeypatch.setenv("COATI_HIP_PIPE", "stream")
    pa, pb = hip.pinned_copy(a_cat), hip.pinned_copy(b_cat)
    same(model.viterbi(pa, a_off, pb, b_off, pinned=True))  # default unit: 3 chunks
    same(model.viterbi(a_cat, a_off, b_cat, b_off))
    monkeypatch.setenv("COATI_HIP_STREAM_UNIT", "30000000")  # ~30 pairs of 1 kb: ~50 chunks, every slot reused
    same(model.viterbi(pa, a_off, pb, b_off, pinned=True))
    same(model.viterbi(a_cat, a_off, b_cat, b_off))
    same(model.viterbi(pa, a_off, pb, b_off))
    model.close()
    # several tables in the model (each wavefront keeps its own copy of the table in LDS): table 0 is used
    tables = np.stack([table, util.random_table(rng)])
    model = hip.Model(tables, consts, 1)
    same(model.viterbi(a_cat, a_off, b_cat, b_off))
    model.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_fused_two_strip_pairs(hip, oracle, kernel_choice, monkeypatch, fuse):
    """Round 6: a pair a little wider than one strip (descendant of 1 025 .. 1 280 nt: a full strip + one of <= 256 columns, 4
    per lane) is, in a launch with many more items than wavefronts, done by ONE wavefront -- both strips one after the other,
    then the walk (common.hpp: kCkFusedFirst; forced here on a small batch, and off).  Related pairs, pairs whose walk leaves
    the kept band (unrelated descendants, a long indel: the strip is filled again by the same wavefront), a second strip of ONE
    column and of 256, neighbours that are not fused (one strip; a second strip of 8 columns per lane; three strips), twice on
    the same batch.  Scores, ops and every decision byte as the oracle has them."""
    if kernel_choice == "bits":
        pytest.skip("viterbi_l1 forced: fused items are viterbi_ck's")
    monkeypatch.setenv("COATI_HIP_CK_FUSE", fuse)
    monkeypatch.setenv("COATI_HIP_VITERBI_CK", "1")
    monkeypatch.setenv("COATI_HIP_STRIP_W", "16")
    rng = np.random.default_rng(66)
    table, consts = util.random_table(rng), oracle.gap_consts()
    nt = list(util.NT)
    pairs = []
    for k in range(10):  # related, descendants of 1 025 .. 1 090 nt
        anc = util.random_anc(rng, int(rng.integers(343, 360)))
        des = util.mutate(rng, anc)
        while len(des) <= 1024:
            des += str(rng.choice(nt))
        pairs.append((anc, des))
    anc = util.random_anc(rng, 200)
    pairs.append((anc, "".join(rng.choice(nt, 1025))))  # second strip: one column; unrelated -- the walk leaves the band
    pairs.append((util.random_anc(rng, 345), "".join(rng.choice(nt, 1060))))  # unrelated
    anc = util.random_anc(rng, 350)
    pairs.append((anc, anc[:300] + "".join(rng.choice(nt, 200)) + anc[300:850]))  # a 200-nt insertion, then a deletion of the tail
    pairs.append((util.random_anc(rng, 100), "".join(rng.choice(nt, 1280))))  # second strip: 256 columns (all of its lanes)
    pairs.append((util.random_anc(rng, 120), "".join(rng.choice(nt, 1400))))  # second strip 8 columns per lane: not fused
    pairs.append((util.random_anc(rng, 90), "".join(rng.choice(nt, 2100))))  # three strips: not fused
    pairs += util.make_pairs(rng, 12, 100, 330, L=1)  # one strip each
    pairs = [pairs[i] for i in rng.permutation(len(pairs))]
    run_and_compare(hip, oracle, table, consts, pairs, check_flags=True)
    enc = util.encode_pairs(pairs)
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.viterbi_launch()
    first = batch.viterbi_fetch()
    assert batch.band_stats()[1] >= 3  # (the walks that leave the kept band: their strips were filled again)
    batch.viterbi_launch()
    second = batch.viterbi_fetch()
    assert (bits(first[0]) == bits(second[0])).all() and (first[3] == second[3]).all() and (first[1] == second[1]).all()
    batch.close()
    model.close()


@pytest.mark.parametrize("walk_items", ["0", "1"])
@pytest.mark.parametrize("plan", ["40,3", "25,2", "60,4", "60,8", "60,3,t", "60,4,t", "60,5,t", "40,3,s3", "60,2,s7", "60,4,s1",
                                  "60,4,s5", "60,8,s3", "60,8,s7"])  # (round 6: shortenings the old guard let empty the last part)
def test_last_pairs_cut_into_row_parts(hip, oracle, kernel_choice, monkeypatch, plan, walk_items):
    """viterbi_ck cuts the last pairs of a large batch's LPT order into row parts (own work items; a part leaves
    the lane state at a 64-step boundary, whichever wavefront takes the next part continues -- abi.hip "the ragged
    end"; by itself from 4 352 pairs, forced here on a small batch): scores, ops and every decision byte as the
    oracle has them, for 2, 3, 4 and 8 parts (equal, tapered, and with a last part 1 .. 7 chunks shorter: round 5), with whole pairs, two-strip pairs, short pairs (too short to cut)
    and empty sides in the same batch; twice on the same batch (the progress words are reset per launch)."""
    if kernel_choice == "bits":
        pytest.skip("viterbi_l1 forced: row parts are viterbi_ck's")
    monkeypatch.setenv("COATI_HIP_CK_SPLIT", plan)
    # (round 5) the cut pairs' tracebacks with their last row part, or as work items of their own behind the last parts
    monkeypatch.setenv("COATI_HIP_CK_WALK_ITEMS", walk_items)
    monkeypatch.setenv("COATI_HIP_VITERBI_CK", "1")  # (a batch this small would go to viterbi_l1's planner rule otherwise)
    monkeypatch.setenv("COATI_HIP_STRIP_W", "16")    # (... and be narrowed to 4-column strips: only 16-column single-strip pairs are cut)
    rng = np.random.default_rng(5)
    table, consts = util.random_table(rng), oracle.gap_consts()
    pairs = util.make_pairs(rng, 50, 100, 420, L=1, amb=0.03)  # 300 .. 1 260 nt
    pairs += util.make_pairs(rng, 10, 0, 30, L=1)
    pairs.append((util.random_anc(rng, 200), "".join(rng.choice(list(util.NT), 1500))))  # two strips
    run_and_compare(hip, oracle, table, consts, pairs, check_flags=True)
    # the same batch launched twice gives the same answer
    enc = util.encode_pairs(pairs)
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.viterbi_launch()
    first = batch.viterbi_fetch()
    batch.viterbi_launch()
    second = batch.viterbi_fetch()
    assert (bits(first[0]) == bits(second[0])).all() and (first[3] == second[3]).all()
    for p in range(len(enc)):
        assert (first[1][int(first[2][p]):int(first[2][p]) + int(first[3][p])] == second[1][int(second[2][p]):int(second[2][p]) + int(second[3][p])]).all()
    batch.close()
    model.close()


def test_streamed_call_many_small_chunks_in_quick_succession(hip, kernel_choice, monkeypatch):
    """Thirty streamed calls whose chunks are a few dozen pairs each, so that the host announces chunk after chunk
    within microseconds while the first chunks' wavefronts are still looking up their chunk-table entries.  (Round 4: a
    wavefront could read another slot's entry half-rewritten, take it for its own and index that chunk's work items
    with ticket - first_ticket = -1 -- a GPU memory fault in one run of eight once chunks were planned ahead on helper
    threads; viterbi_ck.hip, the pilot.)  Every call: the resident batch's scores, lengths and ops."""
    from coati_amd import host

    if kernel_choice == "bits":
        pytest.skip("viterbi_l1 forced: the streamed form is viterbi_ck only")
    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    a_cat, a_off, b_cat, b_off = host.synth_encoded(3, 700)
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    batch.viterbi_launch()
    want = batch.viterbi_fetch()
    batch.close()
    monkeypatch.setenv("COATI_HIP_PIPE", "stream")
    for call in range(30):
        monkeypatch.setenv("COATI_HIP_STREAM_UNIT", str(20000000 + 3000000 * (call % 7)))  # 20 .. 38 pairs in the first chunk
        got = model.viterbi(a_cat, a_off, b_cat, b_off)
        assert (bits(got[0]) == bits(want[0])).all() and (got[3] == want[3]).all(), call
        for p in range(0, 700, 53):
            assert (got[1][int(got[2][p]):int(got[2][p]) + int(got[3][p])] == want[1][int(want[2][p]):int(want[2][p]) + int(want[3][p])]).all()
    model.close()


def test_streamed_call_result_array_kinds(hip, kernel_choice, monkeypatch):
    """Round 6: the streamed call's kernel stores a pair's results straight into host memory -- into the caller's arrays when
    ALL of them are page-locked, else into the slot's staging block.  Every kind of caller: all four arrays page-locked; only
    the ops page-locked (-> the staging block); page-locked with the scores, the offsets or the lengths NULL; pageable with
    NULL arrays; no ops at all (nothing for the kernel to store there: the round-5 download).  Chunks of a few dozen pairs, so
    that every slot is taken again the moment its chunk's completion word is seen (what a chunk left dirty in an L2 must not
    land in its successor's upload).  Every byte of every path against the resident batch."""
    import ctypes as C

    from coati_amd import host

    if kernel_choice == "bits":
        pytest.skip("viterbi_l1 forced: the streamed form is viterbi_ck only")
    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    rng = np.random.default_rng(11)
    a_cat, a_off, b_cat, b_off = host.synth_encoded(5, 900)
    enc = [(a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]) for p in range(900)]
    for la, lb in [(699, 1900), (1101, 1030), (39, 2100)]:  # (two and three strips: checkpoints in the chunk's workspace)
        enc.append((rng.integers(0, 183, la).astype(np.uint8), rng.integers(0, 4, lb).astype(np.uint8)))
    enc = [enc[i] for i in rng.permutation(len(enc))]
    a_cat, a_off, b_cat, b_off = hip.pack_pairs(enc)
    n, total = len(enc), int(a_off[-1] + b_off[-1])
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    batch.viterbi_launch()
    want = batch.viterbi_fetch()
    batch.close()
    monkeypatch.setenv("COATI_HIP_PIPE", "stream")
    monkeypatch.setenv("COATI_HIP_STREAM_UNIT", "25000000")
    hip.reload_env()
    pa, pb = hip.pinned_copy(a_cat), hip.pinned_copy(b_cat)
    lib = hip.load()
    ptr = lambda arr: None if arr is None else arr.ctypes.data_as(C.c_void_p)  # noqa: E731
    # (page-locked by registration: memory the CALLER allocated and handed to hipHostRegister -- what an embedder with its own
    # buffers does; the kernel stores into it through the device-visible address the runtime gives for it)
    rt = C.CDLL("libamdhip64.so")
    rt.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    rt.hipHostUnregister.argtypes = [C.c_void_p]
    registered = []

    def make_registered(shape, dtype):
        arr = np.empty(shape, dtype)
        assert rt.hipHostRegister(arr.ctypes.data_as(C.c_void_p), arr.nbytes, 0) == 0
        registered.append(arr)
        return arr

    make = {"pinned": hip.pinned_empty, "pageable": lambda shape, dtype: np.empty(shape, dtype), "registered": make_registered}
    kinds = [("pinned",) * 4, ("registered",) * 4, ("registered", "pinned", "pinned", "registered"), ("pageable", "pinned", "pageable", "pageable"), ("pinned", "pinned", "pageable", "pinned"),
             (None, "pinned", "pinned", "pinned"), ("pinned", "pinned", None, "pinned"), ("pinned", "pinned", "pinned", None),
             (None, "pageable", "pageable", None), ("pinned", None, "pinned", "pinned"), ("pageable",) * 4]
    for rep, kind in enumerate(kinds + kinds[:2]):
        sc, ops, off, ln = (None if k is None else make[k](shape, dt) for k, (shape, dt) in
                            zip(kind, ((n, np.float32), (total, np.uint8), (n, np.uint64), (n, np.uint32))))
        for arr in (sc, ops, off, ln):
            if arr is not None:
                arr.view(np.uint8)[...] = 0xa5
        rc = lib.coati_hip_viterbi_batch(model._h, n, ptr(pa), ptr(a_off), ptr(pb), ptr(b_off), ptr(sc), ptr(ops), total, ptr(off), ptr(ln))
        assert rc == 0, (kind, lib.coati_hip_last_error())
        if sc is not None:
            assert (bits(sc) == bits(want[0])).all(), kind
        if off is not None:
            assert (off == want[2]).all(), kind
        if ln is not None:
            assert (ln == want[3]).all(), kind
        if ops is not None:
            for p in range(n):
                s0, l0 = int(want[2][p]), int(want[3][p])
                assert (ops[s0:s0 + l0] == want[1][s0:s0 + l0]).all(), (kind, p)
        while registered:
            assert rt.hipHostUnregister(registered.pop().ctypes.data_as(C.c_void_p)) == 0
    model.close()


def test_streamed_call_reports_bad_input_and_recovers(hip, kernel_choice, monkeypatch):
    """An invalid pair deep inside a streamed call (the persistent kernel is already running when its chunk is
    planned): the call returns the reference's error (process_marginal, src/lib/utils.cc:822-835), the kernel is
    told to finish and the model is usable again -- the next call gives the resident batch's results."""
    from coati_amd import host

    if kernel_choice == "bits":
        pytest.skip("viterbi_l1 forced: the streamed form is viterbi_ck only")
    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 900)
    model = hip.Model(table, consts, 1)
    monkeypatch.setenv("COATI_HIP_PIPE", "stream")
    monkeypatch.setenv("COATI_HIP_STREAM_UNIT", "60000000")  # ~15 chunks
    bad = b_cat.copy()
    bad[int(b_off[700]) + 3] = 15  # a descendant code the table has no column for
    with pytest.raises(hip.CoatiHipError) as err:
        model.viterbi(a_cat, a_off, bad, b_off)
    assert "descendant code 15 out of range (pair 700)" in str(err.value)
    bad_a = a_cat.copy()
    bad_a[int(a_off[350]) + 500] = 200  # an ancestor code past the table's 183 rows, deep inside a pair
    with pytest.raises(hip.CoatiHipError) as err:
        model.viterbi(bad_a, a_off, b_cat, b_off)
    assert "ancestor code 200 out of range (pair 350)" in str(err.value)
    short = a_off.copy()
    short[-1] -= 1  # the last ancestor is no longer a whole number of codons
    with pytest.raises(hip.CoatiHipError) as err:
        model.viterbi(a_cat, short, b_cat, b_off)
    assert "multiple of 3" in str(err.value)
    got = model.viterbi(a_cat, a_off, b_cat, b_off)
    batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
    batch.viterbi_launch()
    want = batch.viterbi_fetch()
    assert (bits(got[0]) == bits(want[0])).all() and (got[3] == want[3]).all() and (got[2] == want[2]).all()
    batch.close()
    model.close()

