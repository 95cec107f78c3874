#!/usr/bin/env python3
"""Randomised parity campaign beyond the committed tests: random batches (lengths 0..~3000, related /
unrelated / low-complexity pairs, ambiguity codes, coarse "tie" tables, several tables per batch),
gap_len 1..4, Viterbi bit-exact against the oracle and Forward final cells bit-exact (within 1e-5 with
COATI_HIP_FORWARD_FAST=1).
usage: fuzz.py [seconds] [seed]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip
from oracle import pyoracle as orc
from tests import util

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
consts_default = orc.gap_consts()
exact = util.forward_exact()
t_end = time.time() + budget
rounds = pairs_checked = reuse_checked = oneshot_checked = 0
while time.time() < t_end:
    L = int(rng.choice([1, 1, 1, 2, 3, 3, 4]))
    # gap_len 1: the planner's kernel choice, or one of the two kernels forced (read per batch_create)
    import os
    os.environ.pop("COATI_HIP_VITERBI_CK", None)
    os.environ.pop("COATI_HIP_VITERBI_BITS", None)
    forced = rng.choice(["auto", "auto", "ck", "ck", "bits", "l1"])
    os.environ.pop("COATI_HIP_CK_SPLIT", None)
    os.environ.pop("COATI_HIP_CK_WALK_ITEMS", None)
    os.environ.pop("COATI_HIP_L1_LP", None)
    os.environ.pop("COATI_HIP_STRIP_W", None)
    os.environ.pop("COATI_HIP_LP_PAIRTAB", None)
    os.environ.pop("COATI_HIP_LP_SPLICE", None)
    os.environ.pop("COATI_HIP_CK_SPLICE", None)
    os.environ.pop("COATI_HIP_CK_FUSE", None)
    if forced in ("auto", "bits"):  # small batches run on viterbi_lp: both strip shapes, with and without the pair table
        w = str(rng.choice(["", "", "2", "3", "4"]))
        if w:
            os.environ["COATI_HIP_STRIP_W"] = w
        if rng.random() < 0.5:  # ... the single-column gathers (the library re-reads its switches per entry of this plumbing)
            os.environ["COATI_HIP_LP_PAIRTAB"] = "0"
        sp = str(rng.choice(["", "", "", "0", "miss"]))  # (round 6) the spliced traceback: default (on for multi-strip pairs), off, records that never match
        if sp:
            os.environ["COATI_HIP_LP_SPLICE"] = sp
    if forced == "l1":  # the decision-bit kernel it replaced there
        os.environ["COATI_HIP_VITERBI_BITS"] = "1"
        os.environ["COATI_HIP_L1_LP"] = "0"
    if forced == "ck":
        os.environ["COATI_HIP_VITERBI_CK"] = "1"
        # (round 6) viterbi_ck's spliced traceback of multi-strip pairs: narrow strips make them; default (on), off, without bridges, miss
        w = str(rng.choice(["", "", "4", "8"]))
        if w:
            os.environ["COATI_HIP_STRIP_W"] = w
        sp = str(rng.choice(["", "", "0", "nobridge", "miss"]))
        if sp:
            os.environ["COATI_HIP_CK_SPLICE"] = sp
        # (round 6) two-strip pairs with a narrow second strip as ONE wavefront's work (the planner: only in launches of many items)
        if not w and rng.random() < 0.6:
            os.environ["COATI_HIP_CK_FUSE"] = "1"
            if not sp:
                os.environ["COATI_HIP_CK_SPLICE"] = "0"  # (the marks are honoured where the splice is off)
        if rng.random() < 0.5:  # the last pairs of the LPT order cut into row parts (default only from 4 352 pairs)
            # (round 5: equal / tapered parts / a last part 1-7 chunks shorter; the cut pairs' tracebacks with their last part or
            # as items of their own)
            shape = str(rng.choice(["", "", ",t", ",s%d" % int(rng.integers(1, 8))]))
            os.environ["COATI_HIP_CK_SPLIT"] = "%d,%d%s" % (int(rng.integers(1, 40)), int(rng.integers(2, 9)), shape)
            os.environ["COATI_HIP_CK_WALK_ITEMS"] = str(rng.choice(["0", "1"]))
    elif forced == "bits":
        os.environ["COATI_HIP_VITERBI_BITS"] = "1"
    n_tables = int(rng.integers(1, 4))
    tables = np.stack([util.tie_table() if rng.random() < 0.3 else util.random_table(rng) for _ in range(n_tables)])
    g = float(rng.choice([0.001, 0.01, 0.05]))
    e = float(rng.choice([5 / 6, 0.5, 0.9]))
    consts = orc.gap_consts(g, e)
    n = int(rng.integers(1, 40))
    max_cod = int(rng.choice([20, 120, 400, 1000]))
    pairs = util.make_pairs(rng, n, 0 if rng.random() < 0.2 else 1, max_cod, L=L, amb=0.03)
    if rng.random() < 0.3:  # a pair around the strip boundaries
        nb = int(rng.choice([1020, 1024, 1025, 1030, 1090, 1280, 2050, 3075])) // L * L
        unit = 3 * L if L % 3 else L
        anc = util.random_anc(rng, max(unit // 3, (nb // 3) // (unit // 3) * (unit // 3)))
        pairs.append((anc, "".join(rng.choice(list(util.NT), nb))))
    if L == 1 and rng.random() < 0.25:  # (round 6) related pairs a little wider than one strip: fused items
        for _ in range(int(rng.integers(1, 4))):
            anc = util.random_anc(rng, int(rng.integers(330, 400)))
            des = util.mutate(rng, anc, n_indel=int(rng.integers(0, 8)), mean_len=int(rng.choice([4, 12, 60])))
            pairs.append((anc, des + "".join(rng.choice(list(util.NT), max(0, 1025 + int(rng.integers(0, 200)) - len(des))))))
    if L == 1 and rng.random() < 0.12:  # (round 6) a related pair of 6-10 kb with indels: dozens of strips, records, bridges
        anc = util.random_anc(rng, int(rng.integers(2000, 3400)))
        pairs.append((anc, util.mutate(rng, anc, n_indel=int(rng.integers(4, 60)), mean_len=int(rng.choice([4, 12, 40])))))
    enc = util.encode_pairs(pairs)
    tix = rng.integers(0, n_tables, len(enc)).astype(np.uint32)
    model = hip.Model(tables, consts, L)
    batch = hip.Batch(model, *hip.pack_pairs(enc), table_index=tix)
    batch.viterbi_launch()
    scores, ops, off, ln = batch.viterbi_fetch()
    batch.forward_launch()
    final = batch.forward_final()
    for p, (a, b) in enumerate(enc):
        w_ops, w_sc = orc.viterbi(tables[tix[p]], consts, L, a, b, lowmem=len(a) * len(b) > 4_000_000)
        got = ops[int(off[p]):int(off[p]) + int(ln[p])]
        ok = len(got) == len(w_ops) and (got == w_ops).all() and np.float32(scores[p]).view(np.uint32) == np.float32(w_sc).view(np.uint32)
        if ok and len(a) * len(b) <= 2_000_000:
            M, D, I = orc.fill(orc.LOG, tables[tix[p]], consts, L, a, b)
            want = np.array([M[-1, -1], D[-1, -1], I[-1, -1]], np.float64)
            fin = want > -1e30
            ok = bool((np.abs(final[p][fin] - want[fin]) <= 1e-5 * np.maximum(1.0, np.abs(want[fin]))).all() and (final[p][~fin] < -1e30).all())
            if ok and exact:  # default build: the reference's libm arithmetic, bit for bit
                ok = util.same_bits(final[p], want.astype(np.float32))
        if not ok:
            print("MISMATCH", dict(seed=seed, round=rounds, L=L, pair=p, la=len(a), lb=len(b), g=g, e=e))
            sys.exit(1)
        pairs_checked += 1
    batch.close()
    # the same model again with a reordered subset: the new batch takes over the workspace and the
    # Forward block the first one left behind (different layout, stale contents) -- same answers
    if len(enc) > 1:
        sel = rng.permutation(len(enc))[: max(1, len(enc) * 2 // 3)]
        enc2 = [enc[i] for i in sel]
        batch2 = hip.Batch(model, *hip.pack_pairs(enc2), table_index=tix[sel])
        batch2.viterbi_launch()
        s2, o2, f2, l2 = batch2.viterbi_fetch()
        batch2.forward_launch()
        fin2 = batch2.forward_final()
        for q, i in enumerate(sel):
            same = (np.float32(s2[q]).view(np.uint32) == np.float32(scores[i]).view(np.uint32) and int(l2[q]) == int(ln[i]) and
                    (o2[int(f2[q]):int(f2[q]) + int(l2[q])] == ops[int(off[i]):int(off[i]) + int(ln[i])]).all() and
                    util.same_bits(fin2[q], final[i]))
            if not same:
                print("MISMATCH on the reused workspace", dict(seed=seed, round=rounds, L=L, pair=int(i)))
                sys.exit(1)
        batch2.close()
        reuse_checked += len(sel)
    # the one-shot call on the same pairs (table 0): chunk pipeline, or the streamed form (one persistent
    # viterbi_ck_stream launch) with a small random unit so that a few dozen pairs make many chunks
    if L == 1 and rng.random() < 0.5:
        os.environ.pop("COATI_HIP_STREAM_UNIT", None)
        os.environ.pop("COATI_HIP_STREAM_PARTS", None)
        os.environ.pop("COATI_HIP_STREAM_HELPERS", None)
        # the streamed call's end game (default: ONE last chunk in three row parts, from a model's second call on; 0: none; 1:
        # round 3's small tail chunks; 22 / 24: two / four parts) and which of its helper-thread uses are on
        parts = str(rng.choice(["", "", "0", "1", "22", "24"]))
        if parts:
            os.environ["COATI_HIP_STREAM_PARTS"] = parts
        helpers = str(rng.choice(["", "", "", "0", "1", "2", "6", "7", "23", "39"]))  # (default 55: + 16 / 32, results stored by the kernel into host memory)
        if helpers:
            os.environ["COATI_HIP_STREAM_HELPERS"] = helpers
        form = str(rng.choice(["stream", "stream", "chunks"]))
        os.environ["COATI_HIP_PIPE"] = form
        if form == "stream":
            os.environ["COATI_HIP_STREAM_UNIT"] = str(int(rng.choice([2e4, 3e5, 4e6, 1e9])))
        a_cat, a_off, b_cat, b_off = hip.pack_pairs(enc)
        pinned = bool(rng.random() < 0.5)
        if pinned:
            a_cat, b_cat = hip.pinned_copy(a_cat), hip.pinned_copy(b_cat)
        s3, o3, f3, l3 = model.viterbi(a_cat, a_off, b_cat, b_off, pinned=pinned)
        if form == "stream" and rng.random() < 0.6:  # (a model's SECOND streamed call is the first that may cut its last chunk into row parts)
            s3, o3, f3, l3 = model.viterbi(a_cat, a_off, b_cat, b_off, pinned=pinned)
        os.environ.pop("COATI_HIP_PIPE", None)
        for p, (a, b) in enumerate(enc):
            w_ops, w_sc = orc.viterbi(tables[0], consts, L, a, b, lowmem=len(a) * len(b) > 4_000_000)
            got = o3[int(f3[p]):int(f3[p]) + int(l3[p])]
            if not (len(got) == len(w_ops) and (got == w_ops).all() and np.float32(s3[p]).view(np.uint32) == np.float32(w_sc).view(np.uint32)):
                print("MISMATCH in the one-shot call", dict(seed=seed, round=rounds, form=form, unit=os.environ.get("COATI_HIP_STREAM_UNIT"), pinned=pinned, pair=p, la=len(a), lb=len(b)))
                sys.exit(1)
        oneshot_checked += len(enc)
    model.close()
    rounds += 1
print(f"fuzz ok: {rounds} batches, {pairs_checked} pairs, seed {seed}; {reuse_checked} pairs re-run on a reused workspace with identical results; {oneshot_checked} pairs through the one-shot call (chunked / streamed)")
