#!/usr/bin/env python3
"""Offline design aid for viterbi_ck's traceback (round 5): which tiles should a recompute round hold?

The true alignment paths of the synthetic workload (CPU oracle, build container or any box with oracle/_build) are
mapped into tile space -- tile (t, c) = fill lane t (16 columns) x band c (16 wavefront steps), wavefront step of body
cell (bi, bj) = bi + bj // 16 -- and candidate tile-set rules are replayed against them: a round ends where the walk
needs a tile the round does not hold.  Cost model: a round costs its steps x cells per lane whatever the number of
valid tiles (SIMT), so rounds x (steps x cells) is what counts, not tiles.

    63-tile rounds of rounds 2-4 (21 tile columns x 3 bands | 2 x 32 | 32 x 2), 16 steps x 16 cells per lane
    paired rounds (round 5): 32 tiles, two lanes of 8 columns per tile, 17 steps x 8 cells per lane:
        16 tile columns x (band the path enters + the one before); gap runs as in viterbi_ck.hip: paired_chi

Measured on the GPU (COATI_HIP_CK_DEBUG=2) the first rule takes 3.48 rounds per pair; this replay says 3.47.

usage: python3 tools/experiments/tile_sets.py [pairs]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
M, D, I = 0, 1, 2


def true_paths(n):
    from coati_amd import host
    from oracle import pyoracle as orc  # (a design tool, not product code: the oracle is the source of the true paths)

    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    a_cat, a_off, b_cat, b_off = host.synth_encoded(0, n)
    out = []
    for p in range(n):
        a, b = a_cat[int(a_off[p]):int(a_off[p + 1])], b_cat[int(b_off[p]):int(b_off[p + 1])]
        ops, _ = orc.viterbi(table, consts, 1, a, b, lowmem=True)
        out.append((len(a), len(b), np.asarray(ops, np.uint8).copy()))
    return out


def visited_cells(la, lb, ops):
    """body cells whose decision the walker looks up, in walk order, with the kind of move that arrived there"""
    i, j, moved, out = la, lb, M, []
    for op in ops[::-1]:
        if i >= 1 and j >= 1:
            out.append((i - 1, j - 1, moved))
        if op == M:
            i, j = i - 1, j - 1
        elif op == D:
            i -= 1
        else:
            j -= 1
        moved = op
    assert i == 0 and j == 0
    return out


def tiles_r4(bi, bj, mode, streak):
    t0, s = bj >> 4, set()
    if mode == M:
        for dt in range(0, min(20, t0) + 1):
            t = t0 - dt
            jr = min(bj, 16 * (t + 1) - 1)
            cmid = (bi - (bj - jr) + t - 8) >> 4
            s |= {(t, cmid + o) for o in (-1, 0, 1) if cmid + o >= 0}
    elif mode == D:
        c0 = (bi + t0) >> 4
        s |= {(t0 - o, c0 - u) for o in (0, 1) if o <= t0 for u in range(32) if c0 - u >= 0}
    else:
        for dt in range(0, min(31, t0) + 1):
            t = t0 - dt
            s |= {(t, ((bi + t) >> 4) - o) for o in (0, 1) if ((bi + t) >> 4) - o >= 0}
    return s


def tiles_paired(bi, bj, mode, streak, guess=4, cols=16):
    t0, s = bj >> 4, set()
    if streak >= 1 and mode == D:
        c0 = (bi + t0) >> 4
        return {(t0, c0 - u) for u in range(2 * cols) if c0 - u >= 0}
    if streak >= 1 and mode == I:
        return {(t0 - dt, (bi + t0 - dt) >> 4) for dt in range(min(2 * cols - 1, t0) + 1)}
    for dt in range(min(cols - 1, t0) + 1):
        t = t0 - dt
        if dt == 0:
            k_hi = bi + t0
        else:
            pi, pj = (max(bi - guess, 0), bj) if mode == D else ((bi, bj - guess) if mode == I else (bi, bj))
            jr = min(pj, 16 * t + 15)
            if jr < 16 * t:
                jr = 16 * t + 15
            k_hi = pi - (pj - jr) + t
        c_hi = max(k_hi, 0) >> 4
        s |= {(t, c_hi - o) for o in (0, 1) if c_hi - o >= 0}
    return s


def replay(rule, paths):
    rounds, tiles = [], []
    for la, lb, ops in paths:
        have, r, nt, last, streak = set(), 0, 0, None, 0
        for bi, bj, moved in visited_cells(la, lb, ops):
            t = bj >> 4
            if (t, (bi + t) >> 4) not in have:
                streak = streak + 1 if (moved != M and moved == last) else 0
                last = moved
                have = rule(bi, bj, moved, streak)
                assert (t, (bi + t) >> 4) in have
                r, nt = r + 1, nt + len(have)
        rounds.append(r)
        tiles.append(nt)
    return np.mean(rounds), np.mean(tiles), np.percentile(rounds, [50, 90, 100])


if __name__ == "__main__":
    paths = true_paths(int(sys.argv[1]) if len(sys.argv) > 1 else 400)
    rng = np.random.default_rng(1)

    def with_indel(la, lb, ops, n):
        ops, pos, kind = list(ops), int(rng.integers(100, len(ops) - 100)), D if rng.random() < 0.5 else I
        ops[pos:pos] = [kind] * n
        return la + (n if kind == D else 0), lb + (n if kind == I else 0), np.array(ops, np.uint8)

    long_bag = [with_indel(*p, int(rng.integers(90, 300))) for p in paths[:200]]
    for name, bag in (("synthetic (Poisson(2) indels of mean 6 nt)", paths), ("one 90-300 nt indel added", long_bag)):
        for rule_name, rule, per_round in (("63-tile rounds (rounds 2-4)", tiles_r4, 16 * 16), ("paired rounds (round 5)", tiles_paired, 17 * 8)):
            r, t, pct = replay(rule, bag)
            print(f"{name:44s} {rule_name:28s} rounds/pair {r:5.2f} (p50/p90/max {pct})  tiles/pair {t:6.1f}  cell slots per lane and pair {r * per_round:7.0f}")
