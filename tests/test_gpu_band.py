"""Banded checkpoint storage of viterbi_ck (round 3): only the tiles near a pair's straight line keep their recompute
hints; a walk that leaves the band makes the wavefront fill the pair again with everything kept.  Results must be the
same bits whatever the band."""
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


BAND_CHILD = r'''
import sys, json, zlib, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip, host
from tests import util
rng = np.random.default_rng(99)
pairs = []
for p in range(4400):
    anc = util.random_anc(rng, int(rng.integers(190, 330)))
    des = util.mutate(rng, anc)
    if p %% 20 == 0:    # a long deletion / insertion: the path leaves any narrow band around the straight line
        cut = int(rng.integers(60, 120)) * 3
        at = int(rng.integers(0, max(1, len(des) - cut)))
        des = des[:at] + des[at + cut:] if p %% 40 == 0 else des[:at] + "".join(rng.choice(list("ACGT"), cut)) + des[at:]
    if p %% 97 == 0:
        des = "".join(rng.choice(list("ACGT"), int(rng.integers(520, 1000))))  # unrelated
    des = des[:1020]
    if len(des) <= 520:
        des = des + "".join(rng.choice(list("ACGT"), 530 - len(des)))  # (one 16-column strip: the banded shape)
    pairs.append((anc, des))
enc = util.encode_pairs(pairs)
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
batch = hip.Batch(model, *hip.pack_pairs(enc))
batch.viterbi_launch()
sc, ops, off, ln = batch.viterbi_fetch()
crc = 0
for p in range(len(enc)):
    crc = zlib.crc32(ops[int(off[p]):int(off[p]) + int(ln[p])].tobytes(), crc)
print(json.dumps({"scores_crc": zlib.crc32(sc.tobytes()), "ops_crc": crc, "columns": int(ln.sum())}))
'''


def test_banded_checkpoints_equal_full_checkpoints_and_the_oracle():
    """viterbi_ck keeps the traceback's checkpoints only in a band around the straight line of a pair (round 3); a walk
    that leaves the band makes the wavefront fill the pair again with everything kept.  4 400 pairs that use the
    wave-slot arena, a twentieth of them with a 180-360 nt indel or unrelated sequences: a band of 24 steps (many pairs
    filled twice, reported by COATI_HIP_CK_DEBUG=2), the default band and no band at all give the same bits; a sample
    is compared with the oracle in the narrow-band run's process."""
    import re

    out = {}
    for band in ("24", "96", "0"):
        env = dict(os.environ, COATI_HIP_CK_BAND=band, COATI_HIP_CK_DEBUG="2")
        env.pop("COATI_HIP_VITERBI_BITS", None)
        r = subprocess.run([sys.executable, "-c", BAND_CHILD % str(ROOT)], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        out[band] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        m = re.search(r"(\d+) pairs filled twice \(band (\d+)\)", r.stderr)
        assert m, r.stderr[-2000:]
        out[band]["twice"] = int(m.group(1))
    assert out["24"]["twice"] >= 150, out  # the redo path ran
    assert out["0"]["twice"] == 0
    for band in ("24", "96"):
        assert {k: out[band][k] for k in ("scores_crc", "ops_crc", "columns")} == {k: out["0"][k] for k in ("scores_crc", "ops_crc", "columns")}, out


def test_banded_checkpoints_sample_against_the_oracle(monkeypatch):
    """The same kind of batch in this process (default band): every 60th pair -- the long-indel ones included --
    against the oracle, ops and score bits."""
    from coati_amd import hip, host
    from oracle import pyoracle as orc

    rng = np.random.default_rng(5)
    pairs = []
    for p in range(4300):
        anc = util.random_anc(rng, int(rng.integers(180, 300)))
        des = util.mutate(rng, anc)
        if p % 60 == 0:
            cut = int(rng.integers(60, 110)) * 3
            at = int(rng.integers(0, max(1, len(des) - cut)))
            des = des[:at] + des[at + cut:] if p % 120 == 0 else des[:at] + "".join(rng.choice(list("ACGT"), cut)) + des[at:]
        des = des[:1020]
        if len(des) <= 520:
            des = des + "".join(rng.choice(list("ACGT"), 530 - len(des)))
        pairs.append((anc, des))
    enc = util.encode_pairs(pairs)
    table, consts = host.set_subst("mar-mg"), host.gap_consts()
    model = hip.Model(table, consts, 1)
    batch = hip.Batch(model, *hip.pack_pairs(enc))
    batch.viterbi_launch()
    sc, ops, off, ln = batch.viterbi_fetch()
    for p in range(0, len(enc), 60):
        want_ops, want_score = orc.viterbi(table, consts, 1, enc[p][0], enc[p][1])
        got = ops[int(off[p]):int(off[p]) + int(ln[p])]
        assert np.float32(sc[p]).view(np.uint32) == np.float32(want_score).view(np.uint32), p
        assert len(got) == len(want_ops) and (got == want_ops).all(), p
    batch.close()
    model.close()
