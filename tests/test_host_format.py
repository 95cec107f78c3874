"""`coati-format` (src/lib/format.cc): the reference's own doctest cases
(format.cc:130-352) as known answers, through the executable.  Host only: runs without a GPU."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.host_answers  # also run under `-m gpu` (tests/conftest.py)

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "coati_amd" / "_build" / "coati-format"


def run(tmp_path, names, seqs, args, out_ext="fa", expect_fail=False):
    src = tmp_path / "in.fasta"
    src.write_text("".join(f">{n}\n{s}\n" for n, s in zip(names, seqs)))
    out = tmp_path / f"out.{out_ext}"
    r = subprocess.run([str(BIN), str(src), "-o", str(out)] + args, capture_output=True, text=True, timeout=60)
    if expect_fail:
        assert r.returncode != 0
        return r.stderr
    assert r.returncode == 0, r.stderr
    return out.read_text().split()


CASES_FASTA = [
    # (names, seqs, args, expected tokens)                                   format.cc line
    (["Ancestor", "Descendant"], ["AG---T", "ACCCGT"], ["-p"], [">Ancestor", "AG---T", ">Descendant", "ACCCGT"]),  # 264
    (["Ancestor", "Descendant"], ["A--GT", "ACCGT"], ["-p"], [">Ancestor", "A--?GT", ">Descendant", "ACC?GT"]),  # 279
    (["Ancestor", "Descendant"], ["A-----------GT", "ACCCCCCCCCCCGT"], ["-p"],
     [">Ancestor", "A-----------?GT", ">Descendant", "ACCCCCCCCCCC?GT"]),  # 287
    (["Ancestor", "Descendant"], ["A-----------GT", "ACCCCCCCCCCCGT"], ["-p", "-s", "Ancestor"],
     [">Ancestor", "A-----------?GT"]),  # 306
    (["Ancestor", "Descendant1", "Descendant2"], ["A--GT", "ACCGT", "A-CGT"], ["-p", "-x", "1", "3"],
     [">Ancestor", "A--?GT", ">Descendant2", "A-C?GT"]),  # 314
]


@pytest.mark.parametrize("names,seqs,args,want", CASES_FASTA)
def test_format_fasta_known_answers(tmp_path, names, seqs, args, want):
    assert run(tmp_path, names, seqs, args) == want


CASES_PHYLIP = [
    (["Ancestor", "Descend-1", "Descend-2"], ["AGT", "AGT", "AG-"], [], ["Ancestor", "AGT", "Descend-1", "AGT", "Descend-2", "AG-"]),  # 271
    (["Ancestor", "Descend-1", "Descend-2"], ["A-GT", "ACGT", "ACG-"], ["-p", "-c", "X"],
     ["Ancestor", "A-XXGT", "Descend-1", "ACXXGT", "Descend-2", "ACXXG-"]),  # 295
    (["Ancestor", "Descend-1", "Descend-2"], ["A--GT", "ACCGT", "A-CGT"], ["-p", "-x", "1", "3"],
     ["Ancestor", "A--?GT", "Descend-2", "A-C?GT"]),  # 324
    (["Ancestor", "Descend-1", "Descend-2"], ["ACGT", "A-GT", "AC-T"], ["-p", "-c", "$", "-x", "2", "1"],
     ["Descend-1", "A-$$GT", "Ancestor", "AC$$GT"]),  # 334
]


@pytest.mark.parametrize("names,seqs,args,want", CASES_PHYLIP)
def test_format_phylip_known_answers(tmp_path, names, seqs, args, want):
    toks = run(tmp_path, names, seqs, args, out_ext="phy")
    assert int(toks[0]) == len(want) // 2 and int(toks[1]) == len(want[1])
    assert toks[2:] == want


def test_format_reorder_by_name(tmp_path):
    assert run(tmp_path, ["A", "B"], ["AAA", "CCC"], ["-s", "B", "A"]) == [">B", "CCC", ">A", "AAA"]  # format.cc:139


@pytest.mark.parametrize("args,msg", [
    (["-p", "-c", "-"], "Invalid padding character"),          # format.cc:343
    (["-s", "coati"], "Sequence coati not found."),            # format.cc:349
    (["-s", "C", "D"], "not found"),                           # format.cc:163
    (["-x", "5"], "Positions of seqs to extract are of out range"),  # format.cc:183
    (["-x", "0"], "Positions of seqs to extract are of out range"),  # format.cc:194
    (["-c", "X"], "requires"),                                 # utils.cc:443-445 (needs -p)
    (["-s", "A", "-x", "1"], "excludes"),                      # utils.cc:448-450
])
def test_format_errors(tmp_path, args, msg):
    err = run(tmp_path, ["A", "B", "C"], ["AAA", "GGG", "CCC"], args, expect_fail=True)
    assert msg in err


def test_format_converts_to_json(tmp_path):
    import json

    src = tmp_path / "in.fasta"
    src.write_text(">1\nCTCTGGATAGTG\n>2\nCT----ATAGTG\n")
    out = tmp_path / "o.json"
    r = subprocess.run([str(BIN), str(src), "-o", str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    doc = json.loads(out.read_text())
    assert doc["alignment"] == {"1": "CTCTGGATAGTG", "2": "CT----ATAGTG"}


def test_dispatcher_runs_a_verb(tmp_path):
    """`coati format ...` == `coati-format ...` (src/coati.cc.in)."""
    src = tmp_path / "in.fasta"
    src.write_text(">A\nA--GT\n>B\nACCGT\n")
    out = tmp_path / "o.fa"
    r = subprocess.run([str(BIN.parent / "coati"), "format", str(src), "-p", "-o", str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert out.read_text().split() == [">A", "A--?GT", ">B", "ACC?GT"]
    r = subprocess.run([str(BIN.parent / "coati")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "alignpair" in r.stdout and "msa" in r.stdout


def test_genseed_encodes_the_seeded_state():
    """coati-genseed (src/coati-genseed.cc): four base-58 words of the generator state after
    Seed(string_seed_seq(args)) (random.hpp:416-441); the state itself is pinned bit for bit against
    the reference by the RNG stream tests."""
    from coati_amd import host

    alphabet = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"

    def word(u):
        w = [alphabet[0]] * 6
        for i in range(6):
            if u == 0:
                break
            w[5 - i] = alphabet[u % 58]
            u //= 58
        return "".join(w)

    for seeds in (["42"], ["random42", "7"], [""]):
        st = host.rng_seed(seeds)
        lo, hi = int(st[0]), int(st[1])
        want = "-".join(word(x) for x in (lo & 0xFFFFFFFF, lo >> 32, hi & 0xFFFFFFFF, hi >> 32))
        r = subprocess.run([str(BIN.parent / "coati-genseed")] + seeds, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout.strip() == want
    a = subprocess.run([str(BIN.parent / "coati-genseed")], capture_output=True, text=True, timeout=60).stdout.strip()
    assert len(a) == 27 and a.count("-") == 3  # no arguments: machine entropy
