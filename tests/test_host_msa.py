"""Host pieces of `coati msa`: Newick parser, re-rooting, distances and the merge of
insertion columns, against the reference's own doctest cases (tree.cc, insertions.cc).  No GPU."""
import pytest

pytestmark = pytest.mark.host_answers  # also run under `-m gpu` (tests/conftest.py)

from coati_amd import host

O, C = 111, 99  # open / closed insertion flags


def test_parse_newick_known_answer():
    """tree.cc:198-241"""
    rows = host.newick("(B_b:6.0,(A-a:5.0,C/c:3.0,E.e:4.0)Ancestor:5.0,D%:11.0);")
    assert rows == [(0, "", 0.0, False, 0), (1, "B_b", 6.0, True, 0), (2, "Ancestor", 5.0, False, 0), (3, "A-a", 5.0, True, 2),
                    (4, "C/c", 3.0, True, 2), (5, "E.e", 4.0, True, 2), (6, "D%", 11.0, True, 0)]
    # whitespace is ignored, the semicolon is optional
    assert host.newick(" (A:1,\n\tB:2)r:0.5 ") == [(0, "r", 0.5, False, 0), (1, "A", 1.0, True, 0), (2, "B", 2.0, True, 0)]


@pytest.mark.parametrize("bad", ["", "(A:1,B:2", "(A:1,,B:2);", "(A:x);", "A:1;extra", "(:1);"])
def test_parse_newick_errors(bad):
    with pytest.raises(host.CoatiHostError):  # tree.cc:243-246: "Parsing content of newick tree failed."
        host.newick(bad)


def test_reroot_one_node():
    """tree.cc:373-396"""
    rows = host.newick("(B_b:6.0,(A-a:5.0,C/c:3.0,E.e:4.0)Ancestor:5.0,D%:11.0);", reroot="A-a")
    assert [(r[2], r[4]) for r in rows] == [(5.0, 2), (6.0, 0), (0.0, 2), (5.0, 2), (3.0, 2), (4.0, 2), (11.0, 0)]


CARNIVORES = ("((racoon:19.2,bear:6.8):0.8,((sea_lion:12,seal:12):7.5,((monkey:100.9,cat:47.1):20.6,weasel:18.9):2.1):3.9,"
              "dog:25.5);")


def test_reroot_several_nodes():
    """tree.cc:397-426"""
    rows = host.newick(CARNIVORES, reroot="cat")
    got = {r[0]: (r[4], r[2]) for r in rows}
    assert got[0][0] == 4 and got[0][1] == pytest.approx(3.9)
    assert got[4][0] == 8 and got[4][1] == pytest.approx(2.1)
    assert got[8][0] == 9 and got[8][1] == pytest.approx(20.6)
    assert got[9] == (9, 0.0)


def test_distance_ref():
    """tree.cc:455-480"""
    for node, want in (("racoon", 45.5), ("sea_lion", 48.9), ("weasel", 50.4), ("cat", 99.2)):
        assert host.tree_distance(CARNIVORES, "dog", node) == pytest.approx(want)


MERGE_CASES = [
    # insertions.cc:235-255
    ([(["A"], ["TCATCG"], 14, {5: O}), (["B"], ["TCAGTCG"], 14, {3: O, 6: O})],
     ["TCA-TCG", "TCAGTCG"], {3: C, 6: O}),
    # insertions.cc:257-283
    ([(["A", "B", "C"], ["TCA-TCG", "TCAGTCG", "T-A-TCG"], 14, {3: C, 6: O}), (["D"], ["TCACTCG"], 14, {3: O, 6: O})],
     ["TCA--TCG", "TCAG-TCG", "T-A--TCG", "TCA-CTCG"], {3: C, 4: C, 7: O}),
    # insertions.cc:313-331
    ([(["A"], ["TCACTCG"], 14, {3: O}), (["B"], ["TCAGTCG"], 14, {3: O})],
     ["TCAC-TCG", "TCA-GTCG"], {3: C, 4: C}),
    # insertions.cc:333-363
    ([(["H"], ["AAATTCCAACAACATAAACAAATCTGA"], 54, {}), (["G"], ["AAATTCCAACAACATAAACAAATCTGA"], 54, {}),
      (["C"], ["AAATTCCAACAACATAAACAGATCGGAAGAGAAACTATGCTTTTCTAG"], 96, {i: O for i in range(27, 48)})],
     ["AAATTCCAACAACATAAACAAATCTGA---------------------", "AAATTCCAACAACATAAACAAATCTGA---------------------",
      "AAATTCCAACAACATAAACAGATCGGAAGAGAAACTATGCTTTTCTAG"], {i: C for i in range(27, 48)}),
    # insertions.cc:365-388
    ([(["A"], ["CTTGCAT"], 34, {4: O}), (["B"], ["CTACGTGCAT"], 34, {2: O, 3: O, 4: O, 7: O})],
     ["CT---TGCAT", "CTACGTGCAT"], {2: C, 3: C, 4: C, 7: O}),
]


@pytest.mark.parametrize("sets,want_seqs,want_flags", MERGE_CASES)
def test_merge_indels_known_answers(sets, want_seqs, want_flags):
    names, seqs, flags = host.merge_indels(sets)
    assert seqs == want_seqs
    assert names == [n for s in sets for n in s[0]]
    for pos, f in want_flags.items():
        assert flags.get(pos, 0) == f, (pos, flags)


def test_merge_indels_single_set_fails():
    with pytest.raises(host.CoatiHostError):  # insertions.cc:390-399
        host.merge_indels([(["A"], ["CTTGCAT"], 34, {4: O})])
