"""End-to-end: the coati-alignpair / coati-sample executables (C++ host layer ->
C ABI -> HIP kernels) on the reference's own doctest cases and sample data."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "coati_amd" / "_build"
KNOWN = json.loads((ROOT / "tests" / "golden" / "reference_known_answers.json").read_text())


def run(binary, *args):
    return subprocess.run([str(BIN / binary), *[str(a) for a in args]], capture_output=True, text=True, timeout=300)


def test_example_001_default_json_to_stdout(tmp_path):
    """BASELINE.json configs[0]: coati alignpair sampledata/example-001.fasta -m mar-mg."""
    fa = tmp_path / "example-001.fasta"
    fa.write_text(">1\nCTCTGGATAGTG\n>2\nCTATAGTG\n")  # content of sampledata/example-001.fasta
    r = run("coati-alignpair", fa, "-m", "mar-mg")
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["alignment"] == {"1": "CTCTGGATAGTG", "2": "CT----ATAGTG"}
    assert out["score"] == pytest.approx(1.50913, rel=1e-5)
    assert r.stdout.startswith('{\n  "alignment": {\n    "1": "CTCTGGATAGTG",')


def test_marg_alignment_doctest_cases(tmp_path):
    """src/lib/align_marginal.cc:149-240."""
    for k, case in enumerate(KNOWN["marg_alignment"]):
        names = case.get("names", ["1", "2"])
        fa = tmp_path / f"c{k}.fasta"
        fa.write_text("".join(f">{n}\n{s}\n" for n, s in zip(names, case["seqs"])))
        out = tmp_path / f"c{k}.out.fasta"
        args = [fa, "-m", case["model"], "-o", out]
        if "refs" in case:
            args += ["-r", case["refs"]]
        if case.get("rev"):
            args += ["-v"]
        if "gap_len" in case:
            args += ["-k", case["gap_len"]]
        if "amb" in case:
            args += ["-a", case["amb"]]
        r = run("coati-alignpair", *args)
        assert r.returncode == 0, (case, r.stderr)
        toks = out.read_text().split()
        want_names = case.get("out_names", names)
        assert toks == [">" + want_names[0], case["out"][0], ">" + want_names[1], case["out"][1]], case


def test_marg_alignment_failures(tmp_path):
    for k, case in enumerate(KNOWN["marg_alignment_fail"]):
        fa = tmp_path / f"f{k}.fasta"
        fa.write_text("".join(f">{i + 1}\n{s}\n" for i, s in enumerate(case["seqs"])))
        args = [fa]
        if "gap_len" in case:
            args += ["-k", case["gap_len"]]
        if "refs" in case:
            args += ["-r", case["refs"]]
        r = run("coati-alignpair", *args)
        assert r.returncode == 1 and r.stderr.startswith("ERROR:"), (case, r.stderr)


def test_phylip_output_and_stop_codons(tmp_path):
    fa = tmp_path / "s.fasta"
    fa.write_text(">anc\nGCGATTGCTGTTTGA\n>des\nGCGACTGTT\n")  # terminal stop only in the ancestor
    out = tmp_path / "s.phy"
    r = run("coati-alignpair", fa, "-o", out)
    assert r.returncode == 0, r.stderr
    lines = out.read_text().split("\n")
    assert lines[0] == "2 15" and lines[1] == "anc       GCGATTGCTGTTTGA" and lines[2] == "des       GCGA---CTGTT---"
    # score: -s prints the score of a given alignment; re-scoring the output reproduces the aligner's score
    js = tmp_path / "s.json"
    assert run("coati-alignpair", fa, "-o", js).returncode == 0
    score = json.loads(js.read_text())["score"]
    al = tmp_path / "al.fasta"
    al.write_text(">anc\nGCGATTGCTGTTTGA\n>des\nGCGA---CTGTT---\n")
    r = run("coati-alignpair", al, "-s")
    assert r.returncode == 0 and float(r.stdout.split()[-1]) == pytest.approx(score, rel=1e-5)


def test_batch_extension(tmp_path):
    fa = tmp_path / "b.fasta"
    fa.write_text(">a1\nCTCTGGATAGTG\n>b1\nCTATAGTG\n>a2\nGCGATTGCTGTT\n>b2\nGCGACTGTT\n")
    r = run("coati-alignpair", fa, "--batch")
    assert r.returncode == 0, r.stderr
    arr = json.loads(r.stdout)
    assert [list(x["alignment"].values()) for x in arr] == [["CTCTGGATAGTG", "CT----ATAGTG"], ["GCGATTGCTGTT", "GCGA---CTGTT"]]
    assert arr[0]["score"] == pytest.approx(1.50913, rel=1e-5) and arr[1]["score"] == pytest.approx(3.79779, rel=1e-5)


def test_sample_doctest_cases(tmp_path):
    """src/lib/align_marginal.cc:598-672: seed 42, exact JSON layout; scores to 1e-5 (device libm, own expm)."""
    for k, case in enumerate(KNOWN["marg_sample"]):
        fa = tmp_path / f"s{k}.fasta"
        fa.write_text(f">A\n{case['seqs'][0]}\n>B\n{case['seqs'][1]}\n")
        out = tmp_path / f"s{k}.json"
        r = run("coati-sample", fa, "-n", len(case["out"]), "-s", "42", "-o", out)
        assert r.returncode == 0, r.stderr
        text = out.read_text()
        lines = text.split("\n")
        assert lines[0] == "[" and lines[1] == "{" and lines[2] == '  "alignment": {'
        arr = json.loads(text)
        assert len(arr) == len(case["out"])
        for got, (wa, wb), ws in zip(arr, case["out"], case["scores"]):
            assert got["alignment"] == {"A": wa, "B": wb}
            assert got["score"] == pytest.approx(float(ws), rel=1e-5)
    # (round 6) --fast-forward: the C ABI's tolerance mode (COATI_HIP_OPT_FORWARD_MODE) -- the same samples on these inputs,
    # their log-weights within north_star's 1e-5 relative
    case = KNOWN["marg_sample"][0]
    out = tmp_path / "fast.json"
    r = run("coati-sample", tmp_path / "s0.fasta", "-n", len(case["out"]), "-s", "42", "--fast-forward", "-o", out)
    assert r.returncode == 0, r.stderr
    for got, (wa, wb), ws in zip(json.loads(out.read_text()), case["out"], case["scores"]):
        assert got["alignment"] == {"A": wa, "B": wb}
        assert got["score"] == pytest.approx(float(ws), rel=1e-5)
    assert run("coati-alignpair", tmp_path / "s0.fasta", "--fast-forward").returncode != 0  # (a flag of `sample` only)
    # failures (align_marginal.cc:673-722)
    bad = tmp_path / "bad.fasta"
    bad.write_text(">seq1\nAC\n>seq2\nACG\n")
    assert run("coati-sample", bad).returncode == 1
    one = tmp_path / "one.fasta"
    one.write_text(">A\nCCC\n")
    assert run("coati-sample", one).returncode == 1


def test_align_leafs_per_leaf_branch_lengths(oracle):
    """Host align_leafs (the pairwise step of `coati msa`, align_msa.cc:285-318, batched): each leaf is
    aligned with the table of its own branch length; equals per-leaf oracle runs bit for bit."""
    from coati_amd import host
    from tests import util

    rng = np.random.default_rng(8)
    ref = util.random_anc(rng, 120)
    leaves = [util.mutate(rng, ref, sub=0.03 + 0.02 * k, n_indel=2) for k in range(7)]
    br = [0.0133, 0.05, 0.0133, 0.2, 0.05, 0.8, 0.01]
    got = host.align_leafs(ref, leaves, br)
    consts = host.gap_consts()
    for (aln_ref, aln_leaf, score), leaf, t in zip(got, leaves, br):
        table = host.set_subst("mar-mg", br_len=t)
        a, b = util.encode_anc(ref), util.encode_des(leaf)
        w_ops, w_sc = oracle.viterbi(table, consts, 1, a, b)
        w_a, w_b = oracle.ops_to_strings(w_ops, ref, leaf)
        assert (aln_ref, aln_leaf) == (w_a, w_b)
        assert np.float32(score).view(np.uint32) == np.float32(w_sc).view(np.uint32)


def test_sample_independent_streams_flag(tmp_path):
    """coati-sample --independent-streams (build extension): first sample identical to the default
    mode's first sample; all samples valid alignments of the input pair."""
    fasta = tmp_path / "p.fasta"
    fasta.write_text(">A\nCTCTGGATAGTGACGACG\n>B\nCTATAGTGACGAG\n")
    outs = []
    for extra in ([], ["--independent-streams"]):
        r = subprocess.run([str(BIN / "coati-sample"), str(fasta), "-n", "6", "-s", "42"] + extra, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout))
    assert outs[0][0] == outs[1][0]
    for rec in outs[1]:
        a, b = rec["alignment"]["A"], rec["alignment"]["B"]
        assert len(a) == len(b) and a.replace("-", "") == "CTCTGGATAGTGACGACG" and b.replace("-", "") == "CTATAGTGACGAG"


MSA_SEQS = ">A\nTCATCG\n>B\nTCAGTCG\n>C\nTATCG\n>D\nTCACTCG\n>E\nTCATC\n"
MSA_WANT = {"A": "TCA--TCG", "B": "TCA-GTCG", "C": "T-A--TCG", "D": "TCAC-TCG", "E": "TCA--TC-"}


def run_msa(tmp_path, fasta, newick, ref, extra=()):
    (tmp_path / "in.fasta").write_text(fasta)
    (tmp_path / "tree.newick").write_text(newick)
    out = tmp_path / "msa.fasta"
    r = subprocess.run([str(BIN / "coati-msa"), str(tmp_path / "in.fasta"), str(tmp_path / "tree.newick"), ref, "-o", str(out),
                        *extra], capture_output=True, text=True, timeout=300)
    return r, out


@pytest.mark.parametrize("model", ["mar-mg", "mar-ecm"])
def test_msa_known_answer(tmp_path, model):
    """ref_indel_alignment doctests (align_msa.cc:135-195): 5 leaves, ladder tree."""
    r, out = run_msa(tmp_path, MSA_SEQS, "((((A:0.1,B:0.1):0.1,C:0.1):0.1,D:0.1):0.1,E:0.1);", "A", ["-m", model])
    assert r.returncode == 0, r.stderr
    toks = out.read_text().split()
    assert dict(zip([t[1:] for t in toks[0::2]], toks[1::2])) == MSA_WANT
    assert [t[1:] for t in toks[0::2]] == ["A", "B", "C", "D", "E"]  # input order


def test_msa_more_complex_tree(tmp_path):
    """align_msa.cc:217-258"""
    r, out = run_msa(tmp_path, MSA_SEQS + ">F\nTCATCG", "((A:0.1,B:0.1):0.1,(C:0.1,(D:0.1,E:0.1):0.1):0.1,F:0.1);\n", "A")
    assert r.returncode == 0, r.stderr
    toks = out.read_text().split()
    assert dict(zip([t[1:] for t in toks[0::2]], toks[1::2])) == dict(MSA_WANT, F="TCA--TCG")


def test_msa_errors(tmp_path):
    r, _ = run_msa(tmp_path, ">A\nTCATCG\n>B\nTCAGTCG\n", "(A:0.1,B:0.1);\n", "A")
    assert r.returncode != 0 and "At least three sequences required." in r.stderr  # align_msa.cc:197-215
    r, _ = run_msa(tmp_path, MSA_SEQS, "((((A:0.1,B:0.1):0.1,C:0.1):0.1,D:0.1):0.1,E:0.1);", "A", ["-m", "tri-mg"])
    assert r.returncode != 0 and "MSA only supports marginal models." in r.stderr  # align_msa.cc:260-265
    r, _ = run_msa(tmp_path, MSA_SEQS, "((((A:0.1,B:0.1):0.1,C:0.1):0.1,D:0.1):0.1,E:0.1);", "Z")
    assert r.returncode != 0 and "not found" in r.stderr


def test_user_rate_matrix_reference_doctest(tmp_path):
    """src/lib/align_marginal.cc:304-343, "User-provided codon substitution matrix": the mg94Q CSV through --sub gives
    the default model's alignment, written as FASTA."""
    from tests.test_host_io_cli import write_reference_rate_csv

    known = write_reference_rate_csv(tmp_path / "test-marg-matrix.csv")
    fa = tmp_path / "test-marg.fasta"
    fa.write_text(">1\n%s\n>2\n%s\n" % tuple(known["seqs"]))
    out = tmp_path / "test-marg_alignment-fasta.fasta"
    r = run("coati-alignpair", fa, "--sub", tmp_path / "test-marg-matrix.csv", "-o", out)
    assert r.returncode == 0, r.stderr
    assert out.read_text().split() == [">1", known["out"][0], ">2", known["out"][1]]
    # ... and the same score as the built-in model to the accuracy of the CSV's six significant digits
    js, js0 = tmp_path / "sub.json", tmp_path / "mg.json"
    assert run("coati-alignpair", fa, "--sub", tmp_path / "test-marg-matrix.csv", "-o", js).returncode == 0
    assert run("coati-alignpair", fa, "-m", "mar-mg", "-o", js0).returncode == 0
    assert json.loads(js.read_text())["score"] == pytest.approx(json.loads(js0.read_text())["score"], rel=1e-4)
    # a CSV with a line too many is refused (src/lib/io.cc:137-170)
    write_reference_rate_csv(tmp_path / "bad.csv", extra_line=True)
    r = run("coati-alignpair", fa, "--sub", tmp_path / "bad.csv")
    assert r.returncode == 1 and r.stderr.startswith("ERROR:")


def test_gtr_sigma_end_to_end(tmp_path, oracle):
    """-x / --sigma (src/lib/mutation_coati.cc:317-354, utils.cc:137): the CLI's alignment and score under a GTR
    nucleotide model equal the reference DP engine's on the table the build computes for those parameters."""
    from coati_amd import host
    from tests import util

    sigma = KNOWN["gtr_q"]["sigma"]
    rng = np.random.default_rng(5)
    anc = util.random_anc(rng, 60)
    des = util.mutate(rng, anc, sub=0.08, n_indel=3)
    fa = tmp_path / "gtr.fasta"
    fa.write_text(f">anc\n{anc}\n>des\n{des}\n")
    out = tmp_path / "gtr.json"
    r = run("coati-alignpair", fa, "-x", *sigma, "-o", out)
    assert r.returncode == 0, r.stderr
    got = json.loads(out.read_text())
    table = host.set_subst("mar-mg", sigma=sigma)
    assert np.abs(table - host.set_subst("mar-mg")).max() > 1e-3  # (the parameters do change the model)
    consts = host.gap_consts()
    a, b = util.encode_anc(anc), util.encode_des(des)
    ops, score = oracle.viterbi(table, consts, 1, a, b)
    want = oracle.ops_to_strings(ops, anc, des)
    assert (got["alignment"]["anc"], got["alignment"]["des"]) == want
    assert np.float32(got["score"]).view(np.uint32) == np.float32(score).view(np.uint32)
    if oracle.ref_available():  # the unmodified engine itself (oracle/_ref travels to the GPU box as a binary)
        g, e = np.float32(0.001), np.float32(1.0) - np.float32(1.0) / np.float32(6.0)
        _, _, _, sa, sb, sc = oracle.ref_viterbi(table, g, e, 1, anc, des, a, b)
        assert (sa, sb) == want and np.float32(sc).view(np.uint32) == np.float32(score).view(np.uint32)


def test_marg_sample_failures(tmp_path):
    """src/lib/align_marginal.cc:673-720: every failure subcase of marg_sample, through coati-sample."""
    for k, case in enumerate(KNOWN["marg_sample_fail"]):
        fa = tmp_path / f"sf{k}.fasta"
        fa.write_text("".join(f">{n}\n{s}\n" for n, s in zip(case["names"], case["seqs"])))
        args = [fa, "-n", 1]
        if "gap_len" in case:
            args += ["-k", case["gap_len"]]
        if "output" in case:
            args += ["-o", tmp_path / case["output"]]
        r = run("coati-sample", *args)
        assert r.returncode == 1 and r.stderr.startswith("ERROR:"), (case, r.returncode, r.stderr)


def test_batch_input_errors_are_reported_not_hung(tmp_path):
    """--batch brings the model up on a helper thread while the input is indexed: an input that fails (missing file,
    odd number of records) must unwind -- report and exit -- not wait for that thread for ever."""
    odd = tmp_path / "odd.fasta"
    odd.write_text(">a1\nCTCTGGATAGTG\n>b1\nCTATAGTG\n>a2\nGCGATTGCTGTT\n")
    for bad in (tmp_path / "missing.fasta", odd):
        r = subprocess.run([str(BIN / "coati-alignpair"), str(bad), "--batch"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and r.stderr.startswith("ERROR:"), (bad, r.returncode, r.stderr)
