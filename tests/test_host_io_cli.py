"""Host layer: I/O formats, alignment_score known answers, CLI argument handling
(no GPU needed: the alignment itself is exercised in test_gpu_cli.py)."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest
import scipy.linalg

from coati_amd import host

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
KNOWN = json.loads((GOLD / "reference_known_answers.json").read_text())
BIN = ROOT / "coati_amd" / "_build"


@pytest.mark.host_answers
def test_extract_file_type_known_answers():
    """src/lib/utils.cc:657-676."""
    for path, want in (("foo.bar", ("foo.bar", ".bar")), ("my:foo.bar", ("foo.bar", ".my")), (".bar", (".bar", "")),
                       (".", (".", "")), ("..", ("..", "")), ("my:.foo.bar", (".foo.bar", ".my")),
                       (".foo.bar", (".foo.bar", ".bar")), ("", ("", "")), ("foo:-", ("-", ".foo")), ("foo:bar", ("bar", ".foo")),
                       ("bar:", ("", ".bar")), ("c:foo.bar", ("c:foo.bar", ".bar")),
                       (" \f\n\r\t\vfoo.bar \f\n\r\t\v", ("foo.bar", ".bar")), (" \f\n\r\t\vmy:foo.bar \f\n\r\t\v", ("foo.bar", ".my")),
                       (" \f\n\r\t\v", ("", ""))):
        assert host.extract_file_type(path) == want, path


def test_fasta_reader_and_writers(tmp_path):
    src = tmp_path / "in.fasta"
    src.write_text("; comment\njunk before\n>seq one with spaces\nCTCTGG ATA\nGTC\n\n>2\nCTATAGTC\n")
    host.convert(src, tmp_path / "out.fa")
    assert (tmp_path / "out.fa").read_text() == ">seq one with spaces\nCTCTGGATAGTC\n>2\nCTATAGTC\n"
    long = tmp_path / "long.fasta"
    long.write_text(">1\n" + "A" * 100 + "\n>2\n" + "A" * 100 + "\n")
    host.convert(long, tmp_path / "wrap.fasta")
    assert (tmp_path / "wrap.fasta").read_text() == ">1\n" + "A" * 60 + "\n" + "A" * 40 + "\n>2\n" + "A" * 60 + "\n" + "A" * 40 + "\n"
    # phylip output (src/lib/phylip.cc:253-272): 10-char names, first block 50 columns, then 60
    host.convert(long, tmp_path / "out.phy")
    lines = (tmp_path / "out.phy").read_text().split("\n")
    assert lines[0] == "2 100" and lines[1] == "1         " + "A" * 50 and lines[2] == "2         " + "A" * 50
    assert lines[3] == "" and lines[4] == "A" * 50 and lines[5] == "A" * 50
    # and back
    host.convert(tmp_path / "out.phy", tmp_path / "back.fa")
    assert (tmp_path / "back.fa").read_text() == (tmp_path / "wrap.fasta").read_text()
    bad = tmp_path / "bad.fasta"
    bad.write_text(">\nCTC\n>2\nCTA\n")
    with pytest.raises(host.CoatiHostError, match="without a name"):
        host.convert(bad, tmp_path / "x.fa")
    (tmp_path / "in.txt").write_text(">1\nAAA\n")
    with pytest.raises(host.CoatiHostError, match="Invalid input"):
        host.convert(tmp_path / "in.txt", tmp_path / "x.fa")
    with pytest.raises(host.CoatiHostError, match="Opening input file"):
        host.convert(tmp_path / "missing.fa", tmp_path / "x.fa")
    with pytest.raises(host.CoatiHostError, match="Invalid output format"):
        host.convert(src, tmp_path / "x.txt")


def test_json_format(tmp_path):
    """src/lib/json.cc:176-199: 2-space indent, "score": 0.0; sample arrays json.cc:232-290."""
    src = tmp_path / "in.fa"
    src.write_text(">a\nATGTCTTCTCACAAGACA\n>b\nATGTCTTCTCACAAGACA\n")
    host.convert(src, tmp_path / "out.json", 0.0)
    want = '{\n  "alignment": {\n    "a": "ATGTCTTCTCACAAGACA",\n    "b": "ATGTCTTCTCACAAGACA"\n  },\n  "score": 0.0\n}\n'
    assert (tmp_path / "out.json").read_text() == want
    host.convert(tmp_path / "out.json", tmp_path / "rt.fa")
    assert (tmp_path / "rt.fa").read_text() == src.read_text()
    host.write_json_array(src, tmp_path / "arr.json", 2)
    body = want.rstrip("\n")
    assert (tmp_path / "arr.json").read_text() == "[\n" + body + ",\n" + body + "\n]\n"
    assert json.loads((tmp_path / "arr.json").read_text())[1]["score"] == 0.0
    (tmp_path / "noscore.json").write_text('{"alignment": {"a": "AAA", "b": "AAA"}}')
    with pytest.raises(host.CoatiHostError):
        host.convert(tmp_path / "noscore.json", tmp_path / "x.fa")
    # numbers print like nlohmann::json prints a float widened to double
    assert host.json_number(np.float32(-1.9466571807861328)) == "-1.9466571807861328"
    assert host.json_number(0.0) == "0.0" and host.json_number(2.0) == "2.0" and host.json_number(1.5) == "1.5"


@pytest.mark.host_answers
def test_alignment_score_known_answers():
    """align_marginal.cc:489-508: 19 scores, doctest::Approx."""
    for anc, des, want in KNOWN["alignment_score"]:
        got = host.alignment_score(anc, des)
        assert abs(got - want) < 1.19e-5 * (1 + max(abs(got), abs(want))), (anc, des, got, want)
    with pytest.raises(host.CoatiHostError):
        host.alignment_score("ATAC", "ATA-")  # reference length not a multiple of 3
    with pytest.raises(host.CoatiHostError):
        host.alignment_score("ATACGG", "ATA")  # unequal lengths


def test_alignment_score_equals_viterbi_score_on_golden_alignments(oracle):
    """Rescoring the reference's own optimal alignments reproduces their Viterbi score (same model)."""
    doc = json.loads((GOLD / "viterbi_cases.json").read_text())
    n = 0
    for c in doc["cases"]:
        if c["gap_len"] != 1 or not (0 < len(c["anc"]) <= 200) or set(c["des"]) - set("ACGT"):
            continue
        if c["des"][-3:] in ("TAA", "TAG", "TGA"):
            continue  # alignment_score blanks a terminal stop codon and charges a gap for it
        cols = ["M" if x != "-" and y != "-" else ("D" if y == "-" else "I") for x, y in zip(c["aln_anc"], c["aln_des"])]
        runs = "".join(cols).replace("M", " ").split()
        if any("D" in r and "I" in r for r in runs):
            continue  # mixed gap runs are scored with their own closed form (align_marginal.cc:430-434)
        got = host.alignment_score(c["aln_anc"], c["aln_des"])
        assert got == pytest.approx(c["score"], rel=2e-5, abs=2e-5), c["name"]
        n += 1
    assert n > 50


@pytest.mark.host_answers
def test_user_rate_matrix_csv(tmp_path):
    """--sub (io.cc:48-88): branch length, then 3721 'cod,cod,rate' lines -> exp(Q t)."""
    sense = [c for c in range(64) if c not in (48, 50, 56)]
    cod = lambda c: "ACGT"[(c >> 4) & 3] + "ACGT"[(c >> 2) & 3] + "ACGT"[c & 3]
    rng = np.random.default_rng(2)
    Q = rng.uniform(0, 0.05, (61, 61))
    np.fill_diagonal(Q, 0)
    np.fill_diagonal(Q, -Q.sum(1))
    path = tmp_path / "q.csv"
    with path.open("w") as f:
        f.write("0.5\n")
        for i in range(61):
            for j in range(61):
                f.write(f"{cod(sense[i])},{cod(sense[j])},{Q[i, j]:.9g}\n")
    P = host.parse_matrix_csv(path)
    assert np.abs(P - scipy.linalg.expm(np.float32(Q).astype(np.float64) * 0.5)).max() < 1e-6
    short = tmp_path / "short.csv"
    short.write_text("0.5\nAAA,AAC,0.1\n")
    with pytest.raises(host.CoatiHostError):
        host.parse_matrix_csv(short)


def write_reference_rate_csv(path, extra_line=False):
    """The CSV of the reference's own doctests (src/lib/io.cc:105-125, src/lib/align_marginal.cc:319-336): branch length
    0.0133, then codon,codon,rate for the 61 x 61 sense codons in table order, rates = mg94Q (src/include/coati/mg94q.tcc;
    the VALUES are the fixture tests/golden/mg94Q_rate_matrix.npy) printed the way `ostream << float` prints them (%g)."""
    known = json.loads((GOLD / "reference_known_answers.json").read_text())["user_matrix"]
    Q = np.load(GOLD / "mg94Q_rate_matrix.npy")
    assert Q.shape == (61, 61) and Q.dtype == np.float32 and np.count_nonzero(Q) == 587
    cod = lambda c: "ACGT"[(c >> 4) & 3] + "ACGT"[(c >> 2) & 3] + "ACGT"[c & 3]
    sense = [cod(c) for c in range(64) if c not in (48, 50, 56)]
    with open(path, "w") as f:
        f.write(known["br_len"] + "\n")
        for i in range(61):
            for j in range(61):
                f.write("%s,%s,%g\n" % (sense[i], sense[j], float(Q[i, j])))
        if extra_line:
            f.write("%s,%s,%g\n" % (sense[0], sense[0], float(Q[0, 0])))
    return known


def doctest_approx(a, b):
    """doctest::Approx (contrib/doctest/doctest/doctest.h:3545,3565-3569): |a - b| < 100 * FLT_EPSILON * (1 + max(|a|, |b|))."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) < float(np.finfo(np.float32).eps) * 100 * (1.0 + np.maximum(np.abs(a), np.abs(b)))


@pytest.mark.host_answers
def test_parse_matrix_csv_reference_doctest(tmp_path):
    """src/lib/io.cc:92-172: parse_matrix_csv of the mg94Q CSV == mg94_p(0.0133, 0.2, pi) entry by entry (Approx);
    a file that cannot be opened and a file with one line too many are errors."""
    known = write_reference_rate_csv(tmp_path / "test-marg-matrix.csv")
    P_test = host.parse_matrix_csv(tmp_path / "test-marg-matrix.csv")
    P = host.mg94_p(float(known["br_len"]), known["omega"], known["pi"])
    assert P.shape == P_test.shape == (61, 61)
    assert doctest_approx(P, P_test).all(), float(np.abs(P - P_test).max())
    assert np.abs(P.astype(np.float64) - P_test).max() < 2e-6  # (what the build actually achieves: far inside Approx)
    with pytest.raises(host.CoatiHostError):
        host.parse_matrix_csv("")
    write_reference_rate_csv(tmp_path / "too-many.csv", extra_line=True)
    with pytest.raises(host.CoatiHostError):
        host.parse_matrix_csv(tmp_path / "too-many.csv")


def run(binary, *args):
    return subprocess.run([str(BIN / binary), *args], capture_output=True, text=True, timeout=120)


def test_cli_usage_errors(tmp_path):
    assert "Usage" in run("coati-alignpair", "--help").stdout
    assert run("coati-alignpair").returncode != 0  # input required
    fa = tmp_path / "p.fasta"
    fa.write_text(">1\nCTCTGGATAGTG\n>2\nCTATAGTG\n")
    for bad in (["-t", "-1"], ["-g", "0"], ["--nope"], ["-k", "0"], ["-a", "WORST"], ["-m", "mar-mg", "--sub", "x.csv"],
                ["-r", "1", "-v"], ["-p", "0.1", "0.2"]):
        r = run("coati-alignpair", str(fa), *bad)
        assert r.returncode != 0, bad
    # models of the FST aligner are not served by this build
    r = run("coati-alignpair", str(fa), "-m", "tri-mg")
    assert r.returncode == 1 and "ERROR" in r.stderr
    r = run("coati-sample", str(fa), "-m", "dna")
    assert r.returncode == 1 and "Sampling only available" in r.stderr


def test_writers_reject_degenerate_input_cleanly(tmp_path):
    """Inputs on which the reference's writers index out of range (phylip.cc:194-215 reads seqs[0] of
    an empty set; ragged blocks call substr past the end) must come back as errors, not crashes."""
    empty = tmp_path / "empty.fa"
    empty.write_text("; nothing but a comment\n")
    with pytest.raises(host.CoatiHostError):
        host.convert(str(empty), str(tmp_path / "out.phy"))
    ragged = tmp_path / "ragged.fa"
    ragged.write_text(">a\n" + "ACGT" * 30 + "\n>b\nACG\n")
    with pytest.raises(host.CoatiHostError):
        host.convert(str(ragged), str(tmp_path / "out2.phy"))
    # non-ASCII bytes in a descendant encode as the invalid code 16 (the reference indexes past its
    # 128-entry table there, utils.cc:496-528); coati_hip_batch_create rejects codes 15/16
    _, des = host.encode("ACG", "ACé")
    assert des.tolist() == [0, 1, 16, 16]


def test_batch_mode_reports_the_first_bad_pair(tmp_path):
    """`--batch` prepares its pairs on several threads (host/align.cc: parallel_for); the error that
    comes back must be the one a serial loop would hit first, whatever the threads' timing.
    (Validation happens before anything touches the GPU, so this runs on CPU.)"""
    rng = np.random.default_rng(8)
    good = lambda: ("".join(rng.choice(list("ACG"), 300)), "".join(rng.choice(list("ACGT"), 280)))  # noqa: E731 (no T: no stop codons)
    recs = []
    for p in range(2000):
        a, d = good()
        if p == 700:
            a = a[:150] + "NNN" + a[153:]  # ambiguous ancestor: the first error in file order
        if p in (40 + 700, 1500, 1999):
            a = a[:30] + "TAA" + a[33:]    # early stop codons further down
        recs.append(f">a{p}\n{a}\n>d{p}\n{d}\n")
    fa = tmp_path / "many.fasta"
    fa.write_text("".join(recs))
    for _ in range(3):
        r = run("coati-alignpair", str(fa), "--batch")
        assert r.returncode == 1 and "Ambiguous nucleotides in ancestor" in r.stderr, r.stderr[-300:]


def test_batch_driver_reader_equals_read_fasta(tmp_path):
    """`coati-alignpair --batch` indexes a FASTA file in memory (record starts found in parallel, records parsed
    per pair on the thread pool); every record must come out exactly as io.cc:read_fasta (fasta.cc:38-72 upstream)
    returns it: comment and empty lines, CRLF, white space inside sequence lines, '>' inside a line, text before the
    first record, no final newline, an empty sequence."""
    import numpy as np

    cases = {
        "plain.fasta": ">a\nACGT\nACG\n>b\nTTT\n",
        "crlf.fasta": ">a name\r\nAC GT\r\n\r\nAC\tG\r\n>b\r\nTT T\r\n",
        "comments.fa": "; header comment\n>a\n;inside\nACG\n\n;x\nTTT\n>b\n\n>c\nA>C\nG>T\n",
        "junk_first.fasta": "leading text\nmore\n>a\nACG\n>b\nTGA",
        "nofinal.fasta": ">a\nACG\n>b\nTT",
        "empty_seq.fasta": ">a\n>b\nACG\n>c\n\n\n",
    }
    for name, text in cases.items():
        p = tmp_path / name
        p.write_bytes(text.encode())
        assert host.batch_reader_check(p) == 0, name
    # a large random file: many records, random line lengths, random blank/comment lines
    rng = np.random.default_rng(3)
    parts = []
    for r in range(3000):
        parts.append(f">seq{r} d={rng.integers(0, 99)}\n")
        seq = "".join(rng.choice(list("ACGT"), int(rng.integers(0, 400))))
        while seq:
            k = int(rng.integers(1, 90))
            parts.append(seq[:k] + ("\r\n" if rng.random() < 0.1 else "\n"))
            seq = seq[k:]
            if rng.random() < 0.05:
                parts.append(";c\n" if rng.random() < 0.5 else "\n")
    big = tmp_path / "big.fasta"
    big.write_text("".join(parts))
    assert host.batch_reader_check(big) == 0
    # not a FASTA path: declined (the generic reader takes it)
    other = tmp_path / "x.json"
    other.write_text("{}")
    assert host.batch_reader_check(other) == -1
    missing = tmp_path / "nope.fasta"
    with pytest.raises(Exception):
        host.batch_reader_check(missing)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_multi_gpu_driver_parses_only_its_shard(tmp_path, world):
    """`coati-alignpair --batch --devices ...`, host side of every rank without a device (align.cc:
    marg_alignment_batch_dist; what the reference does per pair: src/lib/align_marginal.cc:44-88): the shard plan
    comes from the file's index alone, so every rank derives the same bounds; the shards tile the input, are balanced
    by (record size a) x (record size b), and a rank's shard parsed + processed + encoded through the block pipeline
    equals the generic reader's records (stop codons trimmed and remembered, codes)."""
    import numpy as np

    rng = np.random.default_rng(11)
    stops = ["TAA", "TAG", "TGA"]
    recs = []
    weights = []
    for p in range(173):
        n_cod = int(rng.integers(1, 120)) * (4 if p % 17 == 0 else 1)  # a few heavy pairs
        anc = "".join(rng.choice(list("ACGT"), 3 * n_cod))
        for s in stops:  # no premature stop codons in the ancestor (process_marginal refuses them)
            anc = "".join(c if c != s else "GCA" for c in (anc[i:i + 3] for i in range(0, len(anc), 3)))
        des = list(anc)
        for _ in range(int(rng.integers(0, 6))):
            des[int(rng.integers(0, len(des)))] = str(rng.choice(list("ACGTN")))
        des = "".join(des)
        if rng.random() < 0.3:
            anc += str(rng.choice(stops))
            des += str(rng.choice(stops))
        a_rec = f">anc{p}\n" + "\n".join(anc[i:i + 60] for i in range(0, len(anc), 60)) + "\n"
        d_rec = f">des{p} x\n" + "\n".join(des[i:i + 71] for i in range(0, len(des), 71)) + "\n"
        recs += [a_rec, d_rec]
        weights.append(len(a_rec) * len(d_rec))
    path = tmp_path / "pairs.fasta"
    path.write_text("".join(recs))
    bounds = []
    for rank in range(world):
        s0, s1, diff = host.batch_shard_check(path, world, rank)
        assert diff == 0, (rank, diff)
        bounds.append((s0, s1))
    assert bounds[0][0] == 0 and bounds[-1][1] == 173
    assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    w = np.array(weights, np.float64)
    shares = np.array([w[s0:s1].sum() for s0, s1 in bounds])
    assert shares.max() - w.sum() / world <= w.max() + 1
