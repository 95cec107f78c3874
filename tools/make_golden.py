#!/usr/bin/env python3
"""Build-container tool: generate the committed fixtures under tests/golden/.

Everything here is DATA (inputs + expected outputs).  Expected outputs come from
the unmodified reference DP engine compiled into oracle/_ref (`make -C oracle
ref`), or are transcribed known answers of the reference's own doctest cases
(file:line given per entry).  Re-run only in a container that has
/root/reference; the GPU box just reads the JSON/NPY files.
"""
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import host  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from tests import util  # noqa: E402

REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
assert orc.ref_available(), "build oracle/_ref first (make -C oracle ref)"

G = np.float32(0.001)
E = np.float32(1.0) - np.float32(1.0) / np.float32(6.0)


def hexf(x):
    return "%08x" % int(np.float32(x).view(np.uint32))


# ---- 1. golden MG94 P matrix of the reference's tests (src/include/coati/mg94p.tcc:26) ----
text = re.sub(r"/\*.*?\*/", "", (REF / "src/include/coati/mg94p.tcc").read_text(), flags=re.S)
body = re.search(r"mg94P\[61\]\[61\]\s*=\s*\{(.*?)\};", text, flags=re.S).group(1)
P = np.array([float(x) for x in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", body)]).reshape(61, 61)
np.save(OUT / "mg94P_golden.npy", P.astype(np.float64))
# the table every DP fixture below was generated with: restated marginal_p over the golden P
table = host.marginal_p(P.astype(np.float32))
np.save(OUT / "table_mg94_goldenP.npy", table)

# ---- 1b. the MG94 rate matrix the reference's --sub / parse_matrix_csv doctests write into their CSV
# (src/include/coati/mg94q.tcc:26,77 scattered as align_marginal.cc:324-327 / io.cc:110-113 do): 61 x 61 fp32 VALUES
text = re.sub(r"/\*.*?\*/", "", (REF / "src/include/coati/mg94q.tcc").read_text(), flags=re.S)
num = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"
idx = [int(x) for x in re.findall(num, re.search(r"mg94_indexes\[587\]\s*=\s*\{(.*?)\};", text, flags=re.S).group(1))]
val = [float(x) for x in re.findall(num, re.search(r"mg94Q\[587\]\s*=\s*\{(.*?)\};", text, flags=re.S).group(1))]
assert len(idx) == 587 and len(val) == 587
Q = np.zeros(3721, np.float32)
Q[idx] = np.array(val, np.float64).astype(np.float32)  # (float Q[3721]: the literals are rounded to fp32 there)
np.save(OUT / "mg94Q_rate_matrix.npy", Q.reshape(61, 61))

# ---- 2. Viterbi cases through the reference ----
def read_fasta(path):
    names, seqs = [], []
    for line in Path(path).read_text().splitlines():
        if line.startswith(">"):
            names.append(line[1:])
            seqs.append("")
        elif line and not line.startswith(";") and names:
            seqs[-1] += "".join(line.split())
    return names, seqs


def trim_stop(s):
    return s[:-3] if len(s) >= 3 and s[-3:].upper().replace("U", "T") in ("TAA", "TAG", "TGA") else s


rng = np.random.default_rng(20241115)
cases = []
named = [
    ("example-001 (align_marginal.cc:149-157)", "CTCTGGATAGTG", "CTATAGTG", 1),
    ("example-002 / phylip case (align_marginal.cc:180-189)", "GCGATTGCTGTT", "GCGACTGTT", 1),
    ("2 dels (align_marginal.cc:201-210)", "ACGTTAAGGGGT", "ACGAAT", 1),
    ("gap len 3 (align_marginal.cc:211-221)", "ACGTTAAGGGGT", "ACGAAT", 3),
    ("ambiguous R (align_marginal.cc:222-230)", "CTCTGGATAGTG", "CTATAGTR", 1),
    ("micro example SURVEY appendix", "CTC", "CT", 1),
    ("empty both", "", "", 1),
    ("empty descendant", "ATGCCC", "", 1),
    ("empty ancestor", "", "ACGT", 1),
]
pairs = [(n, a, d, L) for n, a, d, L in named]
for k, (a, d) in enumerate(util.make_pairs(rng, 150, 1, 100, L=1, amb=0.03)):
    pairs.append((f"random L1 #{k}", a, d, 1))
for k, (a, d) in enumerate(util.make_pairs(rng, 50, 1, 60, L=3, amb=0.03)):
    pairs.append((f"random L3 #{k}", a, d, 3))
for cod in ("AAA", "CCC", "ACG"):
    pairs.append((f"poly-{cod}", cod * 40, cod * 33 + "A", 1))
for name in ("benchmark_156", "benchmark_1k", "benchmark_2k", "benchmark_4k"):
    _, seqs = read_fasta(REF / "benchmark" / "data" / f"{name}.fasta")
    a, d = trim_stop(seqs[0]), trim_stop(seqs[1])
    pairs.append((f"{name}.fasta after stop trimming (benchmark/data)", a.upper(), d.upper(), 1))
for p in range(16):
    a, d = host.synth_raw(p)
    pairs.append((f"synthetic config-2 pair {p}", a, d, 1))

for name, anc, des, L in pairs:
    a, b = util.encode_anc(anc), util.encode_des(des)
    small = (len(a) + L) * (len(b) + L) <= 900
    M, D, I, sa, sb, sc = orc.ref_viterbi(table, G, E, L, anc, des, a, b, want_matrices=True)
    entry = {"name": name, "anc": anc, "des": des, "gap_len": L, "aln_anc": sa, "aln_des": sb, "score_bits": hexf(sc),
             "score": float(sc)}
    if small:
        entry["M_bits"] = [hexf(v) for v in M.ravel()]
        entry["D_bits"] = [hexf(v) for v in D.ravel()]
        entry["I_bits"] = [hexf(v) for v in I.ravel()]
    cases.append(entry)
(OUT / "viterbi_cases.json").write_text(json.dumps({"gap_open": float(G), "gap_extend": float(E), "cases": cases}, indent=0))

# ---- 3. Forward + sampleback cases ----
scases = []
spairs = [("CCCCCC x CCCCCCCC (align_marginal.cc:653-672)", "CCCCCC", "CCCCCCCC", 1, ["42"], 3),
          ("CCCCCC x CCCC (align_marginal.cc:659-664)", "CCCCCC", "CCCC", 1, ["42"], 1)]
for k, (a, d) in enumerate(util.make_pairs(rng, 12, 2, 40, L=1)):
    spairs.append((f"random L1 #{k}", a, d, 1, ["42"] if k % 2 else [f"seed{k}"], 25))
for k, (a, d) in enumerate(util.make_pairs(rng, 4, 2, 30, L=3)):
    spairs.append((f"random L3 #{k}", a, d, 3, ["7", "x"], 10))
for p in range(2):
    a, d = host.synth_raw(100 + p)
    spairs.append((f"synthetic pair {100 + p}", a, d, 1, ["42"], 100))
for name, anc, des, L, seeds, n in spairs:
    a, b = util.encode_anc(anc), util.encode_des(des)
    mats, alns, scores = orc.ref_forward_sample(table, G, E, L, anc, des, a, b, seeds, n)
    scases.append({"name": name, "anc": anc, "des": des, "gap_len": L, "seeds": seeds,
                   "final_M_bits": hexf(mats[0, -1, -1]), "final_D_bits": hexf(mats[1, -1, -1]),
                   "final_I_bits": hexf(mats[2, -1, -1]),
                   "checksum_MDI": [float(np.float64(mats[q]).clip(-1e30, None).sum()) for q in range(3)],
                   "samples": [{"anc": sa, "des": sb, "score_bits": hexf(s)} for (sa, sb), s in zip(alns, scores)]})
(OUT / "sample_cases.json").write_text(json.dumps({"gap_open": float(G), "gap_extend": float(E), "cases": scases}, indent=0))

# ---- 4. RNG streams ----
rngs = [{"seeds": s, "f24_bits": [hexf(v) for v in orc.ref_rng_f24(s, 16)]}
        for s in (["42"], [""], ["random42"], ["1", "2", "abc"], ["-17"], ["2147483648"])]
(OUT / "rng_streams.json").write_text(json.dumps(rngs, indent=0))

# ---- 5. transcribed known answers of the reference's own tests ----
known = {
    "marg_alignment": [  # src/lib/align_marginal.cc:149-240,304-343 : (input seqs, model, options) -> output
        {"seqs": ["CTCTGGATAGTG", "CTATAGTG"], "model": "mar-mg", "out": ["CTCTGGATAGTG", "CT----ATAGTG"]},
        {"seqs": ["CTATAGTG", "CTCTGGATAGTG"], "names": ["1", "2"], "refs": "2", "model": "mar-mg",
         "out_names": ["2", "1"], "out": ["CTCTGGATAGTG", "CT----ATAGTG"]},
        {"seqs": ["CTCTGGATAGTG", "CTATAGTG"], "names": ["1", "2"], "refs": "1", "model": "mar-mg",
         "out_names": ["1", "2"], "out": ["CTCTGGATAGTG", "CT----ATAGTG"]},
        {"seqs": ["GCGACTGTT", "GCGATTGCTGTT"], "model": "mar-mg", "out": ["GCGA---CTGTT", "GCGATTGCTGTT"]},
        {"seqs": ["GCGATTGCTGTT", "GCGACTGTT"], "names": ["A", "B"], "rev": True, "model": "mar-ecm",
         "out_names": ["B", "A"], "out": ["GCGA---CTGTT", "GCGATTGCTGTT"]},
        {"seqs": ["ACGTTAAGGGGT", "ACGAAT"], "model": "mar-mg", "out": ["ACGTTAAGGGGT", "ACG--AA----T"]},
        {"seqs": ["ACGTTAAGGGGT", "ACGAAT"], "model": "mar-mg", "gap_len": 3, "out": ["ACGTTAAGGGGT", "AC------GAAT"]},
        {"seqs": ["CTCTGGATAGTG", "CTATAGTR"], "model": "mar-mg", "out": ["CTCTGGATAGTG", "CT----ATAGTR"]},
        {"seqs": ["CTCTGGATAGTG", "CTATAGTR"], "model": "mar-mg", "amb": "BEST", "out": ["CTCTGGATAGTG", "CT----ATAGTR"]},
    ],
    "marg_alignment_fail": [  # align_marginal.cc:241-262,291-301,344-361
        {"seqs": ["GCGATTGCTGT", "GCGACTGTT"], "gap_len": 3}, {"seqs": ["CTCGGA", "CTCGG"], "gap_len": 3},
        {"seqs": ["CTCTGGATAGTG", "CTATAGTG"], "refs": "seq_name"}, {"seqs": ["CTCTGGATAGTG"]},
        {"seqs": ["CTCTGGATAGTG", "CTATAGTG", "CTCTGGGTG"]},
    ],
    "alignment_score": [  # align_marginal.cc:489-508 (doctest::Approx, eps 1e-5)
        ["CTCTGGATAGTG", "CT----ATAGTG", 1.50914], ["CTCT--AT", "CTCTGGAT", -0.83906], ["ACTCT-A", "ACTCTG-", -10.52864],
        ["ATGCTTTAC", "ATGCT-TAC", 2.13593], ["ATGCTT---", "ATGCTTTGA", 0.70607], ["A-CTAAC", "ACCTAAG", -8.2786],
        ["ACT---", "ACTCTG", -5.04197], ["ACTCTA", "ACT---", -5.04197], ["ACT----", "ACT-CTG", -5.04197],
        ["AAAAAA---AAA", "AAA---AAAAAA", -11.09557], ["AAA---AAAAAA", "AAAAAA---AAA", -11.09557],
        ["AAA-A-A-AAAA", "AAAA-A-A-AAA", -11.09557], ["---AAAAAA", "AAAAAAAAA", -2.03242],
        ["AAAAAA---", "AAAAAAAAA", -2.03242], ["AAAAAAAAA", "---AAAAAA", -2.03242], ["AAAAAAAAA", "AAAAAA---", -2.03242],
        ["ACTCTA", "ACTC--", -3.18537], ["ACTCTA-", "ACTCTAG", -10.45777], ["ACTCTA--", "ACTCT-AG", -10.45777],
    ],
    "marg_sample": [  # align_marginal.cc:653-672 (seed "42"; exact JSON score strings)
        {"seqs": ["CCCCCC", "CCCCCCCC"], "out": [["CC--CCCC", "CCCCCCCC"]], "scores": ["-1.9466571807861328"]},
        {"seqs": ["CCCCCC", "CCCC"], "out": [["CCCCCC", "--CCCC"]], "scores": ["-1.6172490119934082"]},
        {"seqs": ["CCCCCC", "CCCCCCCC"], "out": [["CC--CCCC", "CCCCCCCC"], ["CCCCCC--", "CCCCCCCC"], ["CCCC--CC", "CCCCCCCC"]],
         "scores": ["-1.9466571807861328", "-1.9466569423675537", "-1.9466572999954224"]},
    ],
    "marg_sample_fail": [  # align_marginal.cc:673-720: every subcase throws std::invalid_argument
        {"what": "length of reference not multiple of 3", "names": ["seq1", "seq2"], "seqs": ["AC", "ACG"]},
        {"what": "length of descendant no multiple of gap len", "names": ["A", "B"], "seqs": ["CCC", "CCCC"], "gap_len": 3},
        {"what": "error opening output file", "names": ["A", "B"], "seqs": ["CCC", "CCC"], "output": "no-such-directory/x.json"},
        {"what": "Number of seqs != 2", "names": ["A"], "seqs": ["CCC"]},
        {"what": "Number of seqs != 2", "names": ["A", "B", "C"], "seqs": ["CCC", "CCC", "CCC"]},
    ],
    "user_matrix": {  # align_marginal.cc:304-343 and io.cc:92-133: the CSV holds mg94Q (tests/golden/mg94Q_rate_matrix.npy) and this branch length
        "br_len": "0.0133", "omega": 0.2, "pi": [0.308, 0.185, 0.199, 0.308],
        "seqs": ["CTCTGGATAGTG", "CTATAGTG"], "out": ["CTCTGGATAGTG", "CT----ATAGTG"]},
    "marginal_seq_encoding": {  # src/lib/utils.cc:532-586
        "anc": "AAAGGGTTTCCCACTAGA", "anc_codes": [0, 1, 2, 126, 127, 128, 180, 181, 182, 63, 64, 65, 21, 22, 23, 24, 25, 26],
        "des": "ACGTRYMKSWBDHVN-", "des_codes": list(range(16)),
        "anc_fail": ["AAACCCGGN", "AAACCCGGR", "YAACCCGGG", "AAATAA", "AAATAGGCC", "TGA"]},
    "trim_end_stops": [  # src/lib/utils.cc:970-993 : raw seqs, trimmed seqs, stops
        [["AAA", "CCC"], ["AAA", "CCC"], ["", ""]], [["AAATAA", "AAATTT"], ["AAA", "AAATTT"], ["TAA", ""]],
        [["AAATTT", "AAATAG"], ["AAATTT", "AAA"], ["", "TAG"]], [["AAATGA", "AAAuga"], ["AAA", "AAA"], ["TGA", "uga"]],
        [["AAATAA", "AAATAG"], ["AAA", "AAA"], ["TAA", "TAG"]], [["AAA", "C"], ["AAA", "C"], ["", ""]],
        [["AAATGA", "C"], ["AAA", "C"], ["TGA", ""]], [["AAA", "ctaa"], ["AAA", "c"], ["", "taa"]]],
    "restore_end_stops": [  # src/lib/utils.cc:1067-1090 : aligned seqs, stops, expected
        [["AAA", "AAA"], ["TAA", "TAA"], ["AAATAA", "AAATAA"]], [["", ""], ["TAA", "TAA"], ["TAA", "TAA"]],
        [["CGA", "CGA"], ["", ""], ["CGA", "CGA"]], [["CTA", "CTA"], ["TAG", "TGA"], ["CTATAG", "CTATGA"]],
        [["TGC", "TGC"], ["", "TAA"], ["TGC---", "TGCTAA"]], [["TGC---", "TGCCAC"], ["", "TAA"], ["TGC------", "TGCCACTAA"]],
        [["CGG", "CGG"], ["TAG", ""], ["CGGTAG", "CGG---"]]],
    "cod64_to_61": [[0, 0], [20, 20], [47, 47], [49, 48], [51, 49], [52, 50], [53, 51], [57, 54], [60, 57], [63, 60]],  # utils.cc:1168-1178
    "gtr_q": {  # mutation_coati.cc:358-372
        "pi": [0.308, 0.185, 0.199, 0.308],
        "sigma": [0.009489730, 0.039164824, 0.004318182, 0.015438693, 0.038734091, 0.008550000],
        "expected": [[-0.010879400, 0.001755600, 0.00779380, 0.00133000], [0.002922837, -0.017925237, 0.00307230, 0.01193010],
                     [0.012062766, 0.002856158, -0.01755232, 0.00263340], [0.001330000, 0.007165807, 0.00170145, -0.01019726]]},
}
(OUT / "reference_known_answers.json").write_text(json.dumps(known, indent=1))
print("fixtures written to", OUT, [f"{p.name}:{p.stat().st_size}" for p in sorted(OUT.iterdir())])
