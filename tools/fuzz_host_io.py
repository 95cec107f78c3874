#!/usr/bin/env python3
"""Mutation fuzzer for the host layer's parsers and string routines (FASTA / PHYLIP / JSON readers and
writers, Newick, sequence encoding, stop trimming, rescoring, rate-matrix CSV).  CPU only.  Meant to
run against the sanitizer build of the host library:

    make asan
    COATI_HOST_LIB=coati_amd/_build/asan/libcoati_host.so \
      LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
      ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_host_io.py [iterations] [seed]

Every call must either succeed or raise CoatiHostError; a sanitizer report or a crash is a bug."""
import os, random, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import host
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
tmp = tempfile.mkdtemp()
seeds = {
 "fa": ">a desc\nACGTACGT\nAC-T\n;comment\n>b\nACG\n",
 "phy": "2 12\nseq1      ACGTACGTACGT\nseq2      ACGTAC--ACGT\n",
 "json": '{"alignment": {"a": "ACGT", "b": "AC-T"}, "score": -1.5}',
}
nwk = ["((A:0.1,B:0.2):0.3,C:0.4);", "(A,B,(C,D)E)F;", "((raccoon:19.2,bear:6.8):0.85,((sea_lion:12,seal:12):7.5,weasel:18.9):2.1,dog:25.5);"]
def mutate(s):
    s = list(s)
    for _ in range(random.randint(1, 6)):
        op = random.random()
        if not s: s = list("x")
        i = random.randrange(len(s))
        if op < 0.3: del s[i]
        elif op < 0.6: s.insert(i, random.choice("()[]{}:;,>\n\t \"'-0123456789.eE+ACGTN\\\x00\xff"))
        elif op < 0.8: s[i] = random.choice("()[]{}:;,>\n \"-09.ACGT")
        else: s[i:i] = s[max(0, i - random.randint(1, 20)):i]
    return "".join(s)
n = errs = 0
for it in range(iters):
    ext = random.choice(list(seeds))
    body = mutate(seeds[ext])
    pin = os.path.join(tmp, f"in.{ext}")
    open(pin, "w", encoding="latin-1").write(body)
    for oext in ("fa", "phy", "json"):
        try:
            host.convert(pin, os.path.join(tmp, f"out.{oext}"))
        except host.CoatiHostError:
            errs += 1
        n += 1
    t = mutate(random.choice(nwk))
    try:
        host.newick(t.replace("\x00", ""), random.choice(["", "A", "C", "zz"]))
    except (host.CoatiHostError, ValueError):
        errs += 1
    try:
        host.tree_distance(t.replace("\x00", ""), "A", random.choice(["B", "C", "D"]))
    except (host.CoatiHostError, ValueError):
        errs += 1
    try:
        host.encode(mutate("ACGTTTAAGCCC").replace("\x00", ""), mutate("ACGNRYTT").replace("\x00", ""))
    except (host.CoatiHostError, ValueError):
        errs += 1
    try:
        host.trim_end_stops(mutate("ACGTAA").replace("\x00", ""), mutate("ACGTGA").replace("\x00", ""))
    except (host.CoatiHostError, ValueError):
        errs += 1
    try:
        host.alignment_score(mutate("CTCTGGATAGTG").replace("\x00", ""), mutate("CT----ATAGTG").replace("\x00", ""),
                             gap_len=random.choice([1, 1, 2, 3]))
    except (host.CoatiHostError, ValueError):
        errs += 1
    try:
        host.restore_end_stops(mutate("ACG---").replace("\x00", ""), mutate("ACGTTT").replace("\x00", ""),
                               random.choice(["", "TAA", "TAG"]), random.choice(["", "TGA"]))
    except (host.CoatiHostError, ValueError):
        errs += 1
    csv = os.path.join(tmp, "m.csv")
    open(csv, "w", encoding="latin-1").write(mutate("0.0133\nAAA,AAC,0.001\nAAC,AAA,0.002\nTTT,TTC,0.5\n"))
    try:
        host.parse_matrix_csv(csv)
    except (host.CoatiHostError, ValueError):
        errs += 1
print("calls", n, "errors raised cleanly", errs)
