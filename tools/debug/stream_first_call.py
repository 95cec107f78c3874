#!/usr/bin/env python3
"""ONE streamed call of a fresh process / model into fresh page-locked arrays, with the library's per-chunk trace on stderr
(COATI_HIP_PIPE_TIMING=1); prints the wall time.  usage: stream_first_call.py [n_pairs]"""
import os
import sys
import time
from pathlib import Path

os.environ["COATI_HIP_PIPE_TIMING"] = "1"
os.environ.setdefault("COATI_HIP_PIPE", "stream")
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coati_amd import hip, host  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10000
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
a = host.synth_encoded(0, n)
pa, pb = hip.pinned_copy(a[0]), hip.pinned_copy(a[2])
if "--after-chunks" in sys.argv:  # (tools/stream_probe.py's order: a resident batch and the chunk pipeline have run on the model before)
    batch = hip.Batch(model, *a)
    for r in range(3):
        batch.viterbi_launch()
        batch.viterbi_fetch()
    batch.close()
    os.environ["COATI_HIP_PIPE"] = "chunks"
    for pinned in (True, False):
        for r in range(3):
            model.viterbi(pa, a[1], pb, a[3], pinned=pinned)
    os.environ["COATI_HIP_PIPE"] = "stream"
    print("---- the streamed call", file=sys.stderr, flush=True)
t0 = time.perf_counter()
model.viterbi(pa, a[1], pb, a[3], pinned=True)
print("first call %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
