"""Timing of the one-shot streamed call (page-locked arrays, outputs reused) in both pipeline forms.
usage: python tools/stream_probe.py [n_pairs] [reps]   (COATI_HIP_PIPE_TIMING=1 for the per-chunk log of the last call)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coati_amd import hip, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
table, consts = host.set_subst("mar-mg"), host.gap_consts()
model = hip.Model(table, consts, 1)
a = host.synth_encoded(0, n)
pa, pb = hip.pinned_copy(a[0]), hip.pinned_copy(a[2])
want_timing = os.environ.pop("COATI_HIP_PIPE_TIMING", None)
batch = hip.Batch(model, *a)
for r in range(3):
    batch.viterbi_launch()
    batch.viterbi_fetch()
ts = []
for r in range(reps):
    batch.viterbi_launch(); batch.sync()
    ts.append(sum(batch.viterbi_timing()))
print("resident kernel", n, "pairs:", " ".join("%.2f" % t for t in ts), "ms", flush=True)
batch.close()
for form in ("chunks", "stream"):
    os.environ["COATI_HIP_PIPE"] = form
    for pinned in (True, False):
        out, ts = None, []
        for r in range(reps):
            if want_timing and r == reps - 1: os.environ["COATI_HIP_PIPE_TIMING"] = "1"
            t0 = time.perf_counter()
            if pinned:
                out = model.viterbi(pa, a[1], pb, a[3], out=out, pinned=True)
            else:
                out = model.viterbi(a[0], a[1], a[2], a[3], out=out)
            ts.append((time.perf_counter() - t0) * 1e3)
            os.environ.pop("COATI_HIP_PIPE_TIMING", None)
        print(form, "pinned" if pinned else "pageable", n, "pairs:", " ".join("%.2f" % t for t in ts), "ms", flush=True)
