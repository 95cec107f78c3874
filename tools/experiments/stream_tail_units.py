"""A/B of the SIZE of the streamed call's last chunk (COATI_HIP_STREAM_TAIL_UNITS, tenths of 10^9 cells; default 26 = 2 600 pairs of
1 kb) and its number of row parts (COATI_HIP_STREAM_PARTS 22 .. 28), page-locked arrays; every mode's results against the resident batch.
usage: python tools/experiments/stream_tail_units.py [n_pairs]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from coati_amd import hip, host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
table, consts = host.set_subst("mar-mg"), host.gap_consts()
a = host.synth_encoded(0, n)
m0 = hip.Model(table, consts, 1)
batch = hip.Batch(m0, *a)
ts = []
for r in range(6):
    batch.viterbi_launch(); batch.sync()
    ts.append(sum(batch.viterbi_timing()))
want = batch.viterbi_fetch()
batch.close()
m0.close()
resident = float(np.median(ts[2:]))
print(f"{n} pairs: resident kernel {resident:.2f} ms", flush=True)
os.environ["COATI_HIP_PIPE"] = "stream"
pa, pb = hip.pinned_copy(a[0]), hip.pinned_copy(a[2])
for rep in range(2):
    for units, parts in ((26, 23), (26, 24), (36, 23), (36, 24), (46, 24), (46, 23), (26, 23)):
        os.environ["COATI_HIP_STREAM_TAIL_UNITS"] = str(units)
        os.environ["COATI_HIP_STREAM_PARTS"] = str(parts)
        model = hip.Model(table, consts, 1)  # (the tail workspace is sized per model)
        out, ts = None, []
        for r in range(9):
            t0 = time.perf_counter()
            out = model.viterbi(pa, a[1], pb, a[3], out=out, pinned=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        same = bool((out[0].view(np.uint32) == want[0].view(np.uint32)).all() and (out[3] == want[3]).all())
        med = float(np.median(ts[3:]))
        print(f"  tail {units / 10:.1f} units x {parts - 20} parts: median {med:.2f} ms  best {min(ts):.2f}  resident/median {resident / med:.3f}  results equal: {same}", flush=True)
        model.close()
