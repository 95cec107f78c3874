"""A/B of the streamed call's end game: COATI_HIP_STREAM_PARTS = 0 (no row parts), 1 (row parts in the last ~1 000-pair
chunks), -1 = default = 23, 22 .. 24 (ONE last chunk of <= 2 600 pairs cut into 2 .. 4 row parts, part-major).  Page-locked arrays, median of
the calls after the first two; every mode's scores / lengths against the resident batch.
usage: python tools/stream_tail_ab.py [n_pairs ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coati_amd import hip, host

sizes = [int(x) for x in sys.argv[1:]] or [10000]
table, consts = host.set_subst("mar-mg"), host.gap_consts()
for n in sizes:
    model = hip.Model(table, consts, 1)
    a = host.synth_encoded(0, n)
    pa, pb = hip.pinned_copy(a[0]), hip.pinned_copy(a[2])
    batch = hip.Batch(model, *a)
    ts = []
    for r in range(6):
        batch.viterbi_launch(); batch.sync()
        ts.append(sum(batch.viterbi_timing()))
    want = batch.viterbi_fetch()
    batch.close()
    resident = float(np.median(ts[2:]))
    print(f"{n} pairs: resident kernel {resident:.2f} ms", flush=True)
    os.environ["COATI_HIP_PIPE"] = "stream"
    for mode in (0, -1, 22, 24, 1, 0, -1):
        os.environ["COATI_HIP_STREAM_PARTS"] = str(mode)
        out, ts = None, []
        for r in range(9):
            t0 = time.perf_counter()
            out = model.viterbi(pa, a[1], pb, a[3], out=out, pinned=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        same = bool((out[0].view(np.uint32) == want[0].view(np.uint32)).all() and (out[3] == want[3]).all())
        med = float(np.median(ts[2:]))
        print(f"  STREAM_PARTS={mode:2d}: median {med:.2f} ms  best {min(ts):.2f}  first {ts[0]:.1f}  resident/median {resident / med:.3f}  results equal: {same}", flush=True)
    model.close()
