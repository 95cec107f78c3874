"""GPU check of the streamed one-shot call against one resident batch (bit-exact), both pipeline forms.
usage: python tools/stream_check.py [n_pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coati_amd import hip, host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
table, consts = host.set_subst("mar-mg"), host.gap_consts()
a = host.synth_encoded(0, n)
model = hip.Model(table, consts, 1)
batch = hip.Batch(model, *a)
batch.viterbi_launch()
want = batch.viterbi_fetch()
batch.close()
valid = np.zeros(len(want[1]), bool)
for p in range(n):
    valid[int(want[2][p]):int(want[2][p]) + int(want[3][p])] = True
pa, pb = hip.pinned_copy(a[0]), hip.pinned_copy(a[2])
for form in ("chunks", "stream"):
    os.environ["COATI_HIP_PIPE"] = form
    for pinned in (True, False):
        for r in range(3):
            t0 = time.perf_counter()
            got = model.viterbi(pa, a[1], pb, a[3], pinned=pinned)
            dt = time.perf_counter() - t0
            ok = ((got[0].view(np.uint32) == want[0].view(np.uint32)).all() and (got[3] == want[3]).all() and (got[2] == want[2]).all()
                  and (got[1][:len(want[1])][valid] == want[1][valid]).all())
            print(form, "pinned" if pinned else "pageable", "call", r, "%.2f ms" % (dt * 1e3), "OK" if ok else "MISMATCH", flush=True)
            if not ok:
                sys.exit(1)
