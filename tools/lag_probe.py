"""How closely does a strip of a multi-strip viterbi_ck pair follow its left neighbour?  64 pairs of 8 kb ancestors against
descendants of 1 .. 15 strips (prefixes of the ancestor), from the trace build: fill end and traceback time of the wavefront
that does a pair's LAST strip.  fill(s strips) = (rows + 63 + (s - 1) * lag) * step  ->  lag.
usage: COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so python tools/lag_probe.py [W ...]   (make trace first)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coati_amd import hip, host
from tests import util

table, consts = host.set_subst("mar-mg"), host.gap_consts()
rng = np.random.default_rng(1)
anc = [util.random_anc(rng, 2666) for _ in range(64)]
lib = hip.load()
for w in [int(x) for x in sys.argv[1:]] or [8, 4, 16]:
    os.environ["COATI_HIP_VITERBI_CK"] = "1"
    os.environ["COATI_HIP_STRIP_W"] = str(w)
    rows = []
    for strips in (1, 2, 4, 8, 15):
        if 64 * w * strips > 7998:
            continue
        lb = 64 * w * strips
        enc = util.encode_pairs([(a, a[:lb]) for a in anc])
        model = hip.Model(table, consts, 1)
        batch = hip.Batch(model, *hip.pack_pairs(enc))
        tr = np.zeros(4096 * 16, np.uint64)
        for _ in range(3):
            lib.coati_hip_debug_trace(tr.ctypes.data_as(C.c_void_p))
            batch.viterbi_launch(); batch.sync()
        total = sum(batch.viterbi_timing())
        if hasattr(lib, "coati_hip_debug_poll_stats"):
            ps = np.zeros(4, np.uint64)
            lib.coati_hip_debug_poll_stats(ps.ctypes.data_as(C.c_void_p))
            if ps[0]:
                print(f"   {strips} strips, 3 launches: {int(ps[0])} sub-blocks of consumer strips, {int(ps[1])} began without their boundary ({100.0 * ps[1] / ps[0]:.1f} %), {int(ps[2])} polls for those")
        assert lib.coati_hip_debug_trace(tr.ctypes.data_as(C.c_void_p)) == 0
        raw = tr.reshape(4096, 16).astype(np.float64)
        used = raw[:, 2] > 0  # (only the wavefront of a pair's last strip stamps a traceback)
        t0 = raw[raw[:, 0] > 0, 0].min()
        fill = (raw[used, 1] - t0) / 100.0
        walk = (raw[used, 2] - raw[used, 1]) / 100.0
        rows.append((strips, total, fill.mean(), walk.mean()))
        if strips == 15:  # every strip's fill end: the 64 pairs are alike, so the sorted times fall into one group per strip
            ends = np.sort((raw[raw[:, 1] > 0, 1] - t0) / 100.0)
            print("   items traced:", len(ends))
            groups = np.array([g.mean() for g in np.array_split(ends, strips)])
            print("   fill ends by strip (us):", " ".join(f"{g:.0f}" for g in groups))
            print("   strip-to-strip (us):    ", " ".join(f"{d:.0f}" for d in np.diff(groups)))
        batch.close(); model.close()
    step_us = rows[0][2] / (7998 + 63)
    print(f"W {w}: step {step_us:.3f} us (one strip: fill {rows[0][2]:.0f} us, traceback {rows[0][3]:.0f} us)")
    for s, total, f, wk in rows[1:]:
        print(f"   {s:2d} strips: kernel {total:.3f} ms  last strip's fill ends at {f:.0f} us  traceback {wk:.0f} us  -> lag {(f - rows[0][2]) / step_us / (s - 1):.0f} steps per strip")
