"""tools/trace_long.py for the golden REAL long pair: when the strips finished, how long the traceback took (trace build)"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
from tests import util
key = sys.argv[1] if len(sys.argv) > 1 else "160k"
a, b, case, doc = util.load_long_pair(key)
model = hip.Model(np.load(ROOT / "tests" / "golden" / doc["table"]), host.gap_consts(doc["gap_open"], doc["gap_extend"]), 1)
batch = hip.Batch(model, *hip.pack_pairs([(a, b)]))
tr = np.zeros(4096 * 4, np.uint64)
lib = hip.load()
for _ in range(3):
    lib.coati_hip_debug_trace_lp(tr.ctypes.data_as(C.c_void_p))
    batch.viterbi_launch(); batch.sync()
f, w = batch.viterbi_timing()
assert lib.coati_hip_debug_trace_lp(tr.ctypes.data_as(C.c_void_p)) == 0
raw = tr.reshape(4096, 4).copy()
wk = raw[4000].copy()  # (round 6) the walk's fast loop: iterations, cycles in its load-and-wait statement, cycles in all, windows asked ahead
raw[4000] = 0
t = raw[raw[:, 1] > 0].astype(np.float64)
t0 = t[:, 0].min()
walk = t[:, 2] > 0
print(f"{key}: kernel {f:.2f} ms; {len(t)} strips; first strip done at {(t[:, 1].min() - t0) / 100:.0f} us, last at {(t[:, 1].max() - t0) / 100:.0f} us; "
      f"traceback {((t[walk, 2] - t[walk, 1]).max()) / 100.0:.0f} us")
if wk[0]:
    print(f"walk (fast loop): {int(wk[0])} iterations, {wk[2] / wk[0]:.0f} shader cycles each ({wk[2] / 1e6:.2f} M in all); {int(wk[1]) & 0xffffffff} strips spliced ({int(wk[1]) >> 32} of them by their bridge); "
          f"{int(wk[3])} windows asked ahead")
