#!/usr/bin/env python3
"""Per-launch kernel times of the headline batch (HIP events): the distribution over consecutive launches of ONE resident batch.
usage: launch_times.py [pairs = 10000] [launches = 60]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 60
m = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
b = hip.Batch(m, *host.synth_encoded(0, n))
for _ in range(3):
    b.viterbi_launch()
b.sync()
ts = []
for i in range(k):
    b.viterbi_launch()
    b.sync()
    ts.append(b.viterbi_timing()[0])
ts = np.array(ts)
print("synced launches:", " ".join("%.2f" % t for t in ts))
print("min %.3f p25 %.3f median %.3f p75 %.3f max %.3f" % tuple(np.percentile(ts, [0, 25, 50, 75, 100])))
for i in range(k):
    b.viterbi_launch()
b.sync()
ts2 = np.array([b.viterbi_timing(i)[0] for i in range(min(k, 60))])
print("back-to-back launches (newest first):", " ".join("%.2f" % t for t in ts2))
print("min %.3f p25 %.3f median %.3f p75 %.3f max %.3f" % tuple(np.percentile(ts2, [0, 25, 50, 75, 100])))
