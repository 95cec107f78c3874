import sys, time, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
for n_pairs in (1, 4, 16, 64):
    enc = host.synth_encoded(0, n_pairs)
    batch = hip.Batch(model, *enc)
    batch.forward_launch(); batch.sync()
    st = np.stack([host.rng_seed(["42", str(p)]) for p in range(n_pairs)])
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); batch.sampleback(1000, st, independent=False); ts.append(time.perf_counter() - t0)
    print(f"{n_pairs} pairs x 1000 exact-stream samples: {np.median(ts[1:])*1e3:.1f} ms")
    batch.close()
