#!/usr/bin/env python3
"""Exhaustive DEVICE check of the libm restatements (coati_amd/csrc/glibc_math.hpp): every float of
the ranges the log-semiring path can produce goes through the GPU code (coati_hip_debug_libm) and is
compared bit for bit with the host libm (oracle_libm).  tools/libm_check.cc does the same for the
host build of the header; this one covers what only exists on the device (v_rcp_f32 in the
straight-line log1pf, the compiler's code for the double-precision parts).

    python tools/libm_device_check.py [--stride N]      (N = 1: ~3.5e9 inputs, a few minutes)
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from coati_amd import hip, host  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (the checker)

RANGES = [  # (name, op, low bits, high bits)
    ("expf    [-104, -0]", 0, np.float32(-0.0), np.float32(-104.0)),
    ("log1pf  [0, 1]", 1, np.float32(0.0), np.float32(1.0)),
    ("logf    [2^-126, 4]", 2, np.float32(2.0 ** -126), np.float32(4.0)),
    ("log1pf/straight-line [2^-29, 1]", 3, np.float32(2.0 ** -29), np.float32(1.0)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=1 << 26)
    args = ap.parse_args()
    model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
    bad_total = 0
    for name, op, lo, hi in RANGES:
        a, b = sorted((int(lo.view(np.uint32)), int(hi.view(np.uint32))))
        n = bad = 0
        t0 = time.time()
        for start in range(a, b + 1, args.chunk * args.stride):
            u = np.arange(start, min(b + 1, start + args.chunk * args.stride), args.stride, dtype=np.uint64).astype(np.uint32)
            x = u.view(np.float32)
            got = model.debug_libm(op, x)
            want = orc.libm(1 if op == 3 else op, x)
            m = got.view(np.uint32) != want.view(np.uint32)
            if m.any() and bad < 5:
                i = np.flatnonzero(m)[0]
                print(f"  {name}: x = {float(x[i]).hex()}: device {float(got[i]).hex()}, libm {float(want[i]).hex()}")
            bad += int(m.sum())
            n += x.size
        print(f"{name}: {n} inputs, {bad} mismatches ({time.time() - t0:.1f} s)", flush=True)
        bad_total += bad
    model.close()
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
