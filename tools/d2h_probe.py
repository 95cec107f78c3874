#!/usr/bin/env python3
"""Stage measurements for the multi-GPU budget: what ONE rank's PCIe link and host do with the
results of BASELINE configs[4] (1 000 000 pairs x ~2 kB of ops = 2 GB): device -> host copies into page-locked and
pageable arrays, first touch and warm, and the host-side memcpy of the same bytes (the root's unpack).
usage: d2h_probe.py [MB ...]   (default 256 2048)"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from coati_amd import hip  # noqa: E402

rt = C.CDLL("libamdhip64.so")
rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
rt.hipFree.argtypes = [C.c_void_p]
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
rt.hipDeviceSynchronize.argtypes = []
D2H, H2D, D2D = 2, 1, 3


def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def main():
    hip.load()
    assert hip.device_count() > 0
    sizes = [int(x) for x in sys.argv[1:]] or [256, 2048]
    for mb in sizes:
        n = mb << 20
        d = C.c_void_p()
        assert rt.hipMalloc(C.byref(d), n) == 0
        rt.hipMemset(d, 1, n)
        rt.hipDeviceSynchronize()
        d2 = C.c_void_p()
        assert rt.hipMalloc(C.byref(d2), n) == 0
        pinned = hip.pinned_empty(n, np.uint8)
        pageable = np.empty(n, np.uint8)  # untouched pages
        t_first = timed(lambda: rt.hipMemcpy(pageable.ctypes.data_as(C.c_void_p), d, n, D2H), 1)[0]
        t_page = min(timed(lambda: rt.hipMemcpy(pageable.ctypes.data_as(C.c_void_p), d, n, D2H)))
        t_pin = min(timed(lambda: rt.hipMemcpy(pinned.ctypes.data_as(C.c_void_p), d, n, D2H)))
        t_h2d = min(timed(lambda: rt.hipMemcpy(d, pinned.ctypes.data_as(C.c_void_p), n, H2D)))

        def d2d():
            rt.hipMemcpy(d2, d, n, D2D)
            rt.hipDeviceSynchronize()

        t_d2d = min(timed(d2d))
        dst = np.empty(n, np.uint8)
        dst[:] = 0
        t_cpy = min(timed(lambda: np.copyto(dst, pageable)))
        gb = n / 1e9
        print(f"{mb:5d} MB: D2H pinned {gb / t_pin:6.1f} GB/s ({t_pin * 1e3:.1f} ms)  pageable warm {gb / t_page:6.1f} GB/s ({t_page * 1e3:.1f} ms)  "
              f"pageable first touch {gb / t_first:6.1f} GB/s ({t_first * 1e3:.1f} ms)  H2D pinned {gb / t_h2d:6.1f} GB/s  "
              f"D2D {gb / t_d2d:6.1f} GB/s  host memcpy (1 thread) {gb / t_cpy:6.1f} GB/s ({t_cpy * 1e3:.1f} ms)", flush=True)
        rt.hipFree(d)
        rt.hipFree(d2)


if __name__ == "__main__":
    main()
