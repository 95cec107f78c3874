"""The device restatements of glibc's expf / log1pf / logf (coati_amd/csrc/glibc_math.hpp) against
the host libm, bit for bit, on dense samples of the ranges the log-semiring path produces.  (The
same source, compiled for the host, is compared with libm on EVERY float of those ranges by
tools/libm_check.cc: 3.26e9 inputs, 0 mismatches.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def float_range(lo, hi, n, rng):
    """n float32 values: a uniform sample of the bit patterns between lo and hi plus both ends."""
    a, b = sorted((int(np.float32(lo).view(np.uint32)), int(np.float32(hi).view(np.uint32))))
    u = rng.integers(a, b + 1, n, dtype=np.uint64).astype(np.uint32)
    u[:2] = (a, b)
    return u.view(np.float32)


@pytest.mark.parametrize("op,lo,hi,extra", [
    (0, -0.0, -104.0, [-0.0, -1e-30, -16.0, -16.000002, -87.5, -103.97, -104.5, -1e10, -3.4e38, -np.inf]),
    (1, 0.0, 1.0, [0.0, 1e-38, 2.0 ** -54, 2.0 ** -29, 1.1e-7, 0.41421, 0.41422, 0.5, 1.0]),
    (2, 2.0 ** -126, 4.0, [1.0, 0.9999999, 1.0000001, 2.0, 3.0, 0.70710677, 1.4142135]),
    # op 3: the straight-line log1pf of the Forward fill, [2^-29, 1] (expected values: log1pf)
    (3, 2.0 ** -29, 1.0, [2.0 ** -29, 1.1e-7, 0.41421354, 0.41421357, 0.4142136, 0.414214, 0.5, 0.99999964, 0.9999997,
                          0.99999976, 0.9999998, 0.9999999, 0.99999994, 1.0]),
])
def test_device_libm_bit_exact(oracle, op, lo, hi, extra):
    from coati_amd import hip, host

    rng = np.random.default_rng(op)
    x = np.concatenate([float_range(lo, hi, 4_000_000, rng), np.array(extra, np.float32)])
    model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
    got = model.debug_libm(op, x)
    want = oracle.libm(1 if op == 3 else op, x)
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), (op, x[bad][:5], got[bad][:5], want[bad][:5])
    model.close()
