"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/_build/libcoati_oracle.so
(and, where it was built, oracle/_ref/libcoati_ref.so -- the real reference).

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg only.  Nothing under coati_amd/ imports it.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
ORACLE_LIB = _DIR / "_build" / "libcoati_oracle.so"
REF_LIB = _DIR / "_ref" / "libcoati_ref.so"

TROPICAL, LOG = 0, 1
OP_M, OP_D, OP_I = 0, 1, 2


class Rng(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]


def build() -> None:
    subprocess.run(["make", "-C", str(_DIR)], check=True, capture_output=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not ORACLE_LIB.exists():
            build()
        L = C.CDLL(str(ORACLE_LIB))
        L.oracle_traceback.restype = C.c_int64
        L.oracle_traceback_flags.restype = C.c_int64
        L.oracle_viterbi.restype = C.c_int64
        L.oracle_viterbi_lowmem.restype = C.c_int64
        L.oracle_sampleback.restype = C.c_int64
        L.oracle_sampleback_mdi.restype = C.c_int64
        L.oracle_path_logweight.restype = C.c_float
        L.oracle_path_score.restype = C.c_float
        L.oracle_rng_f24.restype = C.c_float
        L.oracle_rng_bits.restype = C.c_uint64
        L.oracle_viterbi_batch_timed.restype = C.c_double
        _lib = L
    return _lib


def _p(x):
    return None if x is None else x.ctypes.data_as(C.c_void_p)


def _u64(v):
    return C.c_uint64(int(v))


def gap_consts(gap_open=0.001, gap_extend=None) -> np.ndarray:
    """{no_gap, gap_stop, gap_open, gap_extend} in log space for linear g, e (fp32)."""
    g = np.float32(gap_open)
    e = np.float32(1.0) - np.float32(1.0) / np.float32(6.0) if gap_extend is None else np.float32(gap_extend)
    out = np.zeros(4, np.float32)
    lib().oracle_gap_consts(C.c_float(g), C.c_float(e), _p(out))
    return out


def fill(semiring, table, consts, L, a, b, edges=False):
    la, lb = len(a), len(b)
    rows, cols = la + L, lb + L
    M = np.zeros((rows, cols), np.float32)
    D = np.zeros_like(M)
    I = np.zeros_like(M)
    E = np.zeros((8, rows, cols), np.float32) if edges else None
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    rc = lib().oracle_fill(semiring, _p(table), _p(consts), L, _p(a), _u64(la), _p(b), _u64(lb), _p(M), _p(D),
                           _p(I), _p(E))
    assert rc == 0
    return (M, D, I, E) if edges else (M, D, I)


def traceback(M, D, I, consts, L):
    rows, cols = M.shape
    ops = np.zeros(rows + cols, np.uint8)
    score = C.c_float()
    n = lib().oracle_traceback(_p(M), _p(D), _p(I), _u64(rows), _u64(cols), _p(consts), L, _p(ops),
                               C.byref(score))
    return ops[:n].copy(), np.float32(score.value)


def tb_flags(M, D, I, consts):
    rows, cols = M.shape
    out = np.zeros((rows, cols), np.uint8)
    lib().oracle_tb_flags(_p(M), _p(D), _p(I), _u64(rows), _u64(cols), _p(consts), _p(out))
    return out


def viterbi(table, consts, L, a, b, lowmem=False):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    ops = np.zeros(len(a) + len(b) + 1, np.uint8)
    score = C.c_float()
    fn = lib().oracle_viterbi_lowmem if lowmem else lib().oracle_viterbi
    n = fn(_p(table), _p(consts), L, _p(a), _u64(len(a)), _p(b), _u64(len(b)), _p(ops), C.byref(score))
    return ops[:n].copy(), np.float32(score.value)


def ops_to_strings(ops, a_raw: str, b_raw: str):
    ops = np.ascontiguousarray(ops, np.uint8)
    oa = C.create_string_buffer(len(ops) + 1)
    ob = C.create_string_buffer(len(ops) + 1)
    lib().oracle_ops_to_strings(_p(ops), C.c_int64(len(ops)), a_raw.encode(), b_raw.encode(), oa, ob)
    return oa.value.decode(), ob.value.decode()


def _seed_array(seeds):
    return (C.c_char_p * len(seeds))(*[s.encode() for s in seeds])


def rng_seed(seeds) -> Rng:
    r = Rng()
    lib().oracle_rng_seed(C.byref(r), _seed_array(seeds), len(seeds))
    return r


def rng_f24(r: Rng) -> np.float32:
    return np.float32(lib().oracle_rng_f24(C.byref(r)))


def sampleback(mats, L, rng: Rng):
    """mats: (11, rows, cols) in reference order."""
    _, rows, cols = mats.shape
    ops = np.zeros(rows + cols, np.uint8)
    score = C.c_float()
    n = lib().oracle_sampleback(_p(mats), _u64(rows), _u64(cols), L, C.byref(rng), _p(ops), C.byref(score))
    return ops[:n].copy(), np.float32(score.value)


def sampleback_mdi(M, D, I, table, consts, L, a, b, rng: Rng):
    rows, cols = M.shape
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    ops = np.zeros(rows + cols, np.uint8)
    score = C.c_float()
    n = lib().oracle_sampleback_mdi(_p(M), _p(D), _p(I), _u64(rows), _u64(cols), _p(table), _p(consts), L,
                                    _p(a), _p(b), C.byref(rng), _p(ops), C.byref(score))
    return ops[:n].copy(), np.float32(score.value)


def path_logweight(M, D, I, table, consts, L, a, b, ops):
    rows, cols = M.shape
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    ops = np.ascontiguousarray(ops, np.uint8)
    return np.float32(lib().oracle_path_logweight(_p(M), _p(D), _p(I), _u64(rows), _u64(cols), _p(table),
                                                  _p(consts), L, _p(a), _p(b), _p(ops), C.c_int64(len(ops))))


def path_score(table, consts, L, a, b, ops):
    """Viterbi value along a given path, O(len(ops)); equals the DP score for the optimal path."""
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    ops = np.ascontiguousarray(ops, np.uint8)
    return np.float32(lib().oracle_path_score(_p(table), _p(consts), L, _p(a), _u64(len(a)), _p(b), _u64(len(b)), _p(ops),
                                               C.c_int64(len(ops))))


def libm(op: int, x):
    """The host libm (expf / log1pf / logf for op 0 / 1 / 2) applied to a float32 array."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    lib().oracle_libm(op, _p(x), _u64(x.size), _p(out))
    return out


def viterbi_batch_timed(table, consts, L, a_cat, a_off, b_cat, b_off, threads=1):
    n = len(a_off) - 1
    scores = np.zeros(n, np.float32)
    secs = lib().oracle_viterbi_batch_timed(_p(table), _p(consts), L, _u64(n), _p(a_cat), _p(a_off), _p(b_cat),
                                            _p(b_off), threads, _p(scores))
    return float(secs), scores


# ---- the compiled reference (build container only) --------------------------
_ref = None


def ref_available() -> bool:
    return REF_LIB.exists()


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        _ref = C.CDLL(str(REF_LIB))
    return _ref


def ref_viterbi(table, g, e, L, a_raw: str, b_raw: str, a, b, want_matrices=True):
    la, lb = len(a), len(b)
    rows, cols = la + L, lb + L
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    M = np.zeros((rows, cols), np.float32) if want_matrices else None
    D = np.zeros((rows, cols), np.float32) if want_matrices else None
    I = np.zeros((rows, cols), np.float32) if want_matrices else None
    sa = C.create_string_buffer(la + lb + 1)
    sb = C.create_string_buffer(la + lb + 1)
    sc = C.c_float()
    rc = ref().ref_viterbi(_p(table), C.c_float(g), C.c_float(e), L, a_raw.encode(), b_raw.encode(), _p(a), _p(b),
                           _u64(la), _u64(lb), _p(M), _p(D), _p(I), sa, sb, C.byref(sc))
    assert rc == 0
    return M, D, I, sa.value.decode(), sb.value.decode(), np.float32(sc.value)


def ref_forward_sample(table, g, e, L, a_raw, b_raw, a, b, seeds, n_samples, want_matrices=True):
    la, lb = len(a), len(b)
    rows, cols = la + L, lb + L
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    mats = np.zeros((11, rows, cols), np.float32) if want_matrices else None
    slot = la + lb + 1
    buf = C.create_string_buffer(2 * n_samples * slot)
    scores = np.zeros(n_samples, np.float32)
    rc = ref().ref_forward_sample(_p(table), C.c_float(g), C.c_float(e), L, a_raw.encode(), b_raw.encode(), _p(a),
                                  _p(b), _u64(la), _u64(lb), _seed_array(seeds), len(seeds), n_samples, _p(mats),
                                  buf, _p(scores))
    assert rc == 0
    alns = []
    for k in range(n_samples):
        sa = C.string_at(C.addressof(buf) + (2 * k) * slot).decode()
        sb = C.string_at(C.addressof(buf) + (2 * k + 1) * slot).decode()
        alns.append((sa, sb))
    return mats, alns, scores


def ref_rng_f24(seeds, n):
    out = np.zeros(n, np.float32)
    ref().ref_rng_f24(_seed_array(seeds), len(seeds), n, _p(out))
    return out
