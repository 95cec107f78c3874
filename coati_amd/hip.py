"""ctypes binding of libcoati_hip.so (include/coati_hip.h).

Plumbing only: the product is the shared library.  There is no CPU fallback --
if the library is missing or no gfx950 device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parent
LIB_PATH = _ROOT / "_build" / "libcoati_hip.so"

TABLE_ROWS, TABLE_COLS = 183, 15
OP_MATCH, OP_DEL, OP_INS = 0, 1, 2
OPT_PERSISTENT_CALL, OPT_CK_BAND, OPT_FORWARD_MODE = 1, 2, 3  # coati_hip_model_set_option
FORWARD_EXACT, FORWARD_TOLERANCE = 0, 1  # values of OPT_FORWARD_MODE

# every symbol include/coati_hip.h declares
EXPORTS = (
    "coati_hip_version",
    "coati_hip_device_count",
    "coati_hip_last_error",
    "coati_hip_model_create",
    "coati_hip_model_create_tables",
    "coati_hip_model_destroy",
    "coati_hip_model_trim",
    "coati_hip_model_set_option",
    "coati_hip_model_prepare",
    "coati_hip_batch_create",
    "coati_hip_batch_create_tables",
    "coati_hip_batch_destroy",
    "coati_hip_batch_pairs",
    "coati_hip_batch_device_bytes",
    "coati_hip_batch_cells",
    "coati_hip_viterbi_launch",
    "coati_hip_batch_sync",
    "coati_hip_viterbi_wait",
    "coati_hip_viterbi_fetch",
    "coati_hip_viterbi_last_timing",
    "coati_hip_viterbi_band_stats",
    "coati_hip_viterbi_timing",
    "coati_hip_batch_result_ptrs",
    "coati_hip_forward_launch",
    "coati_hip_forward_final",
    "coati_hip_debug_forward_matrices",
    "coati_hip_sampleback",
    "coati_hip_sampleback_prepare",
    "coati_hip_debug_rng_f24",
    "coati_hip_debug_libm",
    "coati_hip_viterbi_batch",
    "coati_hip_shard_bounds",
    "coati_hip_host_alloc",
    "coati_hip_host_free",
    "coati_hip_debug_viterbi_flags",
    "coati_hip_debug_reload_env",
)


class CoatiHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"coati_hip error {code}: {message}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libcoati_hip.so; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("COATI_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise ImportError(
            f"{path} not found: build it with `make lib` (or __graft_entry__.build()); "
            "coati_amd has no CPU fallback"
        )
    lib = C.CDLL(str(path))
    vp, u64, i32, f32 = C.c_void_p, C.c_uint64, C.c_int, C.c_float
    lib.coati_hip_version.restype = C.c_uint32
    lib.coati_hip_device_count.restype = i32
    lib.coati_hip_last_error.restype = C.c_char_p
    lib.coati_hip_model_create.argtypes = [vp, f32, f32, f32, f32, i32, i32, C.POINTER(vp)]
    if hasattr(lib, "coati_hip_model_create_tables"):  # (older builds in the A/B harness lack the multi-table entries)
        lib.coati_hip_model_create_tables.argtypes = [vp, C.c_uint32, f32, f32, f32, f32, i32, i32, C.POINTER(vp)]
        lib.coati_hip_batch_create_tables.argtypes = [vp, u64, vp, vp, vp, vp, vp, C.POINTER(vp)]
    lib.coati_hip_model_destroy.argtypes = [vp]
    lib.coati_hip_model_destroy.restype = None
    if hasattr(lib, "coati_hip_model_trim"):
        lib.coati_hip_model_trim.argtypes = [vp]
    if hasattr(lib, "coati_hip_model_set_option"):
        lib.coati_hip_model_set_option.argtypes = [vp, C.c_int, C.c_int64]
    if hasattr(lib, "coati_hip_model_prepare"):
        lib.coati_hip_model_prepare.argtypes = [vp, u64, u64, u64]
    lib.coati_hip_batch_create.argtypes = [vp, u64, vp, vp, vp, vp, C.POINTER(vp)]
    lib.coati_hip_batch_destroy.argtypes = [vp]
    lib.coati_hip_batch_destroy.restype = None
    lib.coati_hip_batch_device_bytes.argtypes = [vp]
    lib.coati_hip_batch_device_bytes.restype = u64
    lib.coati_hip_batch_cells.argtypes = [vp]
    lib.coati_hip_batch_cells.restype = u64
    lib.coati_hip_viterbi_launch.argtypes = [vp]
    lib.coati_hip_batch_sync.argtypes = [vp]
    if hasattr(lib, "coati_hip_viterbi_wait"):
        lib.coati_hip_viterbi_wait.argtypes = [vp]
    lib.coati_hip_viterbi_fetch.argtypes = [vp, vp, vp, u64, vp, vp]
    lib.coati_hip_viterbi_last_timing.argtypes = [vp, C.POINTER(f32), C.POINTER(f32)]
    lib.coati_hip_viterbi_timing.argtypes = [vp, C.c_uint32, C.POINTER(f32), C.POINTER(f32)]
    if hasattr(lib, "coati_hip_viterbi_band_stats"):
        lib.coati_hip_viterbi_band_stats.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(u64)]
    lib.coati_hip_batch_result_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(vp),
                                                C.POINTER(vp)]
    lib.coati_hip_forward_launch.argtypes = [vp]
    lib.coati_hip_forward_final.argtypes = [vp, vp]
    lib.coati_hip_debug_forward_matrices.argtypes = [vp, u64, vp, vp, vp, u64]
    lib.coati_hip_sampleback.argtypes = [vp, C.c_uint32, vp, i32, vp, vp, u64, vp, vp, vp]
    lib.coati_hip_sampleback_prepare.argtypes = [vp, C.c_uint32, i32]
    lib.coati_hip_debug_rng_f24.argtypes = [vp, vp, C.c_uint32, vp]
    if hasattr(lib, "coati_hip_debug_libm"):
        lib.coati_hip_debug_libm.argtypes = [vp, i32, vp, u64, vp]
    lib.coati_hip_viterbi_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp]
    lib.coati_hip_debug_viterbi_flags.argtypes = [vp, u64, vp, u64]
    if hasattr(lib, "coati_hip_shard_bounds"):
        lib.coati_hip_shard_bounds.argtypes = [u64, vp, vp, i32, vp]
    if hasattr(lib, "coati_hip_host_alloc"):
        lib.coati_hip_host_alloc.argtypes = [u64, C.POINTER(vp)]
        lib.coati_hip_host_free.argtypes = [vp]
        lib.coati_hip_host_free.restype = None
    _lib = lib
    return lib


def reload_env() -> None:
    """The library reads its COATI_HIP_* switches once per process; this plumbing (tests, tools that flip a switch between two
    batches of one process) has it read them again before every entry that consults them."""
    lib = load()
    if hasattr(lib, "coati_hip_debug_reload_env"):
        lib.coati_hip_debug_reload_env()


def _check(rc: int) -> None:
    if rc != 0:
        raise CoatiHipError(rc, load().coati_hip_last_error().decode(errors="replace"))


def _ptr(arr):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


def shard_bounds(a_off, b_off, world: int) -> np.ndarray:
    """coati_hip_shard_bounds: world+1 pair indices of contiguous shards of equal DP-cell count."""
    a_off = np.ascontiguousarray(a_off, np.uint64)
    b_off = np.ascontiguousarray(b_off, np.uint64)
    out = np.zeros(world + 1, np.uint64)
    _check(load().coati_hip_shard_bounds(len(a_off) - 1, _ptr(a_off), _ptr(b_off), world, _ptr(out)))
    return out


class _PinnedBlock:
    """Owner of one coati_hip_host_alloc block; freed when the last array viewing it goes."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        _check(load().coati_hip_host_alloc(max(int(nbytes), 1), C.byref(self.ptr)))
        self.nbytes = max(int(nbytes), 1)

    def __del__(self):
        try:
            if self.ptr:
                load().coati_hip_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """numpy array in page-locked host memory (coati_hip_host_alloc): coati_hip_viterbi_batch copies
    from / into such arrays by DMA, overlapped with its kernels."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    block = _PinnedBlock(n * dtype.itemsize)
    buf = (C.c_char * block.nbytes).from_address(block.ptr.value)
    buf._coati_owner = block  # the array keeps `buf` alive (its .base), `buf` keeps the block: freed with the last view
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def pinned_copy(arr) -> np.ndarray:
    out = pinned_empty(np.shape(arr), np.asarray(arr).dtype)
    out[...] = arr
    return out


def device_count() -> int:
    return int(load().coati_hip_device_count())


def pack_pairs(pairs):
    """[(a_codes, b_codes), ...] -> (a_cat, a_off, b_cat, b_off) as the ABI wants them."""
    a_off = np.zeros(len(pairs) + 1, np.uint64)
    b_off = np.zeros(len(pairs) + 1, np.uint64)
    for p, (a, b) in enumerate(pairs):
        a_off[p + 1] = a_off[p] + len(a)
        b_off[p + 1] = b_off[p] + len(b)
    a_cat = np.zeros(max(int(a_off[-1]), 1), np.uint8)
    b_cat = np.zeros(max(int(b_off[-1]), 1), np.uint8)
    for p, (a, b) in enumerate(pairs):
        a_cat[int(a_off[p]):int(a_off[p + 1])] = a
        b_cat[int(b_off[p]):int(b_off[p + 1])] = b
    return a_cat, a_off, b_cat, b_off


# Test plumbing: the Forward mode every Model of this process is put into right after its creation (None: the library's
# default) -- through coati_hip_model_set_option, so that one pytest process runs the sampling suite in both modes.
DEFAULT_FORWARD_MODE = None


class Model:
    """coati_hip_model_t: the 183x15 table (or a stack of n of them, shape (n, 183, 15), for batches
    whose pairs use different tables) + host-computed log gap constants."""

    def __init__(self, table, consts, gap_len: int = 1, device: int = 0, forward_mode=None):
        table = np.ascontiguousarray(table, np.float32)
        if table.shape == (TABLE_ROWS, TABLE_COLS):
            table = table[None]
        if table.ndim != 3 or table.shape[1:] != (TABLE_ROWS, TABLE_COLS):
            raise ValueError("table must be 183x15 or n x 183 x 15")
        reload_env()
        self._h = C.c_void_p()
        self.gap_len = gap_len
        self.device = device
        self.n_tables = int(table.shape[0])
        c = [float(x) for x in consts]
        lib = load()
        if self.n_tables == 1 and not hasattr(lib, "coati_hip_model_create_tables"):
            _check(lib.coati_hip_model_create(_ptr(table), c[0], c[1], c[2], c[3], gap_len, device, C.byref(self._h)))
        else:
            _check(lib.coati_hip_model_create_tables(_ptr(table), self.n_tables, c[0], c[1], c[2], c[3], gap_len, device,
                                                     C.byref(self._h)))
        mode = forward_mode if forward_mode is not None else DEFAULT_FORWARD_MODE
        if mode is not None:
            self.set_option(OPT_FORWARD_MODE, mode)

    def trim(self):
        """Free the workspaces the model cached from destroyed batches."""
        _check(load().coati_hip_model_trim(self._h))

    def set_option(self, option: int, value: int):
        """coati_hip_model_set_option (OPT_PERSISTENT_CALL = 1: 0 forbids the device-owning one-shot form)."""
        _check(load().coati_hip_model_set_option(self._h, int(option), int(value)))

    def prepare(self, n_pairs: int, len_a: int, len_b: int):
        """coati_hip_model_prepare: allocate now what a one-shot call of about this size would allocate first."""
        _check(load().coati_hip_model_prepare(self._h, int(n_pairs), int(len_a), int(len_b)))

    def close(self):
        if self._h:
            load().coati_hip_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def debug_libm(self, op: int, x):
        """Device restatement of expf (op 0), log1pf (1) or logf (2) applied to a float32 array."""
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        _check(load().coati_hip_debug_libm(self._h, op, _ptr(x), C.c_uint64(x.size), _ptr(out)))
        return out

    def debug_rng_f24(self, state, n: int):
        st = np.ascontiguousarray(state, np.uint64)
        out = np.zeros(n, np.float32)
        _check(load().coati_hip_debug_rng_f24(self._h, _ptr(st), n, _ptr(out)))
        return out

    def viterbi(self, a_cat, a_off, b_cat, b_off, out=None, pinned=False):
        """One-shot coati_hip_viterbi_batch.  Returns (scores, ops, ops_off, ops_len); `out` may pass
        the four arrays of an earlier call back in (a loop over batches then writes into memory whose
        pages exist already -- first-touch page faults are half the cost of downloading the ops)."""
        reload_env()
        n = len(a_off) - 1
        total = int(a_off[-1] - a_off[0] + b_off[-1] - b_off[0])
        if out is not None and len(out[0]) == n and len(out[1]) >= max(total, 1):
            scores, ops, ops_off, ops_len = out
        elif pinned:
            scores, ops = pinned_empty(n, np.float32), pinned_empty(max(total, 1), np.uint8)
            ops_off, ops_len = pinned_empty(n, np.uint64), pinned_empty(n, np.uint32)
        else:
            scores = np.zeros(n, np.float32)
            ops = np.zeros(max(total, 1), np.uint8)
            ops_off = np.zeros(n, np.uint64)
            ops_len = np.zeros(n, np.uint32)
        _check(load().coati_hip_viterbi_batch(self._h, n, _ptr(a_cat), _ptr(a_off), _ptr(b_cat), _ptr(b_off),
                                              _ptr(scores), _ptr(ops), total, _ptr(ops_off), _ptr(ops_len)))
        return scores, ops, ops_off, ops_len


class Batch:
    """coati_hip_batch_t: encoded pairs + workspace resident in HBM."""

    def __init__(self, model: Model, a_cat, a_off, b_cat, b_off, table_index=None):
        reload_env()
        self.model = model
        self.n = len(a_off) - 1
        self.ops_total = int(a_off[-1] - a_off[0] + b_off[-1] - b_off[0])
        self._h = C.c_void_p()
        a_cat = np.ascontiguousarray(a_cat, np.uint8)
        b_cat = np.ascontiguousarray(b_cat, np.uint8)
        a_off = np.ascontiguousarray(a_off, np.uint64)
        b_off = np.ascontiguousarray(b_off, np.uint64)
        self.lens = np.stack([np.diff(a_off), np.diff(b_off)], axis=1).astype(np.int64)
        ti = None if table_index is None else np.ascontiguousarray(table_index, np.uint32)
        if ti is not None and len(ti) != self.n:
            raise ValueError("table_index needs one entry per pair")
        lib = load()
        if ti is None and not hasattr(lib, "coati_hip_batch_create_tables"):
            _check(lib.coati_hip_batch_create(model._h, self.n, _ptr(a_cat), _ptr(a_off), _ptr(b_cat), _ptr(b_off),
                                              C.byref(self._h)))
        else:
            _check(lib.coati_hip_batch_create_tables(model._h, self.n, _ptr(a_cat), _ptr(a_off), _ptr(b_cat),
                                                     _ptr(b_off), _ptr(ti) if ti is not None else None, C.byref(self._h)))

    def close(self):
        if self._h:
            load().coati_hip_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def cells(self) -> int:
        return int(load().coati_hip_batch_cells(self._h))

    @property
    def device_bytes(self) -> int:
        return int(load().coati_hip_batch_device_bytes(self._h))

    def viterbi_launch(self):
        reload_env()
        _check(load().coati_hip_viterbi_launch(self._h))

    def sync(self):
        _check(load().coati_hip_batch_sync(self._h))

    def wait(self):
        """Wait for this batch's last Viterbi launch only (later launches of other batches keep running)."""
        _check(load().coati_hip_viterbi_wait(self._h))

    def viterbi_fetch(self):
        scores = np.zeros(self.n, np.float32)
        ops = np.zeros(max(self.ops_total, 1), np.uint8)
        ops_off = np.zeros(self.n, np.uint64)
        ops_len = np.zeros(self.n, np.uint32)
        _check(load().coati_hip_viterbi_fetch(self._h, _ptr(scores), _ptr(ops), self.ops_total, _ptr(ops_off),
                                              _ptr(ops_len)))
        return scores, ops, ops_off, ops_len

    def viterbi_timing(self, launches_back: int = 0):
        """(fill_ms, walk_ms) of the launch issued `launches_back` launches before the last one."""
        f, w = C.c_float(), C.c_float()
        _check(load().coati_hip_viterbi_timing(self._h, launches_back, C.byref(f), C.byref(w)))
        return f.value, w.value

    def band_stats(self):
        """(band half width in steps the last launch ran with -- 0: everything kept or not the banded kernel --, pairs filled twice)."""
        band, twice = C.c_uint32(0), C.c_uint64(0)
        _check(load().coati_hip_viterbi_band_stats(self._h, C.byref(band), C.byref(twice)))
        return int(band.value), int(twice.value)

    def forward_launch(self):
        reload_env()
        _check(load().coati_hip_forward_launch(self._h))

    def forward_final(self):
        """Terminal-adjusted (M, D, I) of the last cell of every pair: array (n, 3)."""
        out = np.zeros((self.n, 3), np.float32)
        _check(load().coati_hip_forward_final(self._h, _ptr(out)))
        return out

    def debug_forward_matrices(self, pair: int):
        la, lb = (int(x) for x in self.lens[pair])
        M = np.zeros((la, lb), np.float32)
        D = np.zeros_like(M)
        I = np.zeros_like(M)
        if M.size:
            _check(load().coati_hip_debug_forward_matrices(self._h, pair, _ptr(M), _ptr(D), _ptr(I), M.size))
        return M, D, I

    def sampleback_prepare(self, n_samples: int, independent: bool = False):
        """coati_hip_sampleback_prepare: the coming sampleback call's allocations now (behind forward_launch: under the kernel)."""
        _check(load().coati_hip_sampleback_prepare(self._h, int(n_samples), int(independent)))

    def sampleback(self, n_samples: int, rng_states, independent: bool = False, out=None):
        """rng_states: (n, 2) uint64 (lo, hi).  Returns (log_weights (n, S), ops, ops_off (n, S), ops_len (n, S),
        rng_states_out).  out: the tuple a previous call of the same shape returned, to be written again (an embedder
        keeps its result arrays: fresh ones are first touched page by page under the download)."""
        reload_env()
        st = np.ascontiguousarray(rng_states, np.uint64).reshape(self.n, 2)
        total = int(n_samples * self.lens.sum())
        if out is not None and out[0].shape == (self.n, n_samples) and out[1].size == max(total, 1):
            lw, ops, off, ln, st_out = out
        else:
            lw = np.zeros((self.n, n_samples), np.float32)
            ops = np.zeros(max(total, 1), np.uint8)
            off = np.zeros((self.n, n_samples), np.uint64)
            ln = np.zeros((self.n, n_samples), np.uint32)
            st_out = np.zeros_like(st)
        _check(load().coati_hip_sampleback(self._h, n_samples, _ptr(st), int(independent), _ptr(lw), _ptr(ops), total,
                                           _ptr(off), _ptr(ln), _ptr(st_out)))
        return lw, ops, off, ln, st_out

    def result_ptrs(self):
        """Device addresses (scores, ops, ops_bytes, ops_off, ops_len) of the result arrays."""
        sc, ops, off, ln = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nbytes = C.c_uint64()
        _check(load().coati_hip_batch_result_ptrs(self._h, C.byref(sc), C.byref(ops), C.byref(nbytes), C.byref(off),
                                                  C.byref(ln)))
        return sc.value, ops.value, int(nbytes.value), off.value, ln.value

    def debug_flags(self, pair: int):
        reload_env()
        la, lb = self.lens[pair]
        out = np.zeros((int(la), int(lb)), np.uint8)
        if out.size:
            _check(load().coati_hip_debug_viterbi_flags(self._h, pair, _ptr(out), out.size))
        return out
