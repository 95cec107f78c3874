"""ctypes binding of libcoati_hip_dist.so (include/coati_hip_dist.h): the native multi-GPU layer --
one process per GPU, RCCL linked directly.  Plumbing for tests and tools; the product is the library
(and `coati-alignpair --batch --devices ...`, which drives it from C++)."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import hip

LIB_PATH = Path(__file__).resolve().parent / "_build" / "libcoati_hip_dist.so"
ID_BYTES = 128

EXPORTS = (
    "coati_hip_dist_last_error",
    "coati_hip_dist_unique_id",
    "coati_hip_dist_init",
    "coati_hip_dist_destroy",
    "coati_hip_dist_rank",
    "coati_hip_dist_world",
    "coati_hip_dist_allreduce_f64",
    "coati_hip_dist_barrier",
    "coati_hip_dist_broadcast_model",
    "coati_hip_dist_gather",
    "coati_hip_dist_viterbi",
    "coati_hip_dist_viterbi_shard",
    "coati_hip_dist_viterbi_shard_local",
    "coati_hip_dist_chunk_plan",
    "coati_hip_dist_landing_plan",
    "coati_hip_dist_debug_job_times",
    "coati_hip_dist_simulate",
    "coati_hip_dist_simulate_local",
    "coati_hip_dist_job_host",
)

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    hip.load()  # libcoati_hip.so first (the dist library links it)
    path = Path(os.environ.get("COATI_HIP_DIST_LIB", LIB_PATH))
    if not path.exists():
        raise ImportError(f"{path} not found: build it with `make dist`")
    lib = C.CDLL(str(path))
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    lib.coati_hip_dist_last_error.restype = C.c_char_p
    lib.coati_hip_dist_unique_id.argtypes = [vp]
    lib.coati_hip_dist_init.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    lib.coati_hip_dist_destroy.argtypes = [vp]
    lib.coati_hip_dist_destroy.restype = None
    lib.coati_hip_dist_rank.argtypes = [vp]
    lib.coati_hip_dist_world.argtypes = [vp]
    lib.coati_hip_dist_broadcast_model.argtypes = [vp, i32, vp, C.c_uint32, vp, vp, vp]
    lib.coati_hip_dist_allreduce_f64.argtypes = [vp, i32, vp, C.c_uint32]
    lib.coati_hip_dist_barrier.argtypes = [vp]
    lib.coati_hip_dist_gather.argtypes = [vp, i32, vp, vp, vp, vp, u64, vp, vp]
    lib.coati_hip_dist_viterbi.argtypes = [vp, i32, vp, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp]
    lib.coati_hip_dist_viterbi_shard.argtypes = [vp, i32, vp, u64, vp, u64, vp, vp, u64, vp, vp, vp, u64, vp, vp]
    lib.coati_hip_dist_chunk_plan.argtypes = [u64, vp, vp, i32, u64, vp, vp, u64, vp]
    lib.coati_hip_dist_landing_plan.argtypes = [i32, i32, vp, vp, vp]
    lib.coati_hip_dist_simulate.argtypes = [i32, i32, u64, vp, vp, u64, vp, vp, vp, vp, vp, u64, vp, vp]
    lib.coati_hip_dist_viterbi_shard_local.argtypes = [vp, i32, vp, u64, vp, u64, vp, vp, u64, vp, vp, vp, u64, vp, vp, i32, vp, vp]
    lib.coati_hip_dist_simulate_local.argtypes = [i32, i32, u64, vp, vp, u64, i32, vp, vp, vp, vp, vp, u64, vp, vp, vp, vp]
    lib.coati_hip_dist_job_host.argtypes = [vp, i32, i32, i32, u64, vp, vp, u64, i32, i32, vp, vp, vp, vp, vp, u64, vp, vp, vp, vp]
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        raise hip.CoatiHipError(rc, load().coati_hip_dist_last_error().decode(errors="replace"))


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    _check(load().coati_hip_dist_unique_id(buf))
    return buf.raw


class Comm:
    def __init__(self, uid: bytes, world: int, rank: int, device: int = 0):
        self._h = C.c_void_p()
        self.world, self.rank = world, rank
        _check(load().coati_hip_dist_init(C.c_char_p(uid), world, rank, device, C.byref(self._h)))

    def close(self):
        if self._h:
            load().coati_hip_dist_destroy(self._h)
            self._h = C.c_void_p()

    def barrier(self):
        _check(load().coati_hip_dist_barrier(self._h))

    def allreduce(self, values, op: str = "sum"):
        """Element-wise all-reduce of up to 64 doubles ("sum" or "max"); returns the reduced array."""
        v = np.ascontiguousarray(values, np.float64).reshape(-1).copy()
        _check(load().coati_hip_dist_allreduce_f64(self._h, {"sum": 0, "max": 1}[op], hip._ptr(v), len(v)))
        return v

    def gather_device(self, batch, root: int = 0):
        """The gather without the download: the peers' result arrays land in the root's HBM (the landing zone of the
        communicator) and stay there; returns counts[world, 2].  What a multi-step driver times per step."""
        counts = np.zeros(2 * self.world, np.uint64)
        h = batch._h if batch is not None else None
        _check(load().coati_hip_dist_gather(self._h, root, h, hip._ptr(counts), None, None, 0, None, None))
        return counts.reshape(-1, 2)

    def viterbi_shard(self, model, a_cat, a_first, a_off, b_cat, b_first, b_off, root: int = 0, reuse=None, pinned: bool = True):
        """coati_hip_dist_viterbi_shard: a_off / b_off describe ALL pairs, a_cat / b_cat hold this rank's part of the
        concatenations starting at byte a_first / b_first.  Results on root (None elsewhere) -- in page-locked arrays by
        default: everything every rank computed comes down the root's one PCIe link, 57 GB/s into page-locked memory
        against 14-22 GB/s into pages that have to be faulted in first; `reuse` may pass an earlier call's result
        arrays back in."""
        a_cat, b_cat = np.ascontiguousarray(a_cat, np.uint8), np.ascontiguousarray(b_cat, np.uint8)
        a_off, b_off = np.ascontiguousarray(a_off, np.uint64), np.ascontiguousarray(b_off, np.uint64)
        n = len(a_off) - 1
        total = int(a_off[-1] - a_off[0] + b_off[-1] - b_off[0])
        if self.rank == root:
            if reuse is not None and len(reuse[0]) == n and len(reuse[1]) >= max(total, 1):
                scores, ops, off, ln = reuse
            else:
                make = hip.pinned_empty if pinned else (lambda shape, dt: np.zeros(shape, dt))
                scores, ops = make(max(n, 1), np.float32)[:n], make(max(total, 1), np.uint8)
                off, ln = make(max(n, 1), np.uint64)[:n], make(max(n, 1), np.uint32)[:n]
            args = (hip._ptr(scores), hip._ptr(ops), total, hip._ptr(off), hip._ptr(ln))
        else:
            scores = ops = off = ln = None
            args = (None, None, 0, None, None)
        _check(load().coati_hip_dist_viterbi_shard(self._h, root, model._h, n, hip._ptr(a_cat), int(a_first), hip._ptr(a_off), hip._ptr(b_cat),
                                                   int(b_first), hip._ptr(b_off), *args))
        return (scores, ops, off, ln) if self.rank == root else None

    def viterbi_shard_local(self, model, a_cat, a_first, a_off, b_cat, b_first, b_off, root: int = 0, summary: bool = True, pinned: bool = True,
                            reuse=None):
        """coati_hip_dist_viterbi_shard_local: every rank keeps the results of ITS shard -> (scores, ops, ops_off, ops_len) of
        the pairs [bounds[rank], bounds[rank+1]) (page-locked arrays by default: the download is the rank's own PCIe
        transfer), plus, on root when `summary`, (all_scores, all_len) of every pair (else None, None)."""
        a_cat, b_cat = np.ascontiguousarray(a_cat, np.uint8), np.ascontiguousarray(b_cat, np.uint8)
        a_off, b_off = np.ascontiguousarray(a_off, np.uint64), np.ascontiguousarray(b_off, np.uint64)
        n = len(a_off) - 1
        bounds = hip.shard_bounds(a_off, b_off, self.world).astype(np.int64)
        s0, s1 = int(bounds[self.rank]), int(bounds[self.rank + 1])
        n_loc = s1 - s0
        ops_loc = int(a_off[s1] - a_off[s0] + b_off[s1] - b_off[s0])
        if reuse is not None and len(reuse[0]) == n_loc and len(reuse[1]) >= max(ops_loc, 1):
            scores, ops, off, ln = reuse
        else:
            make = hip.pinned_empty if pinned else (lambda shape, dt: np.zeros(shape, dt))
            scores, ops = make(max(n_loc, 1), np.float32)[:n_loc], make(max(ops_loc, 1), np.uint8)
            off, ln = make(max(n_loc, 1), np.uint64)[:n_loc], make(max(n_loc, 1), np.uint32)[:n_loc]
        all_scores = all_len = None
        if summary and self.rank == root:
            all_scores, all_len = np.zeros(n, np.float32), np.zeros(n, np.uint32)
        _check(load().coati_hip_dist_viterbi_shard_local(self._h, root, model._h, n, hip._ptr(a_cat), int(a_first), hip._ptr(a_off), hip._ptr(b_cat),
                                                         int(b_first), hip._ptr(b_off), hip._ptr(scores), hip._ptr(ops), ops_loc, hip._ptr(off),
                                                         hip._ptr(ln), int(summary), hip._ptr(all_scores), hip._ptr(all_len)))
        return (scores, ops, off, ln), (all_scores, all_len)

    def broadcast_model(self, tables=None, consts=None, gap_len=None, root: int = 0, capacity: int = 64):
        """Root passes (tables [n,183,15], consts[4], gap_len); every rank gets them back bit-identical."""
        if self.rank == root:
            t = np.ascontiguousarray(tables, np.float32).reshape(-1, hip.TABLE_ROWS, hip.TABLE_COLS).copy()
            n = C.c_uint32(t.shape[0])
            k = np.ascontiguousarray(consts, np.float32).copy()
            g = C.c_int(int(gap_len))
        else:
            t = np.zeros((capacity, hip.TABLE_ROWS, hip.TABLE_COLS), np.float32)
            n, k, g = C.c_uint32(0), np.zeros(4, np.float32), C.c_int(0)
        _check(load().coati_hip_dist_broadcast_model(self._h, root, hip._ptr(t), t.shape[0], C.byref(n), hip._ptr(k), C.byref(g)))
        return t[:n.value].copy(), k, g.value

    def gather(self, batch, root: int = 0):
        """Collective gather of one launched hip.Batch (or None); on root returns
        (counts[world,2], scores, ops, ops_off, ops_len), elsewhere (counts, None, None, None, None)."""
        counts = np.zeros(2 * self.world, np.uint64)
        h = batch._h if batch is not None else None
        if self.rank != root:
            _check(load().coati_hip_dist_gather(self._h, root, h, hip._ptr(counts), None, None, 0, None, None))
            return counts.reshape(-1, 2), None, None, None, None
        # capacity is not known before the counts are: every rank's share is at most what this rank holds x world
        # (tests use equal shards); a caller with ragged shards passes through Comm.viterbi instead
        cap_p = max(int(batch.n if batch is not None else 0), 1) * self.world * 2
        cap_o = max(int(batch.ops_total if batch is not None else 0), 1) * self.world * 2
        scores, ops = np.zeros(cap_p, np.float32), np.zeros(cap_o, np.uint8)
        off, ln = np.zeros(cap_p, np.uint64), np.zeros(cap_p, np.uint32)
        _check(load().coati_hip_dist_gather(self._h, root, h, hip._ptr(counts), hip._ptr(scores), hip._ptr(ops), cap_o,
                                            hip._ptr(off), hip._ptr(ln)))
        c = counts.reshape(-1, 2)
        n, nb = int(c[:, 0].sum()), int(c[:, 1].sum())
        return c, scores[:n], ops[:nb], off[:n], ln[:n]

    def viterbi(self, model, a_cat, a_off, b_cat, b_off, root: int = 0):
        """coati_hip_dist_viterbi: the whole sharded job; results on root (None elsewhere)."""
        a_cat, b_cat = np.ascontiguousarray(a_cat, np.uint8), np.ascontiguousarray(b_cat, np.uint8)
        a_off, b_off = np.ascontiguousarray(a_off, np.uint64), np.ascontiguousarray(b_off, np.uint64)
        n = len(a_off) - 1
        total = int(a_off[-1] - a_off[0] + b_off[-1] - b_off[0])
        scores, ops = np.zeros(n, np.float32), np.zeros(max(total, 1), np.uint8)
        off, ln = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
        _check(load().coati_hip_dist_viterbi(self._h, root, model._h, n, hip._ptr(a_cat), hip._ptr(a_off), hip._ptr(b_cat),
                                             hip._ptr(b_off), hip._ptr(scores), hip._ptr(ops), total, hip._ptr(off), hip._ptr(ln)))
        return (scores, ops, off, ln) if self.rank == root else None


def chunk_plan(a_off, b_off, world: int, chunk_cells: int = 0):
    """coati_hip_dist_chunk_plan: ([per-rank chunk boundaries], rounds).  Pure host arithmetic."""
    a_off, b_off = np.ascontiguousarray(a_off, np.uint64), np.ascontiguousarray(b_off, np.uint64)
    n = len(a_off) - 1
    idx = np.zeros(world + 1, np.uint64)
    rounds = C.c_uint64(0)
    cap = n + 2 * world + 2
    cuts = np.zeros(cap, np.uint64)
    _check(load().coati_hip_dist_chunk_plan(n, hip._ptr(a_off), hip._ptr(b_off), world, chunk_cells, hip._ptr(idx), hip._ptr(cuts), cap,
                                            C.byref(rounds)))
    return [cuts[int(idx[r]):int(idx[r + 1])].copy() for r in range(world)], int(rounds.value)


def landing_plan(counts, root: int = 0):
    """coati_hip_dist_landing_plan: (land[world, 4] byte offsets of scores/ops/off/len, need)."""
    counts = np.ascontiguousarray(counts, np.uint64).reshape(-1)
    world = len(counts) // 2
    land = np.zeros(4 * world, np.uint64)
    need = C.c_uint64(0)
    _check(load().coati_hip_dist_landing_plan(world, root, hip._ptr(counts), hip._ptr(land), C.byref(need)))
    return land.reshape(world, 4), int(need.value)


def simulate(world: int, root: int, a_off, b_off, pair_scores, pair_ops, pair_ops_len, chunk_cells: int = 0):
    """coati_hip_dist_simulate: the sharded job of `world` ranks in host memory -> (scores, ops, ops_off, ops_len)."""
    a_off, b_off = np.ascontiguousarray(a_off, np.uint64), np.ascontiguousarray(b_off, np.uint64)
    n = len(a_off) - 1
    total = int(a_off[-1] - a_off[0] + b_off[-1] - b_off[0])
    ps = np.ascontiguousarray(pair_scores, np.float32)
    po = np.ascontiguousarray(pair_ops, np.uint8)
    pl = np.ascontiguousarray(pair_ops_len, np.uint32)
    scores, ops = np.zeros(n, np.float32), np.full(max(total, 1), 0xCC, np.uint8)
    off, ln = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    _check(load().coati_hip_dist_simulate(world, root, n, hip._ptr(a_off), hip._ptr(b_off), chunk_cells, hip._ptr(ps), hip._ptr(po), hip._ptr(pl),
                                          hip._ptr(scores), hip._ptr(ops), total, hip._ptr(off), hip._ptr(ln)))
    return scores, ops, off, ln


def simulate_local(world: int, root: int, a_off, b_off, pair_scores, pair_ops, pair_ops_len, chunk_cells: int = 0, summary: bool = True):
    """coati_hip_dist_simulate_local: every rank keeps its shard -> (scores, ops, ops_off, ops_len, all_scores, all_len); rank r's own
    arrays are the slices that start at its shard (ops_off relative to the rank's own ops array)."""
    a_off, b_off = np.ascontiguousarray(a_off, np.uint64), np.ascontiguousarray(b_off, np.uint64)
    n = len(a_off) - 1
    total = int(a_off[-1] - a_off[0] + b_off[-1] - b_off[0])
    ps = np.ascontiguousarray(pair_scores, np.float32)
    po = np.ascontiguousarray(pair_ops, np.uint8)
    pl = np.ascontiguousarray(pair_ops_len, np.uint32)
    scores, ops = np.zeros(n, np.float32), np.full(max(total, 1), 0xCC, np.uint8)
    off, ln = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    all_scores, all_len = np.full(n, np.nan, np.float32), np.zeros(n, np.uint32)
    _check(load().coati_hip_dist_simulate_local(world, root, n, hip._ptr(a_off), hip._ptr(b_off), chunk_cells, int(summary), hip._ptr(ps), hip._ptr(po),
                                                hip._ptr(pl), hip._ptr(scores), hip._ptr(ops), total, hip._ptr(off), hip._ptr(ln), hip._ptr(all_scores),
                                                hip._ptr(all_len)))
    return scores, ops, off, ln, all_scores, all_len


class HostTransport(C.Structure):
    """coati_hip_dist_host_transport_t over three Python callables (the CPU tests pass torch.distributed / gloo calls)."""
    ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32)
    SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64)
    RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64)
    _fields_ = [("ctx", C.c_void_p), ("allgather", ALLGATHER), ("send", SEND), ("recv", RECV)]


def job_host(transport: HostTransport, world: int, rank: int, root: int, a_off, b_off, pair_scores, pair_ops, pair_ops_len, chunk_cells: int = 0,
             local: bool = False, summary: bool = True):
    """coati_hip_dist_job_host: THIS process is rank `rank` of the sharded job, the exchanges go through `transport`.
    gather-all: (scores, ops, ops_off, ops_len) on root, None elsewhere.  local: ((scores, ops, ops_off, ops_len) of the own
    shard, (all_scores, all_len) on root or (None, None))."""
    a_off, b_off = np.ascontiguousarray(a_off, np.uint64), np.ascontiguousarray(b_off, np.uint64)
    n = len(a_off) - 1
    ps = np.ascontiguousarray(pair_scores, np.float32)
    po = np.ascontiguousarray(pair_ops, np.uint8)
    pl = np.ascontiguousarray(pair_ops_len, np.uint32)
    if local:
        bounds = hip.shard_bounds(a_off, b_off, world).astype(np.int64)
        s0, s1 = int(bounds[rank]), int(bounds[rank + 1])
        n_out, ops_out = s1 - s0, int(a_off[s1] - a_off[s0] + b_off[s1] - b_off[s0])
    else:
        n_out, ops_out = (n, int(a_off[-1] - a_off[0] + b_off[-1] - b_off[0])) if rank == root else (0, 0)
    scores, ops = np.zeros(max(n_out, 1), np.float32)[:n_out], np.full(max(ops_out, 1), 0xCC, np.uint8)
    off, ln = np.zeros(max(n_out, 1), np.uint64)[:n_out], np.zeros(max(n_out, 1), np.uint32)[:n_out]
    all_scores = all_len = None
    if local and summary and rank == root:
        all_scores, all_len = np.full(n, np.nan, np.float32), np.zeros(n, np.uint32)
    _check(load().coati_hip_dist_job_host(C.byref(transport), world, rank, root, n, hip._ptr(a_off), hip._ptr(b_off), chunk_cells, int(local), int(summary),
                                          hip._ptr(ps), hip._ptr(po), hip._ptr(pl), hip._ptr(scores) if n_out else None, hip._ptr(ops), ops_out,
                                          hip._ptr(off) if n_out else None, hip._ptr(ln) if n_out else None, hip._ptr(all_scores), hip._ptr(all_len)))
    if local:
        return (scores, ops, off, ln), (all_scores, all_len)
    return (scores, ops, off, ln) if rank == root else None


def rendezvous_id(world: int, rank: int, make_id=None, timeout_s: float = 300.0) -> bytes:
    """The 128-byte communicator id for all ranks of a job started by `torch.distributed.run` (or by hand), through
    the launcher's TCP store -- BEFORE anything touches a GPU: rank 0 makes the id (`make_id`, default
    coati_hip_dist_unique_id) and sets it, everybody gets it.  Under torch.distributed.run the agent already serves a
    store on MASTER_ADDR:MASTER_PORT (TORCHELASTIC_USE_AGENT_STORE) and the workers are clients; otherwise rank 0 serves."""
    from datetime import timedelta

    from torch.distributed import TCPStore

    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29533"))
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    store = TCPStore(addr, port, world, is_master=(rank == 0 and not agent_store), timeout=timedelta(seconds=timeout_s),
                     wait_for_workers=False)
    key = "coati/uid/" + os.environ.get("TORCHELASTIC_RUN_ID", "0") + "/" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    if rank == 0:
        store.set(key, (make_id or unique_id)())
    uid = bytes(store.get(key))
    if len(uid) != ID_BYTES:
        raise RuntimeError(f"rendezvous: got {len(uid)} bytes for the communicator id")
    # (the store object must outlive the exchange on rank 0 when it serves: keep it until every rank has the id)
    store.add(key + "/got", 1)
    if rank == 0 and not agent_store:
        import time

        t0 = time.time()
        while int(store.add(key + "/got", 0)) < world and time.time() - t0 < timeout_s:
            time.sleep(0.01)
    return uid
