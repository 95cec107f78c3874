"""Multi-GPU plumbing: one process per GPU, pairs sharded contiguously, the model
broadcast from rank 0 and the results gathered to rank 0 with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The path has no data-path collective: pairs are independent (the reference
aligns one pair per process, src/lib/utils.cc:810-812).  The only exchanges are
the ~11 KB model blob before the first batch and the result gather after it.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

TABLE_ELEMS = 183 * 15
MODEL_BLOB_ELEMS = TABLE_ELEMS + 4 + 1  # table, 4 log gap constants, gap_len


def shard_bounds(weights, world: int):
    """Contiguous shards of ~equal total weight (cells).  Returns world+1 pair indices."""
    w = np.asarray(weights, np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(cum, total * r / world, side="left")))
    bounds.append(len(w))
    return [min(max(b, 0), len(w)) for b in np.maximum.accumulate(bounds)]


def pack_model(table, consts, gap_len: int) -> torch.Tensor:
    blob = np.zeros(MODEL_BLOB_ELEMS, np.float32)
    blob[:TABLE_ELEMS] = np.asarray(table, np.float32).ravel()
    blob[TABLE_ELEMS:TABLE_ELEMS + 4] = np.asarray(consts, np.float32)
    blob[-1] = float(gap_len)
    return torch.from_numpy(blob)


def unpack_model(blob: torch.Tensor):
    v = blob.detach().cpu().numpy()
    return v[:TABLE_ELEMS].reshape(183, 15).copy(), v[TABLE_ELEMS:TABLE_ELEMS + 4].copy(), int(v[-1])


def broadcast_model(table, consts, gap_len, device, src: int = 0):
    """Rank `src` supplies (table, consts, gap_len); every rank returns them bit-identical."""
    if dist.get_rank() == src:
        blob = pack_model(table, consts, gap_len).to(device)
    else:
        blob = torch.zeros(MODEL_BLOB_ELEMS, dtype=torch.float32, device=device)
    dist.broadcast(blob, src=src)
    return unpack_model(blob)


def gather_ragged(t: torch.Tensor, dst: int = 0):
    """Gather 1-D tensors of different lengths to rank dst (list of tensors there, None elsewhere)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    padded = t if t.numel() == cap else torch.cat([t, t.new_zeros(cap - t.numel())])
    bucket = [torch.empty(cap, dtype=t.dtype, device=t.device) for _ in range(world)] if rank == dst else None
    dist.gather(padded.contiguous(), bucket, dst=dst)
    if rank != dst:
        return None
    return [b[:s] for b, s in zip(bucket, sizes)]


class _DeviceArray:
    """Zero-copy view of library-owned HBM for torch (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def batch_result_tensors(batch, device):
    """torch views (no copy) of a Batch's result arrays: scores f32, ops u8, ops_off i64, ops_len i32."""
    sc, ops, nbytes, off, ln = batch.result_ptrs()
    n = batch.n
    mk = lambda ptr, cnt, ts: torch.as_tensor(_DeviceArray(ptr, max(cnt, 1), ts), device=device)[:cnt]
    return mk(sc, n, "<f4"), mk(ops, nbytes, "|u1"), mk(off, n, "<i8"), mk(ln, n, "<i4")


def gather_results(batch, device, dst: int = 0):
    """The 'final gather over xGMI': every rank's scores / ops / offsets / lengths to rank dst."""
    sc, ops, off, ln = batch_result_tensors(batch, device)
    return tuple(gather_ragged(t, dst) for t in (sc, ops, off, ln))


class PackedGather:
    """Result gather for a loop over the same batch shape: ONE collective per call.

    The four result arrays (scores f32, ops u8, ops_off i64, ops_len i32) are copied device-to-
    device into one byte buffer and gathered to rank `dst`; sizes are exchanged once at
    construction, receive buffers are allocated once.  The call only enqueues work on torch's
    collective stream, so it overlaps with kernels the library runs on its own stream.
    """

    def __init__(self, tensors, dst: int = 0):
        self.views = [t.contiguous().view(torch.uint8) if t.dtype != torch.uint8 else t for t in tensors]
        self.dtypes = [t.dtype for t in tensors]
        self.sizes = [int(v.numel()) for v in self.views]
        self.dst = dst
        world, rank = dist.get_world_size(), dist.get_rank()
        dev = tensors[0].device
        mine = torch.tensor(self.sizes, dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        self.all_sizes = [[int(x) for x in e.tolist()] for e in every]
        self.cap = max(max(sum(self._pad(n) for n in sz) for sz in self.all_sizes), 8)
        self.packed = torch.zeros(self.cap, dtype=torch.uint8, device=dev)
        self.bucket = [torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
        self.work = None

    @staticmethod
    def _pad(n: int) -> int:
        return (n + 7) // 8 * 8  # every segment starts 8-byte aligned (typed views of the buffer)

    def __call__(self, async_op: bool = False):
        if self.work is not None:  # the previous gather of this buffer must be done before it is refilled
            self.work.wait()
            self.work = None
        pos = 0
        for v, n in zip(self.views, self.sizes):
            self.packed[pos:pos + n].copy_(v)
            pos += self._pad(n)
        # The copies above read the library's result arrays on torch's current stream; the library relaunches
        # the batch on ITS stream, which knows nothing of them.  wait_copies() (host wait on this event) is
        # what a caller does before relaunching the batch whose arrays were packed here.
        if self.packed.is_cuda:
            self.copied = torch.cuda.Event()
            self.copied.record()
        work = dist.gather(self.packed, self.bucket, dst=self.dst, async_op=True)
        if async_op:
            self.work = work
            return work
        work.wait()  # (device-side dependency on NCCL; does not block the host there)
        return None

    def wait_copies(self):
        """Block the host until the result arrays this gather packed have been read (then the batch may be relaunched)."""
        ev = getattr(self, "copied", None)
        if ev is not None:
            ev.synchronize()

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def unpack(self, rank: int):
        """On `dst`: rank's (scores, ops, ops_off, ops_len) as typed views of the receive buffer."""
        out, pos = [], 0
        for n, dt in zip(self.all_sizes[rank], self.dtypes):
            out.append(self.bucket[rank][pos:pos + n].view(dt) if dt != torch.uint8 else self.bucket[rank][pos:pos + n])
            pos += self._pad(n)
        return tuple(out)
