"""The multi-GPU layer without a GPU (libcoati_hip_dist.so's per-rank job loop is one piece of code over an
environment: RCCL + HIP on the GPUs; host memory + a host transport here): the plans as pure functions, every rank of
the sharded job as a thread (coati_hip_dist_simulate*, world 1 .. 8), and TWO PROCESSES over torch.distributed's gloo
backend running the same loop through coati_hip_dist_job_host."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _np_shard_bounds(w, world):
    """numpy statement of coati_hip_shard_bounds' rule: rank r's shard ends at the first pair where the cell prefix
    reaches (r+1)/world of the total."""
    w = np.asarray(w, np.float64)
    pre = np.concatenate([[0.0], np.cumsum(w)])
    total = pre[-1]
    out = [0]
    for r in range(1, world):
        out.append(int(np.searchsorted(pre, total * r / world, side="left")))
    out.append(len(w))
    return [min(max(x, 0), len(w)) for x in out]


def test_native_shard_bounds_matches_and_balances():
    """coati_hip_shard_bounds (the C ABI partitioner bench.py, coati_hip_dist_viterbi and
    `coati-alignpair --devices` use): contiguous, monotone, cell-balanced to within one pair."""
    from coati_amd import hip

    rng = np.random.default_rng(5)
    la = rng.integers(0, 400, 3000) * 3
    lb = rng.integers(0, 1500, 3000)
    a_off = np.concatenate([[0], np.cumsum(la)]).astype(np.uint64)
    b_off = np.concatenate([[0], np.cumsum(lb)]).astype(np.uint64)
    w = (la * lb).astype(np.float64)
    for world in (1, 2, 3, 8, 64):
        b = hip.shard_bounds(a_off, b_off, world).astype(np.int64)
        assert b[0] == 0 and b[-1] == len(w) and (np.diff(b) >= 0).all()
        shares = np.array([w[b[r]:b[r + 1]].sum() for r in range(world)])
        assert shares.max() - w.sum() / world <= w.max() + 1
        ref = np.array(_np_shard_bounds(w, world))
        assert np.abs(b - ref).max() <= 1  # (the same rule up to the side a boundary pair falls on)
    assert hip.shard_bounds(np.zeros(1, np.uint64), np.zeros(1, np.uint64), 4).tolist() == [0, 0, 0, 0, 0]
    # one heavy pair among light ones: it sits alone in its shard, nothing is lost or duplicated
    la2, lb2 = np.full(100, 3), np.full(100, 3)
    la2[40], lb2[40] = 30000, 30000
    a2 = np.concatenate([[0], np.cumsum(la2)]).astype(np.uint64)
    b2 = np.concatenate([[0], np.cumsum(lb2)]).astype(np.uint64)
    bb = hip.shard_bounds(a2, b2, 4)
    assert bb[0] == 0 and bb[-1] == 100 and (np.diff(bb.astype(np.int64)) >= 0).all()


def _gloo_transport(rank, world):
    """coati_hip_dist_host_transport_t over gloo: the three exchanges the native job loop asks for."""
    from coati_amd import dist as nd

    def allgather(_ctx, mine, out, words):
        try:
            t = torch.from_numpy(np.ctypeslib.as_array(mine, shape=(words,)).astype(np.int64))
            parts = [torch.zeros(words, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(parts, t)
            np.ctypeslib.as_array(out, shape=(words * world,))[:] = torch.cat(parts).numpy().astype(np.uint64)
            return 0
        except Exception:  # noqa: BLE001 (an exception must not cross the C frame)
            return 3

    def send(_ctx, peer, data, nbytes):
        try:
            buf = (C.c_uint8 * nbytes).from_address(data)
            dist.send(torch.frombuffer(buf, dtype=torch.uint8).clone(), dst=peer)
            return 0
        except Exception:  # noqa: BLE001
            return 3

    def recv(_ctx, peer, data, nbytes):
        try:
            t = torch.zeros(nbytes, dtype=torch.uint8)
            dist.recv(t, src=peer)
            C.memmove(data, t.numpy().ctypes.data, nbytes)
            return 0
        except Exception:  # noqa: BLE001
            return 3

    tr = nd.HostTransport(None, nd.HostTransport.ALLGATHER(allgather), nd.HostTransport.SEND(send), nd.HostTransport.RECV(recv))
    return tr


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from coati_amd import dist as nd
        from coati_amd import hip

        ok = True
        tr = _gloo_transport(rank, world)
        for root, chunk_cells, seed in ((0, 0, 1), (1, 2500, 2), (0, 60000, 3)):
            la, lb, a_off, b_off, pref, ln, ops, scores = _job(seed, 700)
            want_off = (pref[1:] - ln).astype(np.uint64)
            bounds = hip.shard_bounds(a_off, b_off, world).astype(np.int64)
            s0, s1 = int(bounds[rank]), int(bounds[rank + 1])
            # every rank is given only ITS shard's results (the others' entries are poisoned: never read)
            my_scores = np.full_like(scores, np.nan)
            my_scores[s0:s1] = scores[s0:s1]
            my_ops = np.full_like(ops, 0x77)
            my_ops[pref[s0]:pref[s1]] = ops[pref[s0]:pref[s1]]
            my_ln = np.zeros_like(ln)
            my_ln[s0:s1] = ln[s0:s1]
            # ---- gather-all: the root ends up with every pair, in input order
            got = nd.job_host(tr, world, rank, root, a_off, b_off, my_scores, my_ops, my_ln, chunk_cells)
            if rank == root:
                s, o, off, l = got
                ok = ok and (s.view(np.uint32) == scores.view(np.uint32)).all() and (l == ln).all() and (off == want_off).all()
                ok = ok and all((o[int(off[i]):int(off[i]) + int(l[i])] == ops[int(want_off[i]):pref[i + 1]]).all() for i in range(len(ln)))
            else:
                ok = ok and got is None
            # ---- local: every rank keeps its shard, the root gets the summary
            (s, o, off, l), (all_s, all_l) = nd.job_host(tr, world, rank, root, a_off, b_off, my_scores, my_ops, my_ln, chunk_cells, local=True)
            ok = ok and (s.view(np.uint32) == scores[s0:s1].view(np.uint32)).all() and (l == ln[s0:s1]).all()
            ok = ok and (off == want_off[s0:s1] - np.uint64(pref[s0])).all()
            ok = ok and all((o[int(off[i]):int(off[i]) + int(l[i])] == ops[int(want_off[s0 + i]):pref[s0 + i + 1]]).all() for i in range(s1 - s0))
            if rank == root:
                ok = ok and (all_s.view(np.uint32) == scores.view(np.uint32)).all() and (all_l == ln).all()
            else:
                ok = ok and all_s is None
        # a rank that cannot deliver (more ops than the pair's slot holds) reports it in the count exchange: BOTH
        # ranks return an error from the same round, none is left waiting in a receive
        la, lb, a_off, b_off, pref, ln, ops, scores = _job(9, 60, empty=False)
        bounds = hip.shard_bounds(a_off, b_off, world).astype(np.int64)
        bad = ln.copy()
        victim = int(bounds[1])  # a pair of rank 1's shard
        bad[victim] = la[victim] + lb[victim] + 1
        failed = False
        try:
            nd.job_host(tr, world, rank, 0, a_off, b_off, scores, ops, bad)
        except hip.CoatiHipError:
            failed = True
        ok = ok and failed
        dist.barrier()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_native_job_loop_two_processes_over_gloo():
    """world_size 2, one process per rank, gloo in place of RCCL: the per-rank loop of coati_hip_dist_viterbi_shard
    (chunk plan, status / count exchange, validation against the plan, transfer lists, landing zone, placement, offset
    rebase; gather-all and local forms; a failing rank) through coati_hip_dist_job_host."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


# ---- the native layer's plans (libcoati_hip_dist.so), world 2, 3, 8 without a GPU ---------------------------
def _job(seed, n, max_cod=40, max_b=90, empty=True):
    """lengths, per-pair scores and op slots (ops right-aligned in a slot of la+lb bytes, as the walkers leave them)"""
    rng = np.random.default_rng(seed)
    la = rng.integers(0 if empty else 1, max_cod, n) * 3
    lb = rng.integers(0 if empty else 1, max_b, n)
    a_off = np.concatenate([[0], np.cumsum(la)]).astype(np.uint64)
    b_off = np.concatenate([[0], np.cumsum(lb)]).astype(np.uint64)
    pref = np.concatenate([[0], np.cumsum(la + lb)]).astype(np.int64)
    ln = np.array([rng.integers(max(la[i], lb[i]), la[i] + lb[i] + 1) for i in range(n)], np.uint32)
    ops = np.full(max(int(pref[-1]), 1), 0xAA, np.uint8)
    for i in range(n):
        ops[pref[i + 1] - ln[i]:pref[i + 1]] = rng.integers(0, 3, ln[i])
    scores = rng.normal(size=n).astype(np.float32)
    return la, lb, a_off, b_off, pref, ln, ops, scores


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_native_chunk_plan_tiles_every_shard(world):
    """coati_hip_dist_chunk_plan (what coati_hip_dist_viterbi runs on every rank): a rank's chunks tile its shard
    of coati_hip_shard_bounds, no chunk is empty unless the shard is, none exceeds chunk_cells unless it is a
    single pair, and the round count is the largest chunk count."""
    from coati_amd import dist as nd
    from coati_amd import hip

    la, lb, a_off, b_off, *_ = _job(11 + world, 1500)
    cells = (la * lb).astype(np.int64)
    bounds = hip.shard_bounds(a_off, b_off, world).astype(np.int64)
    for chunk_cells in (1500, 40000, 0):
        cuts, rounds = nd.chunk_plan(a_off, b_off, world, chunk_cells)
        assert len(cuts) == world and rounds == max(len(c) - 1 for c in cuts)
        for r, c in enumerate(cuts):
            c = c.astype(np.int64)
            assert len(c) >= 2 and c[0] == bounds[r] and c[-1] == bounds[r + 1] and (np.diff(c) >= 0).all()
            if bounds[r + 1] > bounds[r]:
                assert (np.diff(c) > 0).all()
            if chunk_cells:
                for k in range(len(c) - 1):
                    assert cells[c[k]:c[k + 1]].sum() <= chunk_cells or c[k + 1] - c[k] == 1
    # more ranks than pairs: the late ranks get empty shards, one (empty) round each
    cuts, rounds = nd.chunk_plan(a_off[:3], b_off[:3], 8, 0)
    assert rounds == 1 and sum(int(c[-1] - c[0]) for c in cuts) == 2


@pytest.mark.parametrize("world,root", [(2, 0), (2, 1), (3, 1), (8, 0), (8, 5)])
def test_native_landing_plan_blocks_are_disjoint(world, root):
    """coati_hip_dist_landing_plan: the four arrays of every peer land 256-byte aligned, in rank order, without
    overlap, inside `need`; the root itself takes nothing."""
    from coati_amd import dist as nd

    rng = np.random.default_rng(world * 10 + root)
    counts = np.stack([rng.integers(0, 5000, world), rng.integers(0, 3_000_000, world)], axis=1).astype(np.uint64)
    counts[rng.integers(0, world)] = 0  # a rank with nothing to send
    land, need = nd.landing_plan(counts, root)
    spans = []
    for r in range(world):
        if r == root:
            continue
        n, ob = int(counts[r, 0]), int(counts[r, 1])
        for at, size in zip(land[r], (4 * n, ob, 8 * n, 4 * n)):
            assert int(at) % 256 == 0
            spans.append((int(at), int(at) + size))
    spans.sort()
    assert all(a1 <= b0 for (_, a1), (b0, _) in zip(spans, spans[1:]))
    assert not spans or spans[-1][1] <= need
    assert need <= sum(e - s for s, e in spans) + 256 * len(spans)


@pytest.mark.parametrize("world,root", [(1, 0), (2, 0), (2, 1), (3, 2), (8, 0), (8, 3)])
@pytest.mark.parametrize("chunk_cells", [0, 2500, 60000])
def test_native_sharded_job_simulated_in_host_memory(world, root, chunk_cells):
    """coati_hip_dist_simulate runs every rank of coati_hip_dist_viterbi as a thread of this process -- the same
    per-rank loop the GPUs run (chunk plan, per-round status / count exchange, validation against the plan, every
    sender's transfer list against the root's receive list, landing zone, placement straight into the caller's
    arrays, offset rebase), with host memory for HBM and an in-process transport for RCCL.  The root's arrays must
    equal the single-process answer for every pair, whatever the world, the root, the number of rounds (ragged:
    ranks run out of chunks at different rounds)."""
    from coati_amd import dist as nd

    la, lb, a_off, b_off, pref, ln, ops, scores = _job(7 * world + root, 1200)
    s, o, off, l = nd.simulate(world, root, a_off, b_off, scores, ops, ln, chunk_cells)
    assert (s.view(np.uint32) == scores.view(np.uint32)).all() and (l == ln).all()
    want_off = (pref[1:] - ln).astype(np.uint64)
    assert (off == want_off).all()
    for i in range(len(ln)):
        assert (o[int(off[i]):int(off[i]) + int(l[i])] == ops[int(want_off[i]):pref[i + 1]]).all(), i


@pytest.mark.parametrize("world,root", [(1, 0), (2, 1), (3, 0), (8, 0), (8, 6)])
@pytest.mark.parametrize("chunk_cells", [0, 2500])
@pytest.mark.parametrize("summary", [True, False])
def test_native_local_results_job_simulated(world, root, chunk_cells, summary):
    """coati_hip_dist_viterbi_shard_local's loop (coati_hip_dist_simulate_local): every rank ends up with the scores,
    ops, op offsets (relative to ITS ops array) and op lengths of its own shard, the root with the summary of all
    pairs when asked for -- and with nothing moved at all when not."""
    from coati_amd import dist as nd
    from coati_amd import hip

    la, lb, a_off, b_off, pref, ln, ops, scores = _job(31 * world + root, 900)
    s, o, off, l, all_s, all_l = nd.simulate_local(world, root, a_off, b_off, scores, ops, ln, chunk_cells, summary)
    bounds = hip.shard_bounds(a_off, b_off, world).astype(np.int64)
    assert (s.view(np.uint32) == scores.view(np.uint32)).all() and (l == ln).all()
    want_off = (pref[1:] - ln).astype(np.int64)
    for r in range(world):
        s0, s1 = int(bounds[r]), int(bounds[r + 1])
        assert (off[s0:s1].astype(np.int64) == want_off[s0:s1] - pref[s0]).all(), r
        mine = o[pref[s0]:pref[s1]]
        for i in range(s0, s1):
            assert (mine[int(off[i]):int(off[i]) + int(l[i])] == ops[int(want_off[i]):pref[i + 1]]).all(), i
    if summary:
        assert (all_s.view(np.uint32) == scores.view(np.uint32)).all() and (all_l == ln).all()
    else:
        assert np.isnan(all_s).all() and not all_l.any()


def test_native_simulation_rejects_inconsistent_input():
    from coati_amd import dist as nd
    from coati_amd import hip

    la, lb, a_off, b_off, pref, ln, ops, scores = _job(3, 50, empty=False)
    bad = ln.copy()
    bad[7] = la[7] + lb[7] + 1  # more ops than the pair's slot holds
    with pytest.raises(hip.CoatiHipError):
        nd.simulate(2, 0, a_off, b_off, scores, ops, bad)
    dec = a_off.copy()
    dec[5] = dec[4] - 3  # decreasing offsets: the partitioner refuses, the same way on every rank
    with pytest.raises(hip.CoatiHipError):
        nd.chunk_plan(dec, b_off, 2)


def test_rendezvous_through_the_launcher_store(tmp_path):
    """bench.py's N > 1 path gets its communicator id through torch.distributed.run's TCP store before any GPU call
    (coati_amd/dist.py: rendezvous_id).  Two ranks under the real launcher, and one rank by hand (rank 0 serves)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rdv.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "from coati_amd import dist\n"
        "world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))\n"
        "uid = dist.rendezvous_id(world, rank, make_id=lambda: bytes((7 * i + 3) % 256 for i in range(128)), timeout_s=60)\n"
        "assert uid == bytes((7 * i + 3) % 256 for i in range(128))\n"
        # one file per rank: the launcher merges the ranks' stdout, and two concurrent prints interleave
        f"open(os.path.join({str(tmp_path)!r}, 'rank%d.ok' % rank), 'w').write('ok')\n")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert (tmp_path / "rank0.ok").read_text() == "ok" and (tmp_path / "rank1.ok").read_text() == "ok"
    (tmp_path / "rank0.ok").unlink()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 1))
    for k in ("RANK", "WORLD_SIZE", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and (tmp_path / "rank0.ok").read_text() == "ok", r.stdout[-1000:] + r.stderr[-1000:]
