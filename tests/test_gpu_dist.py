"""The native multi-GPU layer (libcoati_hip_dist.so, RCCL linked directly) with a ONE-rank communicator
on the 1-GPU box: rendezvous id, ncclCommInitRank, model broadcast, counts all-gather, the gather's
download path and the sharded driver -- everything but a second peer.  Runs in a child process so that
the RCCL it loads is the one the library links (the pytest process may hold torch's)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import sys, json, zlib, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip, host, dist
table, consts = host.set_subst("mar-ecm"), host.gap_consts()
comm = dist.Comm(dist.unique_id(), 1, 0, 0)
t, k, g = comm.broadcast_model(table, consts, 1)
assert t.shape == (1, 183, 15) and (t[0].view(np.uint32) == np.asarray(table, np.float32).view(np.uint32)).all()
assert (k.view(np.uint32) == np.asarray(consts, np.float32).view(np.uint32)).all() and g == 1
model = hip.Model(t[0], k, g)
a_cat, a_off, b_cat, b_off = host.synth_encoded(0, 300)
batch = hip.Batch(model, a_cat, a_off, b_cat, b_off)
batch.viterbi_launch()
want = batch.viterbi_fetch()
counts, sc, ops, off, ln = comm.gather(batch)
assert counts.tolist() == [[300, int(a_off[-1] + b_off[-1])]]
ok = (sc.view(np.uint32) == want[0].view(np.uint32)).all() and (ops == want[1]).all() and (off == want[2]).all() and (ln == want[3]).all()
batch.close()
got = comm.viterbi(model, a_cat, a_off, b_cat, b_off)
ok = ok and (got[0].view(np.uint32) == want[0].view(np.uint32)).all() and (got[1][:len(want[1])] == want[1]).all()
ok = ok and (got[2] == want[2]).all() and (got[3] == want[3]).all()
empty = comm.gather(None)
ok = ok and empty[0].tolist() == [[0, 0]]
comm.close(); model.close()
print(json.dumps({"ok": bool(ok)}))
'''


def test_one_rank_rccl_path():
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-c", CHILD % str(root)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # (RCCL prints a version banner on stdout when the communicator is created)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"ok"')]
    assert lines and json.loads(lines[-1]) == {"ok": True}, r.stdout[-2000:]


def test_cli_rank_process_equals_single_gpu_batch(tmp_path):
    """`coati-alignpair --batch --devices a,b,...` starts one process per GPU; each is this binary with the
    internal --dist-* flags.  On the 1-GPU box: run such a rank process with a world of one (C++ driver:
    rendezvous file, RCCL communicator, model broadcast, sharded Viterbi, gather, output) and compare
    its JSON with the plain --batch run; `--devices 0` alone is the plain run on that device."""
    root = Path(__file__).resolve().parent.parent
    exe = root / "coati_amd" / "_build" / "coati-alignpair"
    sys.path.insert(0, str(root))
    from coati_amd import host

    fasta = tmp_path / "pairs.fasta"
    with open(fasta, "w") as f:
        for i in range(40):
            anc, des = host.synth_raw(i)
            f.write(f">a{i}\n{anc}\n>d{i}\n{des}\n")
    plain = subprocess.run([str(exe), "--batch", str(fasta)], capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    one = subprocess.run([str(exe), "--batch", str(fasta), "--devices", "0"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0 and one.stdout == plain.stdout, one.stderr[-2000:]
    rank = subprocess.run([str(exe), "--batch", str(fasta), "--dist-rank", "0", "--dist-world", "1", "--dist-id",
                           str(tmp_path / "id")], capture_output=True, text=True, timeout=600)
    assert rank.returncode == 0, rank.stderr[-2000:]
    got = rank.stdout[rank.stdout.index("["):] if "[" in rank.stdout else rank.stdout  # (RCCL's banner precedes the JSON)
    assert json.loads(got) == json.loads(plain.stdout)
    assert len(json.loads(plain.stdout)) == 40


def test_bench_multi_gpu_path_with_one_rank():
    """bench.py's N > 1 code path on the 1-GPU box (COATI_BENCH_SELFTEST_GATHER=1): rendezvous through a TCP
    store, the native communicator, model broadcast, two alternating resident batches with a per-step
    coati_hip_dist_gather, barrier / max-over-ranks through the library, and the `strong_1M` job
    (coati_hip_dist_viterbi_shard; reduced to 3 000 pairs here) -- no torch.distributed process group anywhere."""
    import os

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, COATI_BENCH_SELFTEST_GATHER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "6", "--warmup", "2", "--pairs", "2000", "--no-cpu-baseline",
                        "--no-extras", "--strong-pairs", "3000"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    doc = json.loads(line)
    assert doc["n_gpus"] == 1 and doc["value"] > 100 and doc["scaling"] == "weak"
    st = doc["strong_1M"]
    assert st["local_results"]["equal_to_gathered"] is True
    assert st["pairs"] == 3000 and st["model"] == "mar-ecm" and st["gcups"] > 10 and st["columns"] > 3000 * 990
    assert "gathered results of rank 0 identical to a direct fetch" in r.stderr
    # the strong job's scores are the single-GPU one-shot call's (same generator, same ECM model)
    sys.path.insert(0, str(root))
    import zlib

    import numpy as np

    from coati_amd import hip, host

    model = hip.Model(host.set_subst("mar-ecm"), host.gap_consts(), 1)
    sc = model.viterbi(*host.synth_encoded(0, 3000))[0]
    assert "%08x" % zlib.crc32(np.ascontiguousarray(sc).tobytes()) == st["scores_crc32"]
    model.close()


SHARD_CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from coati_amd import hip, host, dist
comm = dist.Comm(dist.unique_id(), 1, 0, 0)
model = hip.Model(host.set_subst("mar-mg"), host.gap_consts(), 1)
a_cat, a_off, b_cat, b_off = host.synth_encoded(5, 200)
want = comm.viterbi(model, a_cat, a_off, b_cat, b_off)
# the same pairs described by offsets into a LARGER concatenation of which this rank holds only its part
a_first, b_first = 123456, 7890
got = comm.viterbi_shard(model, a_cat, a_first, a_off + np.uint64(a_first), b_cat, b_first, b_off + np.uint64(b_first))
ok = all((g == w).all() for g, w in zip(got, want))
# arrays that start behind the shard: an error on every rank, not a fault and not a hang
try:
    comm.viterbi_shard(model, a_cat, a_first + 3, a_off + np.uint64(a_first), b_cat, b_first, b_off + np.uint64(b_first))
    ok = False
except hip.CoatiHipError:
    pass
# the local-results form (coati_hip_dist_viterbi_shard_local): the rank keeps its shard's results (here: everything), the
# root also gets the summary; with and without the summary, page-locked and pageable arrays
for summary in (True, False):
    for pinned in (True, False):
        loc, (all_s, all_l) = comm.viterbi_shard_local(model, a_cat, a_first, a_off + np.uint64(a_first), b_cat, b_first, b_off + np.uint64(b_first),
                                                       summary=summary, pinned=pinned)
        ok = ok and all((np.asarray(g) == w).all() for g, w in zip((loc[0], loc[2], loc[3]), (want[0], want[2], want[3])))
        ok = ok and (np.asarray(loc[1])[:len(want[1])] == want[1]).all()
        if summary:
            ok = ok and (all_s.view(np.uint32) == want[0].view(np.uint32)).all() and (all_l == want[3]).all()
        else:
            ok = ok and all_s is None
comm.barrier()
ok = ok and comm.allreduce([2.5, -1.0], "max").tolist() == [2.5, -1.0] and comm.allreduce([2.5], "sum").tolist() == [2.5]
comm.close(); model.close()
print(json.dumps({"ok": bool(ok)}))
'''


def test_shard_entry_point_and_small_collectives():
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-c", SHARD_CHILD % str(root)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"ok"')]
    assert lines and json.loads(lines[-1]) == {"ok": True}, r.stdout[-2000:]
