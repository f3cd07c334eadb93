"""N > 1 host path on CPU (gloo, world_size 2): intra-period sharding, the point-to-point boundary-picture hand-over and
the gather of per-picture hashes (vvcsoftware_vtm_amd/shard.py).  No GPU, no data-path collective."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from vvcsoftware_vtm_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_pictures = 64 * world                              # two intra periods per rank
    mine = shard.chunk_assignment(n_pictures, world)[rank]
    hashes = {}
    g = torch.Generator().manual_seed(100 + rank)
    last = None
    for (a, b) in mine:
        for poc in range(a, b):
            planes = [torch.full((8, 8), poc, dtype=torch.int16), torch.full((4, 4), poc + 1, dtype=torch.int16)]
            hashes[poc] = shard.picture_hash(planes)
            last = planes
    # hand-over: every rank passes the last reconstructed picture of its chunk AND the side record to the next rank.  Asynchronous form: the
    # receives are posted first, "work" happens, the sends follow, the wait comes last -- and rank 0 sends late on purpose: nobody but its
    # successor may be held up by that
    h = shard.Handover(last, rank, world).post_recv()
    rec = shard.empty_side_record()
    rec["sub_merge_blk_size"][0, 2] = 7000 + rank
    rec["prev_poc"] = mine[-1][1] - 1
    if rank == 0:
        import time
        time.sleep(0.5)
    h.send(last, rec)
    got, grec = h.wait()
    prev_last_poc = shard.chunk_assignment(n_pictures, world)[(rank - 1) % world][-1][1] - 1
    ok = int(got[0][0, 0]) == prev_last_poc and int(got[1][0, 0]) == prev_last_poc + 1
    ok = ok and int(grec["sub_merge_blk_size"][0, 2]) == 7000 + (rank - 1) % world and int(grec["prev_poc"][0]) == prev_last_poc
    ok = ok and h.issued == [("irecv", 1)] * 3 + [("isend", 1)] * 3
    # the synchronous wrapper gives the same picture
    got2 = shard.exchange_boundary(last, rank, world)
    ok = ok and bool((got2[0] == got[0]).all())
    # the RCCL form on gloo: nothing posted early, ONE grouped operation (3 receives + 3 sends) when the boundary picture exists
    hb = shard.Handover(last, rank, world, batched=True).post_recv()
    ok = ok and hb.issued == []
    got3, grec3 = hb.send(last, rec).wait()
    ok = ok and hb.issued == [("batch", 6)] and bool((got3[0] == got[0]).all()) and bool((got3[1] == got[1]).all())
    ok = ok and grec3.tobytes() == grec.tobytes()
    merged = shard.gather_hashes(hashes, world)
    dist.barrier()
    if rank == 0:
        q.put((ok, len(merged), sorted(merged)[:3], sorted(merged)[-1]))
    else:
        q.put((ok,))
    dist.destroy_process_group()


def test_chunk_assignment_covers_all_pictures_once():
    sys.path.insert(0, ROOT)
    from vvcsoftware_vtm_amd import shard
    for world in (1, 2, 3, 8):
        a = shard.chunk_assignment(65, world)
        pocs = sorted(p for r in a for (s, e) in r for p in range(s, e))
        assert pocs == list(range(65))
        # chunks are whole intra periods and round-robin over ranks
        assert all(s % shard.INTRA_PERIOD == 0 for r in a for (s, e) in r)
        assert shard.boundary_owner(3, world) == 3 % world


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_shard_exchange_gather(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    full = [r for r in res if len(r) > 1][0]
    assert full[1] == 64 * world and full[2] == [0, 1, 2] and full[3] == 64 * world - 1


def test_side_record_layout():
    sys.path.insert(0, ROOT)
    from vvcsoftware_vtm_amd import shard
    r = shard.empty_side_record()
    assert r.nbytes == 88 and int(r["prev_poc"][0]) == 0xFFFFFFFF
    t = shard.side_record_tensor(r)
    assert t.dtype == torch.uint8 and t.numel() == 88
    assert shard.side_record_from_tensor(t).tobytes() == r.tobytes()


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no torch.distributed environment starts the two ranks itself (before anything touches a GPU);
    --dry-launch keeps the ranks on gloo/CPU: both join, the world size is checked, a boundary picture goes round the ring."""
    rc, res, err = _run_bench(["--gpus", "2", "--dry-launch"])
    assert rc == 0, err
    assert res == {"dry_launch": True, "n_gpus": 2, "ranks_seen": [0, 1], "handover_ok": True, "p2p": "one batch_isend_irecv per hand-over"}


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus N inside a launcher environment with a different WORLD_SIZE fails loudly instead of measuring one GPU"""
    rc, res, err = _run_bench(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and res is None and "--gpus 2" in err
    rc, res, err = _run_bench(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and res is None and "WORLD_SIZE is 2" in err


def test_no_unbatched_p2p_on_rccl(monkeypatch):
    """On an RCCL group initialised with `device_id=` unbatched point-to-point operations are serialised with everything else on the communicator:
    a receive posted at chunk start would sit in front of the rank's own send on every rank of the ring.  `shard.Handover` must therefore reach the
    backend through ONE dist.batch_isend_irecv per hand-over and never through dist.isend / dist.irecv -- checked here with the backend reported as
    nccl and the three entry points replaced by recorders (no GPU needed)."""
    sys.path.insert(0, ROOT)
    from vvcsoftware_vtm_amd import shard
    calls = []

    class _Work:
        def wait(self):
            calls.append("wait")

    monkeypatch.setattr(shard, "_backend", lambda group=None: "nccl")
    monkeypatch.setattr(dist, "isend", lambda *a, **k: (_ for _ in ()).throw(AssertionError("unbatched isend on RCCL")))
    monkeypatch.setattr(dist, "irecv", lambda *a, **k: (_ for _ in ()).throw(AssertionError("unbatched irecv on RCCL")))

    def batch(ops):
        calls.append(("batch", [(o.op.__name__ if hasattr(o.op, "__name__") else "op", o.peer) for o in ops]))
        return [_Work()]

    class _Op:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    monkeypatch.setattr(dist, "P2POp", _Op)
    monkeypatch.setattr(dist, "batch_isend_irecv", batch)
    planes = [torch.zeros((8, 8), dtype=torch.int16), torch.zeros((4, 4), dtype=torch.int16), torch.zeros((4, 4), dtype=torch.int16)]
    for world, rank in ((2, 0), (2, 1), (8, 3)):
        calls.clear()
        h = shard.Handover(planes, rank, world).post_recv()
        assert calls == [] and h.batched
        h.send(planes, shard.empty_side_record())
        assert len(calls) == 1 and calls[0][0] == "batch"
        peers = [p for _, p in calls[0][1]]
        assert peers == [(rank - 1) % world] * 4 + [(rank + 1) % world] * 4          # receives first, then sends: one group
        h.wait()
        assert calls[-1] == "wait"
    # asking for the unbatched form on RCCL is refused outright
    with pytest.raises(RuntimeError):
        shard.Handover(planes, 0, 2, batched=False)
    with pytest.raises(RuntimeError):
        shard._unbatched(dist.irecv, planes[0], 1)
