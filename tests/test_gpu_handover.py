"""The N > 1 hand-over on real device tensors through RCCL, on the ONE GPU a test box has: a fresh child process initialises the `nccl` backend
the way bench.py does (eagerly, `device_id=`) at world size 1 and hands the boundary picture to ITSELF through the backend
(`shard.Handover(loopback=True)`: the same grouped `batch_isend_irecv` of receives + sends the ring issues at N > 1, peer = own rank), then installs
it as a reference picture (device copy + vvcgpu_extend_border).  What this can and cannot show: the tensor / stream / lifetime handling of the RCCL
path and that the hand-over never reaches the backend unbatched -- not the ring order between different ranks (tests/test_multiproc_cpu.py walks
that on gloo with the same grouped code path, and test_no_unbatched_p2p_on_rccl pins the call pattern)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from vvcsoftware_vtm_amd import capi, ops, shard
torch.cuda.set_device(0)
capi.call("vvcgpu_set_device", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = torch.Generator(device="cuda").manual_seed(7)
h_, w_ = 272, 480
planes = [torch.randint(0, 1024, s, generator=g, device="cuda", dtype=torch.int32).to(torch.int16) for s in ((h_, w_), (h_ // 2, w_ // 2), (h_ // 2, w_ // 2))]
side = torch.cuda.Stream()
res = {"rounds": 0}
for rnd in range(3):
    # the picture is produced on the current stream right before the hand-over (as the ALF output is): the transfer must be ordered behind it
    src = [p + rnd for p in planes]
    rec = shard.empty_side_record()
    rec["sub_merge_blk_num"][0, 2] = 40 + rnd
    h = shard.Handover(src, 0, 1, loopback=True).post_recv()
    assert h.issued == [], h.issued                      # nothing may reach RCCL before the boundary picture exists
    h.send(src, rec)
    assert h.issued == [("batch", 8)], h.issued          # ONE grouped operation: 4 receives + 4 sends
    # work of the "next chunk" that does not need the picture runs meanwhile
    with torch.cuda.stream(side):
        busy = torch.ones(1 << 20, device="cuda").cumsum(0)
    got, grec = h.wait()
    assert all(bool((a == b).all()) for a, b in zip(got, src))
    assert int(grec["sub_merge_blk_num"][0, 2]) == 40 + rnd
    # receiving side of the hand-over: padded reference slot + border extension on the device
    margins = [(16, 16), (8, 8), (8, 8)]
    ref = [torch.zeros((p.shape[0] + 2 * my, p.shape[1] + 2 * mx), dtype=torch.int16, device="cuda") for p, (mx, my) in zip(src, margins)]
    shard.install_reference(got, ref, margins)
    torch.cuda.synchronize()
    for p, r, (mx, my) in zip(src, ref, margins):
        want = np.pad(p.cpu().numpy(), ((my, my), (mx, mx)), mode="edge")
        assert (r.cpu().numpy() == want).all()
    res["rounds"] += 1
# the unbatched form is refused on this backend
try:
    shard.Handover(planes, 0, 1, loopback=True, batched=False)
    res["refused"] = False
except RuntimeError:
    res["refused"] = True
res["backend"] = dist.get_backend()
dist.barrier()
dist.destroy_process_group()
print(json.dumps(res))
'''


@pytest.mark.gpu
def test_handover_loopback_on_rccl():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a fresh child with its own bounded lifetime: a hand-over that waits on itself ends in a test failure, not in a hung test run
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "port": port}], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res == {"rounds": 3, "refused": True, "backend": "nccl"}


CHILD2 = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from vvcsoftware_vtm_amd import capi, shard
rank, world = int(sys.argv[1]), 2
torch.cuda.set_device(0)
capi.call("vvcgpu_set_device", 0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
g = torch.Generator(device="cuda").manual_seed(11 + rank)
planes = [torch.randint(0, 1024, s, generator=g, device="cuda", dtype=torch.int32).to(torch.int16) for s in ((136, 240), (68, 120), (68, 120))]
ok = True
for rnd in range(2):
    rec = shard.empty_side_record()
    rec["prev_poc"] = 100 * rank + rnd
    h = shard.Handover(planes, rank, world, batched=True).post_recv()       # the grouped form RCCL takes, between two different ranks
    ok = ok and h.issued == []
    got, grec = h.send(planes, rec).wait()
    ok = ok and h.issued == [("batch", 8)] and int(grec["prev_poc"][0]) == 100 * (1 - rank) + rnd
    # what the other rank holds: regenerate its planes here
    g2 = torch.Generator(device="cuda").manual_seed(11 + (1 - rank))
    want = [torch.randint(0, 1024, s, generator=g2, device="cuda", dtype=torch.int32).to(torch.int16) for s in ((136, 240), (68, 120), (68, 120))]
    ok = ok and all(bool((a == b).all()) for a, b in zip(got, want))
    margins = [(16, 16), (8, 8), (8, 8)]
    ref = [torch.zeros((p.shape[0] + 2 * my, p.shape[1] + 2 * mx), dtype=torch.int16, device="cuda") for p, (mx, my) in zip(got, margins)]
    shard.install_reference(got, ref, margins)
    torch.cuda.synchronize()
    ok = ok and all((r.cpu().numpy() == np.pad(p.cpu().numpy(), ((my, my), (mx, mx)), mode="edge")).all() for p, r, (mx, my) in zip(want, ref, margins))
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"rank": rank, "ok": bool(ok)}))
'''


@pytest.mark.gpu
def test_handover_two_ranks_device_tensors_gloo():
    """two ranks, device tensors, the grouped hand-over between DIFFERENT ranks (rank 0 <-> rank 1) followed by install_reference -- over gloo, because
    RCCL refuses two ranks on one GPU and a test box has one: the ring order and the device-side handling of the N = 2 case, not the RCCL transport"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-c", CHILD2 % {"root": ROOT, "port": port}, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = []
    for p_ in procs:
        try:
            o, e = p_.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        if p_.returncode != 0 and "gloo" in e and ("not supported" in e or "No backend type" in e or "CUDA" in e and "gloo" in e.lower() and "support" in e.lower()):
            pytest.skip("this torch build's gloo does not take device tensors: " + e[-300:])
        assert p_.returncode == 0, e[-3000:]
        outs.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
    assert sorted((o["rank"], o["ok"]) for o in outs) == [(0, True), (1, True)]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """`bench.py --gpus 2 --rehearse`: the whole N = 2 control flow of the bench on the one GPU of a test box (gloo instead of RCCL, both ranks on
    device 0): launcher, rank / world checks, chunk steps with the grouped hand-over installed in front of the first motion compensation, barriers,
    max-over-ranks time, hash gather, input-stream leg -- one JSON line from rank 0.  Not a measurement (gloo stages device tensors through the host)."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "1", "--pictures-per-step", "2",
                        "--width", "1920", "--height", "1080", "--rotate", "2", "--no-cpu-baseline", "--no-real-mix"],
                       capture_output=True, text=True, env=env, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["steps"] == 1 and d["value"] > 0
    assert d["picture_hashes"]["gathered"] == 2 and d["input_stream"]["value"] > 0
    assert d["config"]["parallelism"].startswith("one chunk stream per GPU")
