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
