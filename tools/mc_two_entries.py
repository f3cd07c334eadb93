import sys, os
sys.path.insert(0, "/root/repo")
import torch
from vvcsoftware_vtm_amd import ops
from vvcsoftware_vtm_amd.workload import Workload
wl = Workload(3840, 2160, 10)
st, _ = wl.run_gpu(None, None)
torch.cuda.synchronize()
mx = 1023
for name in ("mc_batch", "mc_picture_batch"):
    fn = lambda: getattr(ops, name)(st["ref0"][0], st["ref1"][0], st["pred"][0], st["mc_pic"], wl.mc_pic.size, 10, (0, mx))
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    print(name, a.elapsed_time(b) / 10 * 1e3, "us")
