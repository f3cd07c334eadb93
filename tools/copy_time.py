"""B4 copyFrom at plane size: torch's copy_ (the runtime's blit kernel) against vvcgpu_pelop_batch op 5 (copyClip, one descriptor per 128x128 band)
on a 1920x1080 chroma plane.  usage (GPU box): python tools/copy_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops
from vvcsoftware_vtm_amd.workload import Workload

wl = Workload(3840, 2160)
w, h = 1920, 1080
src = torch.randint(0, 1023, (h, w), dtype=torch.int16, device="cuda")
dst = torch.empty_like(src)
bands = ops.struct_to_device(wl.bands_chroma)
cfg = ops.PelopCfg(0, 0, 0, 1, 0, 1023)

def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

print("torch copy_      : %.1f us" % timed(lambda: dst.copy_(src)))
print("pelop copyClip   : %.1f us" % timed(lambda: ops.pelop_batch(5, src, src, dst, bands, wl.bands_chroma.size, cfg)))
assert torch.equal(dst, src)
v = src.view(-1); d = dst.view(-1)
print("torch add out=   : %.1f us" % timed(lambda: torch.add(v, 0, out=d)))
