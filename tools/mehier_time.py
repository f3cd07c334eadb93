#!/usr/bin/env python3
"""4K timing: the hierarchical search (one launch, every 16x16 SAD once) against the six per-size searches it replaces (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vvcsoftware_vtm_amd import ops
from vvcsoftware_vtm_amd.workload import Workload, MARGIN

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
wl = Workload(W, H, 10)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
org, ref = d(wl.org[0]), d(wl.ref0_pad[0])
mv = ops.MvCost(wl.mvcost.lambda_, 0, 0, 2, 0)
blks = {s: ops.struct_to_device(b) for s, b in wl.me.items()}


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def per_size():
    out = {}
    for s in (16, 32, 64):
        out[(s, 39)] = ops.sad_search(org, ref, blks[s], wl.me[s].size, s, s, 1, -95, -95, 39, 39, 5, 5, mv, want_sad=False)[1]
        out[(s, 9)] = ops.sad_search(org, ref, blks[s], wl.me[s].size, s, s, 1, -4, -4, 9, 9, 1, 1, mv, want_sad=False)[1]
    return out


def hier(dense=4):
    return ops.me_hier_search(org, ref, (0, 0), (MARGIN, MARGIN), W // 16, H // 16, 1, 96, dense, mv)


want = per_size()
r, dn = hier()
torch.cuda.synchronize()
for k, s in enumerate((16, 32, 64)):
    print("equal raster %d: %s   dense: %s" % (s, torch.equal(r[k], want[(s, 39)]), torch.equal(dn[k], want[(s, 9)])))
print("per-size searches (6 launch groups): %.1f us" % timeit(per_size))
print("hierarchical, raster + dense       : %.1f us" % timeit(hier))
print("hierarchical, raster only          : %.1f us" % timeit(lambda: hier(0)))
os.environ["VVCGPU_MH_DIAG"] = "1"      # (phase stamps: printed by a library built with -DMH_DIAG, e.g. tools/ab_variants.sh build mehier.hip diag "-DMH_DIAG")
hier()
torch.cuda.synchronize()

# the same entry with the caches flushed in front of every call (a 600 MB fill: beyond the 256 MB memory-side cache): what part of the gap between this
# loop and the workload (where the search follows the previous picture's filters) is cache state
junk = torch.empty(300 * 1024 * 1024, dtype=torch.int16, device="cuda")
ts = []
for _ in range(8):
    junk.fill_(1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); hier(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
print("hierarchical, raster + dense, caches flushed in front of every call: %.1f us (min %.1f)" % (sum(ts[2:]) / len(ts[2:]), min(ts)))
