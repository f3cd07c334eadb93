#!/usr/bin/env python3
"""Phase cycles of the ring raster kernel (workgroup 0): VVCGPU_R7_DIAG=1 makes vvcgpu_sad_search print them.  usage: python tools/r7_diag.py [sizes...]"""
import os
import sys
os.environ["VVCGPU_R7_DIAG"] = "1"
os.environ["VVCGPU_R7"] = "1"
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import ops  # noqa: E402
from vvcsoftware_vtm_amd.workload import Workload  # noqa: E402
sizes = [int(x) for x in sys.argv[1:]] or [32, 64, 16]
wl = Workload(3840, 2160, 10, seed=1, me_sizes=tuple(sizes))
org = torch.from_numpy(wl.org[0]).cuda()
refp = torch.from_numpy(wl.ref0_pad[0]).cuda()
dx0, dy0, nx, ny, sx, sy = wl.me_grids[1]
mv = ops.MvCost(wl.mvcost.lambda_, 3, -5, 2, 0)
for s in sizes:
    b = wl.me[s]
    bd = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
    for _ in range(2):
        ops.sad_search(org, refp, bd, b.size, s, s, 1, dx0, dy0, nx, ny, sx, sy, mv, want_sad=False)
        torch.cuda.synchronize()
