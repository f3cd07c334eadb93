#!/usr/bin/env python3
"""Experiment: where does the raster kernel (sad_raster5c_kernel) lose time on wide blocks?  Times the +-95 step-5 raster of every
s x s block of a 3840x2160 picture (a) with the real block list and (b) with every block reading the SAME org block (the scalar
cache then always hits for the wave-uniform org rows).  usage: python tools/raster_probe.py [sizes...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import ops  # noqa: E402
from vvcsoftware_vtm_amd.workload import Workload, SEARCH_BLK  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [16, 32, 64]
    wl = Workload(3840, 2160, 10, seed=1, me_sizes=tuple(sizes))
    org = torch.from_numpy(wl.org[0]).cuda()
    refp = torch.from_numpy(wl.ref0_pad[0]).cuda()
    dx0, dy0, nx, ny, sx, sy = wl.me_grids[1]
    mv = ops.MvCost(wl.mvcost.lambda_, 0, 0, 2, 0)
    for s in sizes:
        blk = wl.me[s]
        same = blk.copy()
        same["org_x"], same["org_y"] = 64, 64
        for label, b in (("real", blk), ("same-org", same)):
            bd = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
            ms = timeit(lambda: ops.sad_search(org, refp, bd, b.size, s, s, 1, dx0, dy0, nx, ny, sx, sy, mv, want_sad=False))
            print("raster %dx%d %-9s %d blocks: %.4f ms" % (s, s, label, b.size, ms), flush=True)


if __name__ == "__main__":
    main()
