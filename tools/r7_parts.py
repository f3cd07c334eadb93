#!/usr/bin/env python3
"""Timing-only ablations of the ring raster kernel (VVCGPU_R7_DBG bit mask, read per launch): which part of a step costs what.
usage: python tools/r7_parts.py [sizes...]"""
import os
import sys

os.environ["VVCGPU_R7"] = "1"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import ops  # noqa: E402
from vvcsoftware_vtm_amd.workload import Workload  # noqa: E402


def timeit(fn, n=10):
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
    sizes = [int(x) for x in sys.argv[1:]] or [32, 64, 16]
    wl = Workload(3840, 2160, 10, seed=1, me_sizes=tuple(sizes))
    org = torch.from_numpy(wl.org[0]).cuda()
    refp = torch.from_numpy(wl.ref0_pad[0]).cuda()
    dx0, dy0, nx, ny, sx, sy = wl.me_grids[1]
    mv = ops.MvCost(wl.mvcost.lambda_, 3, -5, 2, 0)
    masks = [(0, "full"), (32, "every block reads block 0's packed rows (scalar cache hits)"), (16, "full + run lines touched up front"), (1, "no SAD loop"), (2, "no arg-min pass"), (4, "no DMA"), (8, "no final reduce"), (5, "no SAD loop, no DMA"), (7, "barriers + bookkeeping only"),
             (6, "SAD loop only (no DMA, no arg-min)")]
    for s in sizes:
        b = wl.me[s]
        bd = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
        fn = lambda: ops.sad_search(org, refp, bd, b.size, s, s, 1, dx0, dy0, nx, ny, sx, sy, mv, want_sad=False)
        for m, label in masks:
            os.environ["VVCGPU_R7_DBG"] = str(m)
            ts = [timeit(fn) for _ in range(3)]
            print("raster %dx%d dbg %2d %-36s %.1f us" % (s, s, m, label, 1e3 * min(ts)), flush=True)
        os.environ.pop("VVCGPU_R7_DBG", None)


if __name__ == "__main__":
    main()
