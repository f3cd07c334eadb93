#!/usr/bin/env python3
"""A/B: ring raster kernel (csrc/raster7.hip) against the strip kernels of csrc/dist.hip on the raster launches of the canonical 4K workload.
Results (x, y, cost, sad of every block) are compared first, then both forms are timed in interleaved rounds in one process.
usage: python tools/r7_ab.py [sizes...]          VVCGPU_R7 is toggled per call (read per call by vvcgpu_sad_search)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import ops  # noqa: E402
from vvcsoftware_vtm_amd.workload import Workload  # noqa: E402


def run(fn, r7):
    if r7:
        os.environ["VVCGPU_R7"] = "1"
    else:
        os.environ.pop("VVCGPU_R7", None)
    return fn()


def timeit(fn, r7, n=10):
    run(fn, r7)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        run(fn, r7)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [32, 64, 16]
    w, h = (int(os.environ.get("R7_W", 3840)), int(os.environ.get("R7_H", 2160)))
    wl = Workload(w, h, 10, seed=1, me_sizes=tuple(sizes))
    org = torch.from_numpy(wl.org[0]).cuda()
    refp = torch.from_numpy(wl.ref0_pad[0]).cuda()
    dx0, dy0, nx, ny, sx, sy = wl.me_grids[1]
    mv = ops.MvCost(wl.mvcost.lambda_, 3, -5, 2, 0)
    ok = True
    for s in sizes:
        b = wl.me[s]
        bd = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
        fn = lambda: ops.sad_search(org, refp, bd, b.size, s, s, 1, dx0, dy0, nx, ny, sx, sy, mv, want_sad=False)
        _, best7 = run(fn, True)
        torch.cuda.synchronize()
        r7 = best7.cpu().numpy().copy()
        _, best5 = run(fn, False)
        torch.cuda.synchronize()
        r5 = best5.cpu().numpy().copy()
        same = np.array_equal(r7, r5)
        ok &= same
        print("raster %dx%d: %d blocks, results %s" % (s, s, b.size, "identical" if same else "DIFFER"), flush=True)
        if not same:
            bad = np.nonzero((r7.view(np.uint8).reshape(b.size, -1) != r5.view(np.uint8).reshape(b.size, -1)).any(axis=1))[0]
            print("  %d blocks differ, first: %s" % (bad.size, bad[:8]))
            for i in bad[:4]:
                print("   block %d r7 %s strip %s" % (i, r7.view(np.uint8).reshape(b.size, -1)[i].view(np.int64), r5.view(np.uint8).reshape(b.size, -1)[i].view(np.int64)))
        t7, t5 = [], []
        for _ in range(5):
            t7.append(timeit(fn, True))
            t5.append(timeit(fn, False))
        print("  ring form %.1f us (min %.1f)   strip form %.1f us (min %.1f)   [launch groups incl. packing / decode]" %
              (1e3 * float(np.median(t7)), 1e3 * min(t7), 1e3 * float(np.median(t5)), 1e3 * min(t5)), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
