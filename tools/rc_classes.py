#!/usr/bin/env python3
"""Per-class cost of the residual chain inside the canonical picture layout: one vvcgpu_resi_chain_batch call per TU size (64 .. 4 squared, 1.66 M luma
samples each = the canonical workload's share, TUs tiled row-major over a 3840-wide plane), `reps` rounds in a fixed order.  Run under
rocprofv3 (tools/rc_classes.sh): the reducer groups the rc_chain_kernel launches by their position in the round.
usage: python3 tools/rc_classes.py [reps]      (prints event-timed ms per call as well)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops  # noqa: E402

SIZES = [64, 32, 16, 8, 4]
W, H, bd = 3840, 432, 10          # 3840 x 432 = 1 658 880 samples per class


def build(s, rng):
    nx, ny = W // s, (H // s)
    xs, ys = np.meshgrid(np.arange(nx) * s, np.arange(ny) * s)
    x, y = xs.ravel(), ys.ravel()
    n = x.size
    d = np.zeros(n, ops.RC_DESC)
    d["org_off"] = d["pred_off"] = d["rec_off"] = y.astype(np.int64) * W + x
    d["org_stride"] = d["pred_stride"] = d["rec_stride"] = W
    d["level_off"] = np.arange(n, dtype=np.int64) * s * s
    d["w"] = d["h"] = s
    if s <= 32:
        d["tr_hor"], d["tr_ver"] = rng.integers(0, 3, n), rng.integers(0, 3, n)
    d["qp"], d["sign_hiding"] = 32 + 12, 1
    org = torch.from_numpy(rng.integers(0, 1 << bd, (H, W), dtype=np.int16)).cuda()
    pred = torch.clamp(org + torch.from_numpy(rng.integers(-40, 41, (H, W), dtype=np.int16)).cuda(), 0, (1 << bd) - 1).to(torch.int16)
    rec = torch.zeros((H, W), dtype=torch.int16, device="cuda")
    level = torch.zeros(n * s * s, dtype=torch.int32, device="cuda")
    dd = ops.struct_to_device(d)
    return lambda: ops.resi_chain_batch(org, pred, rec, level, dd, n, bd, (0, (1 << bd) - 1))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rng = np.random.default_rng(3)
    fns = [build(s, rng) for s in SIZES]
    ms = {s: [] for s in SIZES}
    for r in range(reps + 1):
        for s, f in zip(SIZES, fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record()
            torch.cuda.synchronize()
            if r:
                ms[s].append(a.elapsed_time(b))
    for s in SIZES:
        print("%2dx%-2d call (3 launches, event-timed) %7.1f us" % (s, s, 1e3 * sum(ms[s]) / len(ms[s])))


if __name__ == "__main__":
    main()
