#!/usr/bin/env python3
"""Runs the dependent-quantisation trellis kernel on every BxB TU of a 3840x2160 picture a few times (for rocprofv3 PMC passes).
usage: python3 tools/dq_run.py [B] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vvcsoftware_vtm_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.default_rng(2)
W, H = 3840, 2160
g = np.load(os.path.join(ROOT, "tests", "golden", "depquant.npz"))
rates = np.ascontiguousarray(g["rates"][:4]).view(ops.DQ_RATES)
n = (W // B) * (H // B)
yy, xx = np.mgrid[0:B, 0:B]
decay = np.exp(-(xx / B * 3 + yy / B * 3)).reshape(-1)
coef = (rng.normal(0, 1500, (n, B * B)) * decay).astype(np.int32).reshape(-1)
d = np.zeros(n, ops.DEPQUANT_DESC)
d["coeff_off"] = d["level_off"] = np.arange(n) * B * B
d["lambda"], d["qp"], d["rates_idx"], d["w"], d["h"], d["luma"] = 60.0, 44, rng.integers(0, 4, n), B, B, 1
dc, dd, dr = torch.from_numpy(coef).cuda(), ops.struct_to_device(d), ops.struct_to_device(rates)
level = torch.zeros(n * B * B, dtype=torch.int32, device="cuda")
for _ in range(reps):
    ops.depquant_batch(dc, level, dd, n, dr, n * B * B, 10)
torch.cuda.synchronize()
print("done", n)
