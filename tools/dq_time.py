#!/usr/bin/env python3
"""N1 timing alone (the first table of tools/n13_time.py without the host reference): vvcgpu_dequant_tr_inv_batch on one TU per B x B tile of a 4K
picture, dependent quantisation.  For `rocprofv3 --kernel-trace --stats -- python3 tools/dq_time.py`: kernel time against the event time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops  # noqa: E402

rng = np.random.default_rng(2)
W, H, bd = 3840, 2160, 10
for B in (8, 16, 32, 64):
    n = (W // B) * (H // B)
    d = np.zeros(n, ops.DQTR_DESC)
    d["resi_off"] = d["level_off"] = np.arange(n) * B * B
    d["resi_stride"], d["w"], d["h"] = B, B, B
    pair = rng.integers(0, 3, n) if B <= 32 else np.zeros(n, np.int64)
    d["tr_hor"] = np.where(pair == 0, 0, np.where(pair == 1, 2, 1))
    d["tr_ver"] = np.where(pair == 0, 0, np.where(pair == 1, 2, 2))
    d["dep_quant"] = 1
    d["qp"] = rng.integers(22, 38, n)
    lv = (rng.integers(-12, 13, n * B * B) * (rng.random(n * B * B) < 0.35)).astype(np.int32)
    if B == 64:
        lv = lv.reshape(n, 64, 64)
        lv[:, 32:, :] = 0
        lv[:, :, 32:] = 0
        lv = lv.reshape(-1)
    dl, dd = torch.from_numpy(lv).cuda(), ops.struct_to_device(d)
    res = torch.zeros(n * B * B, dtype=torch.int16, device="cuda")
    coef = torch.zeros(n * B * B, dtype=torch.int32, device="cuda")
    for name, co in (("with coeff_out", coef), ("no coeff_out", None)):
        fn = lambda: ops.dequant_tr_inv_batch(dl, res, dd, n, bd, co)
        fn(); torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        byts = n * B * B * (4 + 2)
        print("N1 dequant+T2 %2dx%-2d %-14s: %6d TUs %.3f ms (host side of a call %.3f ms)  %.0f GB/s (%.1f%% of 8 TB/s)" %
              (B, B, name, n, ms, (t1 - t0) / reps * 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80), flush=True)
