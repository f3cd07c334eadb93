#!/usr/bin/env python3
"""Dependent-quantisation trellis alone (vvcgpu_depquant_batch), event-timed: (1) the TU list of the 4K `with_depquant` leg of bench.py on the
coefficients its own forward transforms produce, (2) all B x B TUs of a 4K picture for B = 8, 16, 32 (the table of tools/n13_time.py), (3) the
real call mix of the committed encoder trace.  Prints the md5 of the 4K leg's level buffer so that variants can be compared bit by bit.
usage: python tools/dq_leg_time.py [leg,sizes,mix]"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import ops, shape_mix  # noqa: E402
from vvcsoftware_vtm_amd.workload import Workload  # noqa: E402

what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["leg", "sizes", "mix"]
bd = 10


def gpu_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if "leg" in what:
    wl = Workload(3840, 2160, bd, seed=20261003, qp=32, depquant=True)
    st, _ = wl.run_gpu(None, None, overlap=False)
    st, out = wl.run_gpu(st, None, overlap=False)
    torch.cuda.synchronize()
    fn = lambda: ops.depquant_batch(st["coef"], st["level"], st["dq"], wl.tr.size, st["dq_rates"], wl.n_coef, bd)
    ms = gpu_ms(fn)
    print("4K leg: %d TUs %s, %.3f ms   levels md5 %s  abs sum %d" % (wl.tr.size, wl.tu_runs, ms, hashlib.md5(st["level"].cpu().numpy().tobytes()).hexdigest()[:12],
                                                                   int(fn().to(torch.int64).sum().item())), flush=True)
    del wl, st, out

if "sizes" in what:
    rng = np.random.default_rng(2)
    g = np.load(os.path.join(ROOT, "tests", "golden", "depquant.npz"))
    rates = np.ascontiguousarray(g["rates"][:4]).view(ops.DQ_RATES)
    W, H = 3840, 2160
    for B in (4, 8, 16, 32):
        n = (W // B) * (H // B)
        yy, xx = np.mgrid[0:B, 0:B]
        decay = np.exp(-(xx / B * 3 + yy / B * 3)).reshape(-1)
        coef = (rng.normal(0, 1500, (n, B * B)) * decay).astype(np.int32).reshape(-1)
        d = np.zeros(n, ops.DEPQUANT_DESC)
        d["coeff_off"] = d["level_off"] = np.arange(n) * B * B
        d["lambda"], d["qp"], d["rates_idx"], d["w"], d["h"], d["luma"] = 60.0, 44, rng.integers(0, 4, n), B, B, 1
        dc, dd, dr = torch.from_numpy(coef).cuda(), ops.struct_to_device(d), ops.struct_to_device(rates)
        level = torch.zeros(n * B * B, dtype=torch.int32, device="cuda")
        ms = gpu_ms(lambda: ops.depquant_batch(dc, level, dd, n, dr, n * B * B, bd), reps=3)
        print("all %2dx%-2d TUs of a 4K picture: %6d TUs %.3f ms   md5 %s" % (B, B, n, ms, hashlib.md5(level.cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)

if "mix" in what:
    res = shape_mix.run(1 << 21, only=["depquant_batch"])
    for name, r in res.items():
        print("%s real mix: %d calls, %.3f ms; 16x16: %.3f ms; ratio %.3f" % (name, r["calls"], r["real_ms"], r["square_ms"], r["ratio"]), flush=True)
