#!/usr/bin/env python3
"""N2 measurement: vvcgpu_tz_search_batch on every 16x16 / 32x32 / 64x64 PU of a 3840x2160 picture (start vectors spread
around a true global displacement) -- kernel time by HIP events, PUs/s, and the compiled reference's own xTZSearch
(oracle/_ref, 1 host core) on a bounded sample of the same PUs.  usage: python3 tools/tz_time.py [--cpu-sample N]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cases  # noqa: E402
from oraclelib import oracle, ref, ref_available, p  # noqa: E402
from vvcsoftware_vtm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cpu-sample", type=int, default=1500)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--team", type=int, default=0, help="cfg.wg_per_pu")
ap.add_argument("--range", type=int, default=96, help="cfg.search_range")
ap.add_argument("--split", action="store_true", help="cfg.uniform_pu = h << 16 | w: uniform batch, the raster stage as its own launch")
ap.add_argument("--near", action="store_true", help="start vectors within +-2 samples of the true displacement (the raster stage is rarely entered)")
a = ap.parse_args()
rng = np.random.default_rng(11)
W, H, M = 3840, 2160, 160
org, ref_ = cases.tz_planes(rng, W, H, M, 10, motion=(11, -6))
dorg, dref = torch.from_numpy(org).cuda(), torch.from_numpy(ref_).cuda()
cfg = cases.tz_cfg(W, H, M, 30.0, search_range=a.range, wg_per_pu=a.team)
for size in (16, 32, 64):
    xs, ys = np.meshgrid(np.arange(0, W - size + 1, size), np.arange(0, H - size + 1, size))
    n = xs.size
    pus = np.zeros(n, cases.TZ_PU)
    pus["org_x"], pus["org_y"] = xs.ravel(), ys.ravel()
    pus["ref_x"], pus["ref_y"] = pus["org_x"] + M, pus["org_y"] + M
    pus["pos_x"], pus["pos_y"] = pus["org_x"], pus["org_y"]
    pus["w"], pus["h"], pus["sub_shift"] = size, size, 1
    if a.near:
        pus["start_x"], pus["start_y"] = 44 + rng.integers(-8, 9, n), -24 + rng.integers(-8, 9, n)
    else:
        pus["start_x"], pus["start_y"] = rng.integers(-40, 41, n), rng.integers(-40, 41, n)
    pus["pred_hor"], pus["pred_ver"] = pus["start_x"], pus["start_y"]
    dp = ops.struct_to_device(pus)
    cfg["reserved"] = ((size << 16) | size) if a.split else 0       # vvcgpu_tz_cfg.uniform_pu
    best = ops.tz_search_batch(dorg, dref, dp, n, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        best = ops.tz_search_batch(dorg, dref, dp, n, cfg)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    got = best.cpu().numpy().view(cases.BEST)
    line = ("split " if a.split else "") + "%dx%d: %d PUs  gpu %.3f ms  %.2f M PU/s  found %.0f%%" % (size, size, n, ms, n / ms / 1e3, 100 * np.mean((got["x"] == 11) & (got["y"] == -6)))
    k = min(a.cpu_sample, n)
    if k == 0:
        print(line)
        continue
    pick = np.ascontiguousarray(pus[rng.choice(n, k, replace=False)])
    res = np.zeros(k, cases.BEST)
    oracle().orc_tz_search(p(org), W, p(ref_), W + 2 * M, p(pick), k, p(cfg), p(res))
    st = np.zeros(3, np.uint64)
    oracle().orc_tz_stats(p(st))
    line += "  [per PU: %.0f probes in %.1f rounds, %.0f of them raster]" % (st[0] / k, st[1] / k, st[2] / k)
    if ref_available():
        t = time.perf_counter()
        ref().vtmref_tz_search(p(org), W, p(ref_), W + 2 * M, p(pick), k, p(cfg), 10, p(res))
        dt = time.perf_counter() - t
        line += "  | reference xTZSearch 1 core: %.1f us/PU  %.3f M PU/s  -> x%.0f" % (dt / k * 1e6, k / dt / 1e6, (n / ms / 1e3) / (k / dt / 1e6))
    print(line)
