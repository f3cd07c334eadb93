#!/usr/bin/env python3
"""N4 measurement at 3840x2160 10-bit: intra prediction (one block per 16x16 / 32x32 / 8x8 tile, random modes, half of them with
the reference filter), border extension (luma, margin 144) and CRC / checksum of the luma plane -- HIP-event time per launch,
algorithmic bytes / time against 8 TB/s, and the compiled reference on one host core beside it."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oraclelib import ref, ref_available, p  # noqa: E402
from vvcsoftware_vtm_amd import ops  # noqa: E402


def gpu_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rng = np.random.default_rng(1)
W, H = 3840, 2160
R = ref() if ref_available() else None
for B in (8, 16, 32):
    T, L = ops.intra_ref_lengths(B, B)
    nb = (W // B) * (H // B)
    refs = rng.integers(0, 1024, nb * (T + L + 1)).astype(np.int16)
    d = np.zeros(nb, ops.INTRA_DESC)
    bx, by = np.meshgrid(np.arange(W // B), np.arange(H // B))
    d["ref_off"] = np.arange(nb) * (T + L + 1)
    d["dst_off"] = (by.ravel() * B) * W + bx.ravel() * B
    d["dst_stride"], d["w"], d["h"] = W, B, B
    d["mode"] = rng.integers(0, 67, nb)
    d["filter_refs"] = rng.integers(0, 2, nb)
    out = torch.zeros((H, W), dtype=torch.int16, device="cuda")
    dr, dd = torch.from_numpy(refs).cuda(), ops.struct_to_device(d)
    ms = gpu_ms(lambda: ops.intra_pred_batch(dr, out, dd, nb))
    byts = nb * ((T + L + 1) * 2 + B * B * 2 + 32)
    line = "intra %2dx%-2d: %6d blocks %.3f ms  %.1f M blocks/s  %.0f GB/s (%.1f%% of 8 TB/s)" % (B, B, nb, ms, nb / ms / 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80)
    if R is not None:
        k = 4000
        pred = np.zeros((B, B), np.int16)
        t = time.perf_counter()
        for i in range(k):
            R.vtmref_intra_pred(p(refs[i * (T + L + 1):]), p(pred), B, B, B, int(d["mode"][i]), 0, 1023, 10, int(d["filter_refs"][i]), None)
        dt = time.perf_counter() - t
        line += "  | reference predIntraAng 1 core %.2f us/block -> x%.0f" % (dt / k * 1e6, (nb / ms / 1e3) / (k / dt / 1e6))
    print(line)

M = 144
pad = torch.from_numpy(rng.integers(0, 1024, (H + 2 * M, W + 2 * M)).astype(np.int16)).cuda()
ms = gpu_ms(lambda: ops.extend_border(pad, M, M))
byts = ((H + 2 * M) * (W + 2 * M) - H * W) * 2 * 2
print("extend_border luma m=144: %.3f ms  %.0f GB/s (margin read+written)" % (ms, byts / ms / 1e6))
plane = pad[M:M + H, M:M + W]
for name, meth in (("CRC", ops.HASH_CRC), ("checksum", ops.HASH_CHECKSUM)):
    ms = gpu_ms(lambda: ops.picture_hash(meth, plane, 10))
    line = "%s luma: %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (name, ms, W * H * 2 / ms / 1e6, W * H * 2 / ms / 1e6 / 80)
    if R is not None:
        host = np.ascontiguousarray(plane.cpu().numpy())
        t = time.perf_counter()
        getattr(R, "vtmref_" + name.lower())(10, p(host), W, W, H)
        line += "  | reference 1 core %.1f ms" % ((time.perf_counter() - t) * 1e3)
    print(line)
