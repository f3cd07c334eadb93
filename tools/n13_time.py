#!/usr/bin/env python3
"""N1 / N3 measurement at 3840x2160 10-bit.
N1: vvcgpu_dequant_tr_inv_batch on one TU per 8x8 / 16x16 / 32x32 tile of the picture (35 % non-zero levels, dependent
    quantisation, DCT2 and DST7/DCT8 pairs) -- HIP-event time, algorithmic bytes / time, and the compiled reference's
    DepQuant::dequant + xITrMxN_EMT on one host core on a sample.
N3: vvcgpu_affine_sobel_batch (both planes) + vvcgpu_affine_equal_coeff_batch on one PU per 16x16 / 64x64 tile, reference SIMD
    table slots on one core beside it."""
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


rng = np.random.default_rng(2)
W, H, bd = 3840, 2160, 10
R = ref() if ref_available() else None
for B in (8, 16, 32):
    n = (W // B) * (H // B)
    d = np.zeros(n, ops.DQTR_DESC)
    d["resi_off"] = np.arange(n) * B * B
    d["level_off"] = np.arange(n) * B * B
    d["resi_stride"], d["w"], d["h"] = B, B, B
    pair = rng.integers(0, 3, n)
    d["tr_hor"] = np.where(pair == 0, 0, np.where(pair == 1, 2, 1))
    d["tr_ver"] = np.where(pair == 0, 0, np.where(pair == 1, 2, 2))
    d["dep_quant"] = 1
    d["qp"] = rng.integers(22, 38, n)
    lv = (rng.integers(-12, 13, n * B * B) * (rng.random(n * B * B) < 0.35)).astype(np.int32)
    dl, dd = torch.from_numpy(lv).cuda(), ops.struct_to_device(d)
    res = torch.zeros(n * B * B, dtype=torch.int16, device="cuda")
    coef = torch.zeros(n * B * B, dtype=torch.int32, device="cuda")
    ms = gpu_ms(lambda: ops.dequant_tr_inv_batch(dl, res, dd, n, bd, coef))
    byts = n * B * B * (4 + 2)
    line = "N1 dequant+T2 %2dx%-2d: %6d TUs %.3f ms  %.1f M TU/s  %.0f GB/s (%.1f%% of 8 TB/s)" % (B, B, n, ms, n / ms / 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80)
    if R is not None:
        k = 3000
        out = np.zeros(B * B, np.int32); r16 = np.zeros(B * B, np.int16)
        t = time.perf_counter()
        for i in range(k):
            lvl = lv[i * B * B:(i + 1) * B * B]
            R.vtmref_dequant(1, bd, int(d["qp"][i]), 0, p(lvl), p(out), B, B)
            R.vtmref_inv_tr2d(bd, p(out), p(r16), B, B, B, int(d["tr_hor"][i]), int(d["tr_ver"][i]))
        dt = time.perf_counter() - t
        line += "  | reference DepQuant::dequant + xITrMxN_EMT 1 core %.2f us/TU -> x%.0f" % (dt / k * 1e6, (n / ms / 1e3) / (k / dt / 1e6))
    print(line)


# ---- N1: dependent-quantisation trellis, rate tables and coefficient statistics from the committed golden fixture
g = np.load(os.path.join(ROOT, "tests", "golden", "depquant.npz"))
rates = np.ascontiguousarray(g["rates"][:4]).view(ops.DQ_RATES)
if R is not None:
    R.vtmref_depquant.restype = C.c_uint32
for B in (8, 16, 32):
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
    nzf = float((level != 0).float().mean().cpu())
    line = "N1 DepQuant trellis %2dx%-2d: %6d TUs %.3f ms  %.2f M TU/s  (%.0f%% non-zero levels)" % (B, B, n, ms, n / ms / 1e3, 100 * nzf)
    if R is not None:
        k = 1500
        lv = np.zeros(B * B, np.int32)
        t = time.perf_counter()
        for i in range(k):
            R.vtmref_depquant(p(coef[i * B * B:(i + 1) * B * B]), p(lv), B, B, 0, bd, 44, C.c_double(60.0), 32, 0, None)
        dt = time.perf_counter() - t
        line += "  | reference DepQuant::quant 1 core %.2f us/TU -> x%.0f" % (dt / k * 1e6, (n / ms / 1e3) / (k / dt / 1e6))
    print(line)

# ---- N1: rate-distortion optimised quantiser (dependent quantisation off), rate tables from the committed golden fixture
gq = np.load(os.path.join(ROOT, "tests", "golden", "rdoq.npz"))
rq_rates = np.ascontiguousarray(gq["rates"][:4]).view(ops.RDOQ_RATES)
if R is not None:
    R.vtmref_rdoq.restype = C.c_uint32
for B in (4, 8, 16, 32):
    n = (W // B) * (H // B) if B > 4 else (W // 8) * (H // 8)
    yy, xx = np.mgrid[0:B, 0:B]
    decay = np.exp(-(xx / B * 3 + yy / B * 3)).reshape(-1)
    coef = (rng.normal(0, 1500, (n, B * B)) * decay).astype(np.int32).reshape(-1)
    d = np.zeros(n, ops.RDOQ_DESC)
    d["coeff_off"] = d["level_off"] = np.arange(n) * B * B
    d["lambda"], d["qp"], d["rates_idx"], d["w"], d["h"], d["luma"], d["sign_hiding"] = 60.0, 44, rng.integers(0, 4, n), B, B, 1, 1
    dc, dd, dr = torch.from_numpy(coef).cuda(), ops.struct_to_device(d), ops.struct_to_device(rq_rates)
    level = torch.zeros(n * B * B, dtype=torch.int32, device="cuda")
    ms = gpu_ms(lambda: ops.rdoq_batch(dc, level, dd, n, dr, n * B * B, bd), reps=3)
    nzf = float((level != 0).float().mean().cpu())
    line = "N1 RDOQ + sign hiding %2dx%-2d: %6d TUs %.3f ms  %.2f M TU/s  (%.0f%% non-zero levels)" % (B, B, n, ms, n / ms / 1e3, 100 * nzf)
    if R is not None:
        k = 1500
        lv = np.zeros(B * B, np.int32)
        t = time.perf_counter()
        for i in range(k):
            R.vtmref_rdoq(p(coef[i * B * B:(i + 1) * B * B]), p(lv), B, B, 0, bd, 44, C.c_double(60.0), 32, 0, 0, 1, 0, None)
        dt = time.perf_counter() - t
        line += "  | reference QuantRDOQ::quant 1 core %.2f us/TU -> x%.0f" % (dt / k * 1e6, (n / ms / 1e3) / (k / dt / 1e6))
    print(line)

pred = rng.integers(0, 1024, (H, W)).astype(np.int16)
resi = rng.integers(-255, 256, H * W).astype(np.int16)
for B in (16, 64):
    n = (W // B) * (H // B)
    bx, by = np.meshgrid(np.arange(W // B), np.arange(H // B))
    g = np.zeros(n, ops.AFG_DESC); e = np.zeros(n, ops.AFE_DESC)
    g["pred_off"] = (by.ravel() * B) * W + bx.ravel() * B
    g["deriv_off"] = np.arange(n) * B * B
    g["pred_stride"], g["deriv_stride"], g["w"], g["h"] = W, B, B, B
    e["resi_off"] = e["deriv_off"] = np.arange(n) * B * B
    e["deriv_stride"], e["w"], e["h"] = B, B, B
    e["six_param"] = rng.integers(0, 2, n)
    dp, dr = torch.from_numpy(pred).cuda(), torch.from_numpy(resi).cuda()
    gx = torch.zeros(n * B * B, dtype=torch.int32, device="cuda"); gy = torch.zeros_like(gx)
    dg, de = ops.struct_to_device(g), ops.struct_to_device(e)

    def step():
        ops.affine_sobel_batch(0, dp, gx, dg, n)
        ops.affine_sobel_batch(1, dp, gy, dg, n)
        return ops.affine_equal_coeff_batch(dr, gx, gy, de, n)
    ms = gpu_ms(step)
    byts = n * B * B * (2 * 2 + 4 * 2 + 4 * 2 + 2)        # two Sobel passes (2 B in, 4 B out each), equal coeff (2 x 4 B + 2 B in)
    line = "N3 affine sobel x2 + equal coeff %2dx%-2d: %6d PUs %.3f ms  %.1f M PU/s  %.0f GB/s (%.1f%% of 8 TB/s)" % (B, B, n, ms, n / ms / 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80)
    if R is not None:
        k = 2000 if B == 16 else 300
        hx = np.zeros(B * B, np.int32); hy = np.zeros(B * B, np.int32); o = np.zeros(49, np.int64)
        blk = np.ascontiguousarray(pred[:B, :B]); rb = np.ascontiguousarray(resi[:B * B])
        t = time.perf_counter()
        for i in range(k):
            R.vtmref_affine_sobel(1, 0, p(blk), B, p(hx), B, B, B)
            R.vtmref_affine_sobel(1, 1, p(blk), B, p(hy), B, B, B)
            R.vtmref_affine_equal_coeff(1, p(rb), p(hx), p(hy), B, B, B, int(e["six_param"][i]), p(o))
        dt = time.perf_counter() - t
        line += "  | reference SIMD slots 1 core %.2f us/PU -> x%.0f" % (dt / k * 1e6, (n / ms / 1e3) / (k / dt / 1e6))
    print(line)
