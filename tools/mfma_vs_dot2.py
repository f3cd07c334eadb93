#!/usr/bin/env python3
"""the 16 / 32 / 64-point transform stages on the matrix cores against the v_dot2 / integer form, measured (GPU box).

Every luma sample of a 3840x2160 picture is covered by square TUs of ONE size N; timed per launch with HIP events:
  dot2 : vvcgpu_tr_fwd_batch + vvcgpu_tr_inv_batch  (transform.hip: both 1-D stages in one wave, int16 matrices in LDS, v_dot2_i32_i16 /
         integer multiply-adds) -- the forward and the inverse transform only
  mfma : vvcgpu_resi_chain_batch (resichain.hip: v_mfma_f32_16x16x32_f16 with 8-bit limb splitting) -- subtract, forward transform, Quant::quant
         with sign hiding, Quant::dequant, inverse transform and reconstruction, i.e. MORE work than the dot2 leg
Second table (round 3): the STANDALONE entries with their own matrix-core kernels (transform.hip tr_fwd_mfma_kernel / tr_inv_mfma_kernel, stages
of mfma_tr.h) against the dot2 kernels (VVCGPU_NO_MFMA=1, read per call), squares and rectangles; the coefficients and the residual of the two
forms are compared before timing.
Results of the transforms are covered by tests/test_gpu_resichain.py / test_gpu_transform.py (bit-exact against the oracle).

usage: python tools/mfma_vs_dot2.py > profiles/rNN_mfma_vs_dot2.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops  # noqa: E402
from vvcsoftware_vtm_amd.workload import TR_DESC  # noqa: E402

W, H, BD = 3840, 2112, 10          # 2112 = 33 x 64 rows: whole TUs of every size


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    rng = np.random.default_rng(7)
    org = torch.from_numpy(rng.integers(0, 1024, (H, W), dtype=np.int16)).cuda()
    pred = torch.from_numpy(np.clip(org.cpu().numpy() + rng.integers(-40, 41, (H, W)), 0, 1023).astype(np.int16)).cuda()
    resi = (org - pred).contiguous()
    print("transform stages, dot2 / integer form against the matrix-core form; %dx%d luma, 10 bit, every sample in a square TU of size N" % (W, H))
    print("%4s %8s | %12s %12s %12s | %12s | %s" % ("N", "TUs", "fwd dot2 ms", "inv dot2 ms", "fwd+inv ms", "chain mfma ms", "chain / (fwd + inv)"))
    for n in (64, 32, 16, 8, 4):
        ys, xs = np.meshgrid(np.arange(0, H, n), np.arange(0, W, n), indexing="ij")
        k = ys.size
        tr = np.zeros(k, TR_DESC)
        tr["resi_off"] = (ys * W + xs).ravel()
        tr["coeff_off"] = np.arange(k, dtype=np.int64) * n * n
        tr["resi_stride"], tr["w"], tr["h"] = W, n, n
        rc = np.zeros(k, ops.RC_DESC)
        rc["org_off"] = rc["pred_off"] = rc["rec_off"] = tr["resi_off"]
        rc["level_off"] = tr["coeff_off"]
        rc["org_stride"] = rc["pred_stride"] = rc["rec_stride"] = W
        rc["w"] = rc["h"] = n
        rc["qp"], rc["sign_hiding"] = 32 + 12, 1
        dtr, drc = ops.struct_to_device(tr), ops.struct_to_device(rc)
        coef = torch.zeros(H * W, dtype=torch.int32, device="cuda")
        back = torch.zeros((H, W), dtype=torch.int16, device="cuda")
        rec = torch.zeros((H, W), dtype=torch.int16, device="cuda")
        level = torch.zeros(H * W, dtype=torch.int32, device="cuda")
        tf = timed(lambda: ops.tr_fwd_batch(resi, coef, dtr, k, BD))
        ti = timed(lambda: ops.tr_inv_batch(coef, back, dtr, k, BD))
        tc = timed(lambda: ops.resi_chain_batch(org, pred, rec, level, drc, k, BD, (0, 1023)))
        print("%4d %8d | %12.4f %12.4f %12.4f | %12.4f | %.2f%s" % (n, k, tf, ti, tf + ti, tc, tc / (tf + ti), "   (lane groups, no MFMA)" if n <= 8 else ""))


def standalone():
    rng = np.random.default_rng(9)
    resi = torch.from_numpy(rng.integers(-300, 301, (H, W), dtype=np.int16)).cuda()
    print()
    print("standalone vvcgpu_tr_fwd_batch / vvcgpu_tr_inv_batch: matrix-core kernels against the dot2 kernels (same entry point, VVCGPU_NO_MFMA=1)")
    print("%7s %8s | %10s %10s %6s | %10s %10s %6s | %s" % ("W x H", "TUs", "fwd dot2", "fwd mfma", "x", "inv dot2", "inv mfma", "x", "results"))
    for (w, h) in ((64, 64), (32, 32), (64, 32), (32, 64), (64, 16), (16, 64), (32, 16), (16, 32)):
        ys, xs = np.meshgrid(np.arange(0, H - h + 1, h), np.arange(0, W, w), indexing="ij")
        k = ys.size
        tr = np.zeros(k, TR_DESC)
        tr["resi_off"] = (ys * W + xs).ravel()
        tr["coeff_off"] = np.arange(k, dtype=np.int64) * w * h
        tr["resi_stride"], tr["w"], tr["h"] = W, w, h
        dtr = ops.struct_to_device(tr)
        out = {}
        t = {}
        for form in ("dot2", "mfma"):
            if form == "dot2":
                os.environ["VVCGPU_NO_MFMA"] = "1"
            else:
                os.environ.pop("VVCGPU_NO_MFMA", None)
            coef = torch.full((H * W,), 7, dtype=torch.int32, device="cuda")
            back = torch.full((H, W), 7, dtype=torch.int16, device="cuda")
            t[form] = (timed(lambda: ops.tr_fwd_batch(resi, coef, dtr, k, BD)), timed(lambda: ops.tr_inv_batch(coef, back, dtr, k, BD)))
            out[form] = (coef.cpu().numpy(), back.cpu().numpy())
        same = np.array_equal(out["dot2"][0], out["mfma"][0]) and np.array_equal(out["dot2"][1], out["mfma"][1])
        print("%3dx%-3d %8d | %10.4f %10.4f %6.2f | %10.4f %10.4f %6.2f | %s" % (w, h, k, t["dot2"][0], t["mfma"][0], t["dot2"][0] / t["mfma"][0],
                                                                               t["dot2"][1], t["mfma"][1], t["dot2"][1] / t["mfma"][1],
                                                                               "identical" if same else "DIFFER"))


if __name__ == "__main__":
    main()
    standalone()
