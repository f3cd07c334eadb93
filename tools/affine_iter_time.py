#!/usr/bin/env python3
"""loop body of the affine gradient search at 4K: the fused entry (vvcgpu_affine_me_iter_batch) against the chain of the separate entry points it
replaces (sub-block descriptors, MC, subtract, two Sobel planes, equation sums, Hadamard distortion)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops  # noqa: E402

rng = np.random.default_rng(4)
W, H, bd, M = 3840, 2160, 10, 144
ref = torch.from_numpy(np.pad(rng.integers(0, 1024, (H, W)).astype(np.int16), M, mode="edge")).cuda()
org = torch.from_numpy(rng.integers(0, 1024, (H, W)).astype(np.int16)).cuda()
PW = W + 2 * M


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for B in (16, 32, 64, 128):
    xs, ys = np.arange(0, W - B + 1, B), np.arange(0, H - B + 1, B)
    gx, gy = (v.reshape(-1) for v in np.meshgrid(xs, ys))
    n = gx.size
    items = np.zeros(n, ops.AFFINE_ITER)
    nsb = (B // 4) ** 2
    items["pu"]["pos_x"], items["pu"]["pos_y"], items["pu"]["w"], items["pu"]["h"] = gx, gy, B, B
    items["pu"]["six_param"] = rng.integers(0, 2, n)
    items["pu"]["mv"][:, 0] = rng.integers(-64, 65, (n, 3, 2))
    items["pu"]["dst_off"], items["pu"]["dst_stride"], items["pu"]["first_desc"] = np.arange(n) * B * B, B, np.arange(n) * nsb
    items["org_off"], items["org_stride"] = gy * W + gx, W
    di = ops.struct_to_device(items)
    pus = ops.struct_to_device(np.ascontiguousarray(items["pu"]))
    pred = torch.zeros(n * B * B, dtype=torch.int16, device="cuda")
    resi = torch.zeros(n * B * B, dtype=torch.int16, device="cuda")
    gxp = torch.zeros(n * B * B, dtype=torch.int32, device="cuda"); gyp = torch.zeros(n * B * B, dtype=torch.int32, device="cuda")
    o = np.arange(n) * B * B
    gd = np.zeros(n, ops.AFG_DESC); gd["pred_off"], gd["deriv_off"], gd["pred_stride"], gd["deriv_stride"], gd["w"], gd["h"] = o, o, B, B, B, B
    ed = np.zeros(n, ops.AFE_DESC); ed["resi_off"], ed["deriv_off"], ed["deriv_stride"], ed["w"], ed["h"], ed["six_param"] = o, o, B, B, B, items["pu"]["six_param"]
    dd = np.zeros(n, ops.DIST_DESC); dd["org_off"], dd["cur_off"], dd["org_stride"], dd["cur_stride"], dd["w"], dd["h"] = items["org_off"], o, W, B, B, B
    sd = np.zeros(n, ops.PELOP_DESC); sd["src0_off"], sd["src1_off"], sd["dst_off"] = items["org_off"], o, o
    sd["src0_stride"], sd["src1_stride"], sd["dst_stride"], sd["w"], sd["h"] = W, B, B, B, B
    dgd, ded, ddd, dsd = (ops.struct_to_device(a) for a in (gd, ed, dd, sd))
    sub = ops.PelopCfg(0, 0, 0, 0, 0, 1023)

    def chain():
        descs = ops.affine_subblock_descs(pus, n, n * nsb, 0, W, H, (M, M), PW, PW)
        ops.mc_batch(ref, ref, pred, descs, n * nsb, bd, (0, 1023))
        ops.pelop_batch(3, org, pred, resi, dsd, n, sub)
        ops.affine_sobel_batch(0, pred, gxp, dgd, n)
        ops.affine_sobel_batch(1, pred, gyp, dgd, n)
        c = ops.affine_equal_coeff_batch(resi, gxp, gyp, ded, n)
        return c, ops.dist_batch(1, org, pred, ddd, n, bd)

    def fused():
        return ops.affine_me_iter_batch(org, ref, pred, di, n, n * nsb, 1, W, H, (M, M), PW, bd, (0, 1023))

    def pred_chain():
        descs = ops.affine_subblock_descs(pus, n, n * nsb, 0, W, H, (M, M), PW, PW)
        ops.mc_batch(ref, ref, pred, descs, n * nsb, bd, (0, 1023))

    def pred_one():
        ops.affine_pred_batch(ref, None, pred, pus, n, n * nsb, 0, W, H, (M, M), PW, PW, bd, (0, 1023))

    pred_chain(); a = pred.clone(); pred.zero_(); pred_one()
    assert torch.equal(a, pred)
    print("affine prediction   %3dx%-3d: %6d PUs  descriptors + vvcgpu_mc_batch %.3f ms   vvcgpu_affine_pred_batch %.3f ms" % (B, B, n, timed(pred_chain), timed(pred_one)),
          flush=True)
    c0, d0 = chain()
    c1, d1 = fused()
    assert torch.equal(c0.reshape(-1), c1.reshape(-1)) and torch.equal(d0.reshape(-1).to(torch.int64), d1.reshape(-1))
    print("affine ME iteration %3dx%-3d: %6d PUs  chain of 8 launches %.3f ms   fused entry %.3f ms" % (B, B, n, timed(chain), timed(fused)), flush=True)
