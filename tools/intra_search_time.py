#!/usr/bin/env python3
"""N4 in context: the SATD pre-selection of the encoder's intra mode search (IntraSearch::estIntraPredLumaQT, EncoderLib/IntraSearch.cpp:
400-500: every luma mode is predicted and ranked by Hadamard SATD against the original) composed from the C-ABI entry points --
vvcgpu_intra_fill_refs_batch (reference samples of every block), vvcgpu_intra_pred_batch (67 modes per block, all sharing the block's
reference samples) and vvcgpu_dist_batch(HAD) -- for every 16x16 block of a 3840x2160 picture.  Prints HIP-event times and checks a
sample of (block, mode) costs against the oracle."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oraclelib import oracle, p  # noqa: E402
from vvcsoftware_vtm_amd import ops  # noqa: E402

rng = np.random.default_rng(5)
W, H, B, bd = 3840, 2160, 16, 10
org = rng.integers(0, 1024, (H, W)).astype(np.int16)
rec = np.clip(org.astype(np.int32) + rng.integers(-12, 13, (H, W)), 0, 1023).astype(np.int16)     # "reconstruction" the references come from
T, L = ops.intra_ref_lengths(B, B)
unit = 4
nU = (T + unit - 1) // unit + (L + unit - 1) // unit + 1
bx, by = np.meshgrid(np.arange(1, W // B - 2), np.arange(1, H // B - 2))        # interior blocks: all neighbours exist
nb = bx.size
fd = np.zeros(nb, ops.INTRA_FILL_DESC)
fd["rec_off"] = (by.ravel() * B) * W + bx.ravel() * B
fd["flags_off"] = 0
fd["ref_off"] = np.arange(nb) * (T + L + 1)
fd["rec_stride"], fd["w"], fd["h"], fd["unit_w"], fd["unit_h"] = W, B, B, unit, unit
flags = np.ones(nU, np.uint8)
pd = np.zeros(nb * 67, ops.INTRA_DESC)
pd["ref_off"] = np.repeat(fd["ref_off"], 67)
pd["dst_off"] = np.arange(nb * 67) * B * B
pd["dst_stride"], pd["w"], pd["h"] = B, B, B
pd["mode"] = np.tile(np.arange(67), nb)
pd["filter_refs"] = (pd["mode"] % 2 == 0)
dd = np.zeros(nb * 67, ops.DIST_DESC)
dd["org_off"] = np.repeat(fd["rec_off"], 67)
dd["cur_off"] = pd["dst_off"]
dd["org_stride"], dd["cur_stride"], dd["w"], dd["h"] = W, B, B, B

d_org, d_rec, d_flags = torch.from_numpy(org).cuda(), torch.from_numpy(rec).cuda(), torch.from_numpy(flags).cuda()
refs = torch.zeros(nb * (T + L + 1), dtype=torch.int16, device="cuda")
pred = torch.zeros(nb * 67 * B * B, dtype=torch.int16, device="cuda")
d_fd, d_pd, d_dd = ops.struct_to_device(fd), ops.struct_to_device(pd), ops.struct_to_device(dd)


def step():
    ops.intra_fill_refs_batch(d_rec, d_flags, refs, d_fd, nb, bd)
    ops.intra_pred_batch(refs, pred, d_pd, nb * 67)
    return ops.dist_batch(ops.HAD, d_org, pred, d_dd, nb * 67, bd)


cost = step(); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record(); ops.intra_fill_refs_batch(d_rec, d_flags, refs, d_fd, nb, bd)
ev[1].record(); ops.intra_pred_batch(refs, pred, d_pd, nb * 67)
ev[2].record(); cost = ops.dist_batch(ops.HAD, d_org, pred, d_dd, nb * 67, bd)
ev[3].record(); torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
print("intra mode pre-selection, %d blocks x 67 modes: fill %.3f ms, predict %.3f ms (%.0f M predictions/s), SATD %.3f ms; total %.3f ms" %
      (nb, t[0], t[1], nb * 67 / t[1] / 1e3, t[2], sum(t)))
# the fused entry (round 3): predict -> SATD in one launch, the prediction stays in LDS
sd = np.zeros(nb * 67, ops.INTRA_SATD_DESC)
sd["ref_off"], sd["org_off"], sd["org_stride"], sd["w"], sd["h"] = pd["ref_off"], dd["org_off"], W, B, B
sd["mode"], sd["filter_refs"] = pd["mode"], pd["filter_refs"]
d_sd = ops.struct_to_device(sd)
cost2 = ops.intra_satd_batch(refs, d_org, d_sd, nb * 67); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    cost2 = ops.intra_satd_batch(refs, d_org, d_sd, nb * 67)
e1.record(); torch.cuda.synchronize()
print("vvcgpu_intra_satd_batch (predict + SATD fused, one launch): %.3f ms against predict + SATD %.3f ms; costs %s" %
      (e0.elapsed_time(e1) / 5, t[1] + t[2], "identical" if torch.equal(cost, cost2) else "DIFFER"))
best = cost.view(nb, 67).argmin(1).cpu().numpy()
print("best-mode histogram (planar, DC, angular):", int((best == 0).sum()), int((best == 1).sum()), int((best > 1).sum()))
# oracle check of a sample
O = oracle()
O.orc_satd.restype = C.c_uint64
got = cost.cpu().numpy().view(np.uint64).reshape(nb, 67)
for i in rng.choice(nb, 20, replace=False):
    r = np.zeros(T + L + 1, np.int16)
    O.orc_intra_fill_refs(C.c_void_p(rec.ctypes.data + int(fd["rec_off"][i]) * 2), W, p(flags), p(r), B, B, unit, unit, bd)
    for m in rng.choice(67, 6, replace=False):
        src = r
        if m % 2 == 0:
            src = np.zeros_like(r); O.orc_intra_filter_refs(p(r), p(src), B, B)
        pb = np.zeros((B, B), np.int16)
        O.orc_intra_pred(p(src), p(pb), B, B, B, int(m), 0, 1023)
        want = O.orc_satd(C.c_void_p(org.ctypes.data + int(fd["rec_off"][i]) * 2), W, p(pb), B, B, B)
        assert int(got[i, m]) == int(want), (i, m, got[i, m], want)
print("oracle sample ok")
