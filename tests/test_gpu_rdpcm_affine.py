"""T3 residual DPCM and I3 affine sub-block vectors on the GPU against the golden vectors of the compiled reference (tests/golden/rdpcm.npz:
TrQuant::applyForwardRDPCM / invRdpcmNxN; tests/golden/affine_mv.npz: InterPrediction::xPredAffineBlk) and against the oracle on fresh inputs."""
import os

import numpy as np
import pytest
import torch

from oraclelib import oracle, p

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_rdpcm_golden():
    from vvcsoftware_vtm_amd import ops
    g = np.load(os.path.join(G, "rdpcm.npz"))
    rows = g["rows"]
    for bd in (8, 10):
        sel = np.nonzero(rows[:, 0] == bd)[0]
        offs = np.concatenate([[0], np.cumsum(rows[:, 1] * rows[:, 2])])
        d = np.zeros(sel.size, ops.RDPCM_DESC)
        for k, i in enumerate(sel):
            _, w, h, mode, lossless, rot, intra, qp = rows[i]
            d[k] = (offs[i], offs[i], w, w, h, mode, lossless, rot, intra, qp, 0, 0)
        coef = torch.zeros(g["coef"].size, dtype=torch.int32, device="cuda")
        s = ops.rdpcm_fwd_batch(dev(g["resi"]), coef, ops.struct_to_device(d), sel.size, bd).cpu().numpy().view(np.uint32)
        c = coef.cpu().numpy()
        for k, i in enumerate(sel):
            assert np.array_equal(c[offs[i]:offs[i + 1]], g["coef"][offs[i]:offs[i + 1]]), rows[i]
        assert np.array_equal(s, g["abs_sum"][sel])
        # inverse, in place (rotation does not apply to it)
        keep = [k for k, i in enumerate(sel) if rows[i][5] == 0]
        r = dev(g["inv_in"])
        ops.rdpcm_inv_batch(r, ops.struct_to_device(d[keep]), len(keep))
        r = r.cpu().numpy()
        for k in keep:
            i = sel[k]
            assert np.array_equal(r[offs[i]:offs[i + 1]], g["inv_out"][offs[i]:offs[i + 1]]), rows[i]


def test_rdpcm_strided_blocks_vs_oracle():
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(8)
    W, H, bd = 192, 96, 10
    plane = rng.integers(-700, 701, (H, W)).astype(np.int16)
    rows, off = [], 0
    for y in range(0, H, 32):
        for x in range(0, W, 32):
            w, h = int(rng.choice([4, 8, 16, 32])), int(rng.choice([4, 8, 16, 32]))
            rows.append((y * W + x, off, W, w, h, int(rng.integers(0, 3)), int(rng.integers(0, 2)), 0, int(rng.integers(0, 2)), int(rng.integers(20, 50)), 0, 0))
            off += w * h
    d = np.array(rows, dtype=ops.RDPCM_DESC)
    want = np.zeros(off, np.int32); ws = np.zeros(d.size, np.uint32)
    oracle().orc_rdpcm_fwd_batch(p(plane), p(want), p(d), d.size, bd, p(ws))
    coef = torch.zeros(off, dtype=torch.int32, device="cuda")
    s = ops.rdpcm_fwd_batch(dev(plane), coef, ops.struct_to_device(d), d.size, bd)
    assert np.array_equal(coef.cpu().numpy(), want) and np.array_equal(s.cpu().numpy().view(np.uint32), ws)
    inv = plane.copy()
    oracle().orc_rdpcm_inv_batch(p(inv), p(d), d.size)
    t = dev(plane)
    ops.rdpcm_inv_batch(t, ops.struct_to_device(d), d.size)
    assert np.array_equal(t.cpu().numpy(), inv)


def test_affine_subblock_vectors_golden():
    """the device-derived sub-block descriptors, interpolated by vvcgpu_mc_batch, reproduce the reference's own xPredAffineBlk prediction"""
    from vvcsoftware_vtm_amd import ops
    g = np.load(os.path.join(G, "affine_mv.npz"))
    W, H, bd, M = 256, 128, 10, 144
    planes = [g["Y"], g["Cb"], g["Cr"]]
    pads = [np.ascontiguousarray(np.pad(pl, M >> (1 if c else 0), mode="edge")) for c, pl in enumerate(planes)]
    dpads = [dev(a) for a in pads]
    rows = g["rows"]
    pos = 0
    for comp in range(3):
        c = 1 if comp else 0
        pus = np.zeros(rows.shape[0], ops.AFFINE_PU)
        first, dst_off = 0, 0
        for i, r in enumerate(rows):
            px, py, w, h, six = r[:5]
            mv = np.zeros((2, 3, 2), np.int32); mv[0] = r[5:11].reshape(3, 2)
            pus[i] = (px, py, w, h, six, 0, mv, dst_off, w >> c, first)
            first += (w // 4) * (h // 4)
            dst_off += (w >> c) * (h >> c)
        descs = ops.affine_subblock_descs(ops.struct_to_device(pus), rows.shape[0], first, c, W, H, (M >> c, M >> c), pads[comp].shape[1], pads[comp].shape[1])
        # against the oracle's derivation, field by field
        want_d = np.zeros(first, ops.MC_DESC)
        oracle().orc_affine_subblock_descs(p(pus), rows.shape[0], c, W, H, 128, 128, M >> c, M >> c, pads[comp].shape[1], pads[comp].shape[1], p(want_d))
        assert np.array_equal(descs.cpu().numpy().view(ops.MC_DESC), want_d)
        dst = torch.zeros(dst_off, dtype=torch.int16, device="cuda")
        ops.mc_batch(dpads[comp], dpads[comp], dst, descs, int(first), bd, (0, 1023))
        got = dst.cpu().numpy()
        # golden predictions are stored PU by PU, component by component
        o = 0
        for i, r in enumerate(rows):
            w, h = int(r[2]), int(r[3])
            base = sum(((int(q[2]) * int(q[3])) + 2 * ((int(q[2]) >> 1) * (int(q[3]) >> 1))) for q in rows[:i])
            if comp == 1:
                base += w * h
            elif comp == 2:
                base += w * h + (w >> 1) * (h >> 1)
            n = (w >> c) * (h >> c)
            assert np.array_equal(got[o:o + n], g["pred"][base:base + n]), (i, comp)
            o += n


def test_affine_pred_batch_golden():
    """vvcgpu_affine_pred_batch (sub-block vectors + packed 4x4 interpolation, four sub-blocks per wavefront) reproduces the reference's own
    xPredAffineBlk prediction of the fixture, luma and both chroma planes; and bi-predictive PUs (two lists with different vectors) equal the
    oracle's orc_affine_subblock_descs -> orc_mc_batch."""
    from vvcsoftware_vtm_amd import ops
    g = np.load(os.path.join(G, "affine_mv.npz"))
    W, H, bd, M = 256, 128, 10, 144
    planes = [g["Y"], g["Cb"], g["Cr"]]
    pads = [np.ascontiguousarray(np.pad(pl, M >> (1 if c else 0), mode="edge")) for c, pl in enumerate(planes)]
    dpads = [dev(a) for a in pads]
    rows = g["rows"]
    for comp in range(3):
        c = 1 if comp else 0
        pus = np.zeros(rows.shape[0], ops.AFFINE_PU)
        first, dst_off = 0, 0
        for i, r in enumerate(rows):
            px, py, w, h, six = r[:5]
            mv = np.zeros((2, 3, 2), np.int32); mv[0] = r[5:11].reshape(3, 2)
            pus[i] = (px, py, w, h, six, 0, mv, dst_off, w >> c, first)
            first += (w // 4) * (h // 4)
            dst_off += (w >> c) * (h >> c)
        dst = torch.zeros(dst_off, dtype=torch.int16, device="cuda")
        ops.affine_pred_batch(dpads[comp], None, dst, ops.struct_to_device(pus), rows.shape[0], int(first), c, W, H, (M >> c, M >> c),
                              pads[comp].shape[1], pads[comp].shape[1], bd, (0, 1023))
        got = dst.cpu().numpy()
        o = 0
        for i, r in enumerate(rows):
            w, h = int(r[2]), int(r[3])
            base = sum(((int(q[2]) * int(q[3])) + 2 * ((int(q[2]) >> 1) * (int(q[3]) >> 1))) for q in rows[:i])
            if comp == 1:
                base += w * h
            elif comp == 2:
                base += w * h + (w >> 1) * (h >> 1)
            n = (w >> c) * (h >> c)
            assert np.array_equal(got[o:o + n], g["pred"][base:base + n]), (i, comp)
            o += n
    # bi-prediction, luma: list 1 = the vectors of the next row, second reference = the Cb-sized noise plane stretched (any plane will do)
    rng = np.random.default_rng(3)
    ref1 = np.ascontiguousarray(np.pad(rng.integers(0, 1024, (H, W)).astype(np.int16), M, mode="edge"))
    pus = np.zeros(rows.shape[0], ops.AFFINE_PU)
    first, dst_off = 0, 0
    for i, r in enumerate(rows):
        px, py, w, h, six = r[:5]
        mv = np.zeros((2, 3, 2), np.int32); mv[0] = r[5:11].reshape(3, 2); mv[1] = rows[(i + 1) % len(rows)][5:11].reshape(3, 2)
        pus[i] = (px, py, w, h, six, i % 3 != 0, mv, dst_off, w, first)
        first += (w // 4) * (h // 4)
        dst_off += w * h
    wd = np.zeros(first, ops.MC_DESC)
    oracle().orc_affine_subblock_descs(p(pus), rows.shape[0], 0, W, H, 128, 128, M, M, pads[0].shape[1], ref1.shape[1], p(wd))
    want = np.zeros(dst_off, np.int16)
    oracle().orc_mc_batch(p(pads[0]), p(ref1), p(want), p(wd), int(first), bd, 0, 1023)
    dst = torch.zeros(dst_off, dtype=torch.int16, device="cuda")
    ops.affine_pred_batch(dpads[0], dev(ref1), dst, ops.struct_to_device(pus), rows.shape[0], int(first), 0, W, H, (M, M), pads[0].shape[1], ref1.shape[1],
                          bd, (0, 1023))
    assert np.array_equal(dst.cpu().numpy(), want)


def test_affine_pred_batch_reference_samples_outside_the_bit_depth():
    """the packed 4x4 interpolation narrows its first pass per sample; sub-blocks whose window holds a sample outside the bit depth (what a caller may hand
    in; the reference filters it as it finds it) are served by the sample-wise body instead (round 6): uni- and bi-predictive PUs against
    orc_affine_subblock_descs -> orc_mc_batch on planes with such samples sprinkled in"""
    from vvcsoftware_vtm_amd import ops
    g = np.load(os.path.join(G, "affine_mv.npz"))
    W, H, bd, M = 256, 128, 10, 144
    rng = np.random.default_rng(17)
    rows = g["rows"]

    def wild(plane):
        a = plane.astype(np.int32)
        m = rng.random(a.shape) < 0.002
        a[m] = rng.choice(np.array([-40, -1, 1024, 1100, 3069]), int(m.sum()))
        return a.astype(np.int16)
    ref0 = np.ascontiguousarray(np.pad(wild(g["Y"]), M, mode="edge"))
    ref1 = np.ascontiguousarray(np.pad(wild(rng.integers(0, 1024, (H, W)).astype(np.int16)), M, mode="edge"))
    pus = np.zeros(rows.shape[0], ops.AFFINE_PU)
    first, dst_off = 0, 0
    for i, r in enumerate(rows):
        px, py, w, h, six = r[:5]
        mv = np.zeros((2, 3, 2), np.int32); mv[0] = r[5:11].reshape(3, 2); mv[1] = rows[(i + 1) % len(rows)][5:11].reshape(3, 2)
        pus[i] = (px, py, w, h, six, i % 2, mv, dst_off, w, first)
        first += (w // 4) * (h // 4)
        dst_off += w * h
    wd = np.zeros(first, ops.MC_DESC)
    oracle().orc_affine_subblock_descs(p(pus), rows.shape[0], 0, W, H, 128, 128, M, M, ref0.shape[1], ref1.shape[1], p(wd))
    want = np.zeros(dst_off, np.int16)
    oracle().orc_mc_batch(p(ref0), p(ref1), p(want), p(wd), int(first), bd, 0, 1023)
    dst = torch.zeros(dst_off, dtype=torch.int16, device="cuda")
    ops.affine_pred_batch(dev(ref0), dev(ref1), dst, ops.struct_to_device(pus), rows.shape[0], int(first), 0, W, H, (M, M), ref0.shape[1], ref1.shape[1],
                          bd, (0, 1023))
    assert np.array_equal(dst.cpu().numpy(), want)


def test_affine_me_iteration_fused():
    """vvcgpu_affine_me_iter_batch (sub-block vectors -> prediction -> error, Sobel planes, equation sums and distortion in one pass) against the
    oracle's chain of the same steps: orc_affine_subblock_descs -> orc_mc_batch -> org - pred -> orc_affine_sobel_batch x2 ->
    orc_affine_equal_coeff_batch, orc_dist_batch.  PUs and control-point vectors of the reference's own affine fixture, 4- and 6-parameter."""
    from vvcsoftware_vtm_amd import ops
    g = np.load(os.path.join(G, "affine_mv.npz"))
    W, H, bd, M = 256, 128, 10, 144
    ref = np.ascontiguousarray(np.pad(g["Y"], M, mode="edge"))
    rng = np.random.default_rng(5)
    org = np.clip(g["Y"].astype(np.int32) + rng.integers(-40, 41, g["Y"].shape), 0, 1023).astype(np.int16)
    org[:, 128:] = (2 * org[:, 128:].astype(np.int32) - rng.integers(0, 1024, org[:, 128:].shape)).astype(np.int16)   # "2 org - other prediction" blocks
    rows = [r for r in g["rows"] if r[2] >= 16 and r[3] >= 16]
    assert len(rows) >= 8 and any(r[4] for r in rows) and not all(r[4] for r in rows)
    for kind in (1, 0):
        items = np.zeros(len(rows), ops.AFFINE_ITER)
        first, dst_off = 0, 0
        for i, r in enumerate(rows):
            px, py, w, h, six = (int(v) for v in r[:5])
            mv = np.zeros((2, 3, 2), np.int32); mv[0] = r[5:11].reshape(3, 2)
            items[i]["pu"] = (px, py, w, h, six, 0, mv, dst_off, w, first)
            items[i]["org_off"], items[i]["org_stride"] = py * W + px, W
            first += (w // 4) * (h // 4)
            dst_off += w * h
        # oracle chain
        pus = np.ascontiguousarray(items["pu"])
        wd = np.zeros(first, ops.MC_DESC)
        oracle().orc_affine_subblock_descs(p(pus), len(rows), 0, W, H, 128, 128, M, M, ref.shape[1], ref.shape[1], p(wd))
        wpred = np.zeros(dst_off, np.int16)
        oracle().orc_mc_batch(p(ref), p(ref), p(wpred), p(wd), int(first), bd, 0, 1023)
        resi = np.zeros(dst_off, np.int16)
        gd, ed, dd = [], [], []
        for it in items:
            w, h, o = int(it["pu"]["w"]), int(it["pu"]["h"]), int(it["pu"]["dst_off"])
            y, x = divmod(int(it["org_off"]), W)
            resi[o:o + w * h] = (org[y:y + h, x:x + w].astype(np.int32) - wpred[o:o + w * h].reshape(h, w)).astype(np.int16).reshape(-1)
            gd.append((o, o, w, w, w, h, 0)); ed.append((o, o, w, w, h, int(it["pu"]["six_param"]), 0))
            dd.append((int(it["org_off"]), o, W, w, w, h, 0, 0))
        gd, ed, dd = np.array(gd, dtype=ops.AFG_DESC), np.array(ed, dtype=ops.AFE_DESC), np.array(dd, dtype=ops.DIST_DESC)
        wx = np.zeros(dst_off, np.int32); wy = np.zeros(dst_off, np.int32)
        oracle().orc_affine_sobel_batch(0, p(wpred), p(wx), p(gd), len(gd))
        oracle().orc_affine_sobel_batch(1, p(wpred), p(wy), p(gd), len(gd))
        wcoef = np.zeros((len(ed), 49), np.int64)
        oracle().orc_affine_equal_coeff_batch(p(resi), p(wx), p(wy), p(ed), len(ed), p(wcoef))
        wdist = np.zeros(len(dd), np.uint64)
        oracle().orc_dist_batch(kind, p(org), p(wpred), p(dd), len(dd), p(wdist))
        # device
        pred = torch.zeros(dst_off, dtype=torch.int16, device="cuda")
        coef, dist = ops.affine_me_iter_batch(dev(org), dev(ref), pred, ops.struct_to_device(items), len(rows), int(first), kind, W, H, (M, M),
                                              ref.shape[1], bd, (0, 1023))
        assert np.array_equal(pred.cpu().numpy(), wpred)
        assert np.array_equal(coef.cpu().numpy().reshape(len(rows), 49), wcoef)
        assert np.array_equal(dist.cpu().numpy().view(np.uint64), wdist), kind
    coef2, none = ops.affine_me_iter_batch(dev(org), dev(ref), pred, ops.struct_to_device(items), len(rows), int(first), 0, W, H, (M, M), ref.shape[1],
                                           bd, (0, 1023), want_dist=False)
    assert none is None and np.array_equal(coef2.cpu().numpy().reshape(len(rows), 49), wcoef)
