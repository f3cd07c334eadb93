"""GPU parity: single-pass LDS-tiled deblocking kernel vs the CPU oracle (two-pass picture order)."""
import ctypes as C
import numpy as np
import pytest
import torch

import cases
from oraclelib import oracle, p

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("w,h", [(64, 64), (136, 72), (416, 240), (1920, 1080), (200, 120)])
@pytest.mark.parametrize("bd,kind,mode,offs", [(10, "smooth", "cu", (0, 0, 0, 0)), (10, "uniform", "random", (2, -1, 1, -2)),
                                               (8, "smooth", "cu", (-2, 3, 0, 0)), (10, "flat", "random", (0, 0, 5, 7)),
                                               (10, "extreme", "cu", (6, 6, 0, 0))])
def test_deblock(w, h, bd, kind, mode, offs):
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(w + 3 * h + bd)
    Y = cases.rand_plane(rng, h, w, bd, kind)
    Cb = cases.rand_plane(rng, h // 2, w // 2, bd, kind)
    Cr = cases.rand_plane(rng, h // 2, w // 2, bd, kind)
    ev, eh, qpl, qpc = cases.deblock_maps(rng, w, h, mode)
    cfg = ops.deblock_cfg(bd, *offs)
    wY, wCb, wCr = Y.copy(), Cb.copy(), Cr.copy()
    oracle().orc_deblock(p(wY), w, p(wCb), p(wCr), w // 2, w, h, p(ev), p(eh), p(qpl), p(qpc), C.byref(cfg))
    if w > 128:
        assert not (np.array_equal(wY, Y) and np.array_equal(wCb, Cb) and np.array_equal(wCr, Cr))
    dY, dCb, dCr = dev(Y), dev(Cb), dev(Cr)
    ops.deblock(dY, dCb, dCr, dev(ev), dev(eh), dev(qpl), dev(qpc), cfg)
    assert np.array_equal(dY.cpu().numpy(), wY)
    assert np.array_equal(dCb.cpu().numpy(), wCb)
    assert np.array_equal(dCr.cpu().numpy(), wCr)


def test_deblock_unaligned_rows_tile_form():
    """planes whose rows are no whole 8-byte words (a stride of w + 2 samples) take the LDS tile form of the kernel; the block form serves the rest"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(91)
    w, h, bd = 264, 136, 10
    Y = cases.rand_plane(rng, h, w, bd, "uniform")
    Cb = cases.rand_plane(rng, h // 2, w // 2, bd, "smooth")
    Cr = cases.rand_plane(rng, h // 2, w // 2, bd, "uniform")
    ev, eh, qpl, qpc = cases.deblock_maps(rng, w, h, "random")
    cfg = ops.deblock_cfg(bd, 1, -1, 2, -2)
    wY, wCb, wCr = Y.copy(), Cb.copy(), Cr.copy()
    oracle().orc_deblock(p(wY), w, p(wCb), p(wCr), w // 2, w, h, p(ev), p(eh), p(qpl), p(qpc), C.byref(cfg))
    big = [torch.zeros((a.shape[0], a.shape[1] + 2), dtype=torch.int16, device="cuda") for a in (Y, Cb, Cr)]
    views = [b[:, :a.shape[1]] for a, b in zip((Y, Cb, Cr), big)]
    for v, a in zip(views, (Y, Cb, Cr)):
        v.copy_(dev(a))
    ops.deblock(views[0], views[1], views[2], dev(ev), dev(eh), dev(qpl), dev(qpc), cfg)
    for v, want in zip(views, (wY, wCb, wCr)):
        assert np.array_equal(v.cpu().numpy(), want)
    assert all(int(b[:, -2:].abs().sum()) == 0 for b in big)          # nothing written beside the planes


def test_deblock_luma_only_and_idempotent_on_flat():
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(5)
    w, h, bd = 256, 128, 10
    Y = np.full((h, w), 512, np.int16)
    ev, eh, qpl, qpc = cases.deblock_maps(rng, w, h, "cu")
    cfg = ops.deblock_cfg(bd)
    dY = dev(Y)
    ops.deblock(dY, None, None, dev(ev), dev(eh), dev(qpl), None, cfg)
    assert np.array_equal(dY.cpu().numpy(), Y)      # a flat picture is a fixed point of every filter


def test_deblock_reference_golden():
    """the kernel on the compiled reference's own pictures (tests/golden/deblock.npz: planes in front of LoopFilter::loopFilterPic, maps from the
    reference's own CU walk, planes behind its own filters): bit-equal, three planes per picture, 416x240 inter with affine / > 64 CUs and 1920x1080 intra"""
    from vvcsoftware_vtm_amd import ops
    for r in cases.deblock_golden():
        h = r["hdr"]
        cfg = ops.DeblockCfg(h["bd_luma"], h["bd_chroma"], h["beta_offset_div2"], h["tc_offset_div2"], h["cb_qp_offset"], h["cr_qp_offset"],
                             (C.c_int32 * 3)(h["clp_min0"], h["clp_min1"], h["clp_min2"]), (C.c_int32 * 3)(h["clp_max0"], h["clp_max1"], h["clp_max2"]))
        d = [dev(x.copy()) for x in r["pre"]]
        ops.deblock(d[0], d[1], d[2], dev(r["ev"]), dev(r["eh"]), dev(r["qp_luma"]), dev(r["qp_chroma"]), cfg)
        for got, want, name in zip(d, r["post"], "Y Cb Cr".split()):
            g = got.cpu().numpy()
            assert np.array_equal(g, want), "poc %d %s: %d samples differ" % (h["poc"], name, int((g != want).sum()))
