"""GPU parity: fused fractional refinement (planes in LDS + 9+9 SATD/SAD candidates + MV cost) vs the CPU oracle."""
import ctypes as C
import numpy as np
import pytest
import torch

import cases
from oraclelib import oracle, p

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (16, 8), (8, 16), (32, 32), (64, 64), (8, 4), (4, 8), (32, 16),
                                 (128, 128), (64, 32), (16, 64), (128, 64)])
@pytest.mark.parametrize("bd,kind,had", [(10, "smooth", 1), (10, "uniform", 1), (8, "smooth", 1), (10, "smooth", 0), (10, "extreme", 1)])
def test_frac_refine(w, h, bd, kind, had):
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(w * 5 + h + bd + had)
    mx = (1 << bd) - 1
    W, H, M = 256, 224, 16
    ref = cases.rand_plane(rng, H + 2 * M, W + 2 * M, bd, kind)
    org = ref[M + 1:M + 1 + H, M + 2:M + 2 + W].astype(np.int32) + rng.integers(-6, 7, (H, W))
    org = np.ascontiguousarray(np.clip(org, 0, mx).astype(np.int16))
    nb = 9
    blk = np.zeros(nb, ops.FRAC_BLK)
    for i in range(nb):
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        mvx, mvy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
        blk[i] = (x, y, M + x + mvx, M + y + mvy, mvx, mvy)
    mv = ops.MvCost(float(rng.uniform(2, 40)), int(rng.integers(-20, 20)), int(rng.integers(-20, 20)), 0, 0)
    want = np.zeros(nb, ops.FRAC_RESULT)
    oracle().orc_frac_refine(p(org), W, p(ref), W + 2 * M, p(blk), nb, w, h, bd, 0, mx, had, C.byref(mv), p(want))
    got = ops.frac_refine(dev(org), dev(ref), ops.struct_to_device(blk), nb, w, h, bd, mv, bool(had), (0, mx))
    got = got.cpu().numpy().view(ops.FRAC_RESULT)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("pattern", ["flat", "checker", "random"])
def test_frac16_hadamard_at_the_int16_bound(pattern):
    """16x16 PUs take the packed 16-bit Hadamard (five stages in int16, the last as 2 max(|a|, |b|)): residuals of +-1023 in the patterns that
    put the largest coefficients into one place (flat: DC = 64 x 1023; checkerboards: the highest frequency) and random full-range content"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(7)
    bd, mx, W, H, M, w, h = 10, 1023, 128, 96, 16, 16, 16
    if pattern == "flat":
        ref = np.zeros((H + 2 * M, W + 2 * M), np.int16); org = np.full((H, W), mx, np.int16)
    elif pattern == "checker":
        yy, xx = np.mgrid[0:H + 2 * M, 0:W + 2 * M]
        ref = (((yy + xx) & 1) * mx).astype(np.int16)
        org = np.ascontiguousarray((((yy + xx + 1) & 1) * mx).astype(np.int16)[M:M + H, M:M + W])
    else:
        ref = (rng.integers(0, 2, (H + 2 * M, W + 2 * M)) * mx).astype(np.int16)
        org = (rng.integers(0, 2, (H, W)) * mx).astype(np.int16)
    nb = 12
    blk = np.zeros(nb, ops.FRAC_BLK)
    for i in range(nb):
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        blk[i] = (x, y, M + x, M + y, 0, 0)
    mv = ops.MvCost(4.0, 0, 0, 0, 0)
    want = np.zeros(nb, ops.FRAC_RESULT)
    oracle().orc_frac_refine(p(org), W, p(ref), W + 2 * M, p(blk), nb, w, h, bd, 0, mx, 1, C.byref(mv), p(want))
    got = ops.frac_refine(dev(org), dev(ref), ops.struct_to_device(blk), nb, w, h, bd, mv, True, (0, mx)).cpu().numpy().view(ops.FRAC_RESULT)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h", [(16, 16), (32, 32), (8, 8)])
@pytest.mark.parametrize("kind", ["bipred", "extreme"])
def test_frac_refine_bipred_original(w, h, kind):
    """Bi-predictive refinement hands in 2 org - otherPred (InterSearch.cpp:1682-1692, ClipForBiPredMEEnabled off by default): samples in
    [-1023, 2046], so |org - pred| reaches 2046 and the packed 16-bit Hadamard of the 16x16 kernel does not hold -- such PUs take its
    32-bit form (a wave-uniform test of the original's range); mixed lists, so both forms run in one launch."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(w + 31 * h + len(kind))
    bd, mx, W, H, M = 10, 1023, 192, 160, 16
    ref = cases.rand_plane(rng, H + 2 * M, W + 2 * M, bd, "smooth")
    base = cases.rand_plane(rng, H, W, bd, "smooth").astype(np.int32)
    if kind == "bipred":
        org = 2 * base - cases.rand_plane(rng, H, W, bd, "uniform")
    else:                                                        # the corners of the range, in checkerboards (largest Hadamard sums)
        yy, xx = np.mgrid[0:H, 0:W]
        org = np.where((yy + xx) & 1, 2046, -1023)
    org[:, :W // 2] = np.clip(org[:, :W // 2], 0, mx)          # left half in range: those PUs keep the packed form
    org = np.ascontiguousarray(org.astype(np.int16))
    assert org.min() < 0 and org.max() > mx
    nb = 24
    blk = np.zeros(nb, ops.FRAC_BLK)
    for i in range(nb):
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        mvx, mvy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
        blk[i] = (x, y, M + x + mvx, M + y + mvy, mvx, mvy)
    mv = ops.MvCost(float(rng.uniform(2, 40)), int(rng.integers(-20, 20)), int(rng.integers(-20, 20)), 0, 0)
    want = np.zeros(nb, ops.FRAC_RESULT)
    oracle().orc_frac_refine(p(org), W, p(ref), W + 2 * M, p(blk), nb, w, h, bd, 0, mx, 1, C.byref(mv), p(want))
    got = ops.frac_refine(dev(org), dev(ref), ops.struct_to_device(blk), nb, w, h, bd, mv, True, (0, mx)).cpu().numpy().view(ops.FRAC_RESULT)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("bd,kind", [(10, "smooth"), (10, "extreme"), (8, "uniform"), (10, "wild")])
def test_frac16_every_alignment(bd, kind):
    """The matrix-core refinement of 16x16 PUs (frac16m_kernel) reads ALIGNED 16-byte words and selects its Toeplitz table row by the window's
    start inside them: every start (reference x mod 16, y mod 4) x every original x mod 8, 512 PUs of one launch, each with its 9 half- and
    9 quarter-sample Hadamard candidates.  'wild': reference samples outside the bit depth -- those PUs are flagged and served
    by the vector-pipe body."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(bd * 7 + len(kind))
    mx = (1 << bd) - 1
    W, H, M, w, h = 320, 192, 16, 16, 16
    ref = cases.rand_plane(rng, H + 2 * M, W + 2 * M, bd, "smooth" if kind == "wild" else kind).astype(np.int32)
    if kind == "wild":
        m = rng.random(ref.shape) < 0.001
        ref[m] = rng.choice(np.array([-9, -1, mx + 1, mx + 70]), int(m.sum()))
    ref = ref.astype(np.int16)
    org = np.clip(ref[M + 1:M + 1 + H, M + 2:M + 2 + W].astype(np.int32), 0, mx) + rng.integers(-6, 7, (H, W))
    org = np.ascontiguousarray(np.clip(org, 0, mx).astype(np.int16))
    rows = []
    for ax in range(16):
        for ay in range(4):
            for ox in range(8):
                x = 16 * int(rng.integers(1, (W - w) // 16 - 1)) + ox
                y = int(rng.integers(0, H - h + 1))
                rx = 16 * int(rng.integers(1, (W - w) // 16 - 1)) + ax + M
                ry = 4 * int(rng.integers(1, (H - h) // 4 - 1)) + ay + M
                rows.append((x, y, rx, ry, rx - M - x, ry - M - y))
    blk = np.array(rows, ops.FRAC_BLK)
    nb = len(blk)
    mv = ops.MvCost(7.5, 3, -5, 0, 0)
    want = np.zeros(nb, ops.FRAC_RESULT)
    oracle().orc_frac_refine(p(org), W, p(ref), W + 2 * M, p(blk), nb, w, h, bd, 0, mx, 1, C.byref(mv), p(want))
    got = ops.frac_refine(dev(org), dev(ref), ops.struct_to_device(blk), nb, w, h, bd, mv, True, (0, mx)).cpu().numpy().view(ops.FRAC_RESULT)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad.size, bad[:5], got[bad[:3]], want[bad[:3]])
