"""GPU parity: HIP in-loop filter kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest
import torch

import cases
from oraclelib import oracle, p

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(a).cuda()


ALF_SHAPES = [(64, 64, 64), (208, 120, 64), (416, 240, 128), (1920, 1080, 128), (136, 72, 32)]


@pytest.mark.parametrize("w,h,ctu", ALF_SHAPES)
@pytest.mark.parametrize("bd,kind", [(10, "uniform"), (10, "smooth"), (8, "uniform"), (10, "extreme")])
def test_alf_classify(w, h, ctu, bd, kind):
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(w * 7 + h + bd)
    Y = cases.rand_plane(rng, h, w, bd, kind)
    want = np.zeros((h // 4, w // 4), np.uint16)
    oracle().orc_alf_classify(p(Y), w, w, h, bd, p(want))
    got = ops.alf_classify(dev(Y), bd).cpu().numpy().view(np.uint16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h,ctu", ALF_SHAPES)
@pytest.mark.parametrize("ft", [0, 1])
@pytest.mark.parametrize("bd,kind", [(10, "uniform"), (10, "smooth"), (8, "uniform"), (10, "extreme")])
def test_alf_filter(w, h, ctu, ft, bd, kind):
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(w * 3 + h + bd + ft)
    mx = (1 << bd) - 1
    Y = cases.rand_plane(rng, h, w, bd, kind)
    Cb = cases.rand_plane(rng, h // 2, w // 2, bd, kind)
    lc, cc = cases.alf_coeffs(rng, 200 if kind == "extreme" else 60)
    nx, ny = cases.n_ctus(w, h, ctu)
    enY = rng.integers(0, 2, nx * ny).astype(np.uint8)
    enC = rng.integers(0, 2, nx * ny).astype(np.uint8)
    cls = np.zeros((h // 4, w // 4), np.uint16)
    oracle().orc_alf_classify(p(Y), w, w, h, bd, p(cls))
    wantY, wantC = Y.copy(), Cb.copy()
    oracle().orc_alf_filter_luma(p(Y), w, p(wantY), w, w, h, ctu, p(cls), ft, p(lc), p(enY), 0, mx)
    oracle().orc_alf_filter_chroma(p(Cb), w // 2, p(wantC), w // 2, w // 2, h // 2, ctu // 2, p(cc), p(enC), 0, mx)
    dY = torch.full((h, w), -1, dtype=torch.int16, device="cuda")
    dC = torch.full((h // 2, w // 2), -1, dtype=torch.int16, device="cuda")
    ops.alf_filter_luma(dev(Y), dY, ctu, dev(cls.view(np.int16)), ft, lc, dev(enY), (0, mx))
    ops.alf_filter_chroma(dev(Cb), dC, ctu // 2, cc, dev(enC), (0, mx))
    assert np.array_equal(dY.cpu().numpy(), wantY)
    assert np.array_equal(dC.cpu().numpy(), wantC)
    # all-enabled (NULL flag array) path
    wantY2 = Y.copy()
    oracle().orc_alf_filter_luma(p(Y), w, p(wantY2), w, w, h, ctu, p(cls), ft, p(lc), None, 0, mx)
    ops.alf_filter_luma(dev(Y), dY, ctu, dev(cls.view(np.int16)), ft, lc, None, (0, mx))
    assert np.array_equal(dY.cpu().numpy(), wantY2)


def test_alf_strided_padded_picture():
    """planes living inside a padded picture buffer (reference layout: margin + stride, Buffer.cpp:323-366)."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(99)
    w, h, bd, ctu, m = 208, 120, 10, 64, 16
    Y = cases.rand_plane(rng, h, w, bd)
    lc, _ = cases.alf_coeffs(rng)
    cls = np.zeros((h // 4, w // 4), np.uint16)
    oracle().orc_alf_classify(p(Y), w, w, h, bd, p(cls))
    want = Y.copy()
    oracle().orc_alf_filter_luma(p(Y), w, p(want), w, w, h, ctu, p(cls), 1, p(lc), None, 0, 1023)
    big = torch.full((h + 2 * m, w + 2 * m + 8), 777, dtype=torch.int16, device="cuda")   # garbage margin
    big[m:m + h, m:m + w] = dev(Y)
    out = torch.zeros_like(big)
    src_v, dst_v = big[m:m + h, m:m + w], out[m:m + h, m:m + w]
    got_cls = ops.alf_classify(src_v, bd)
    assert np.array_equal(got_cls.cpu().numpy().view(np.uint16), cls)
    ops.alf_filter_luma(src_v, dst_v, ctu, got_cls, 1, lc, None)
    assert np.array_equal(dst_v.cpu().numpy(), want)
    assert int(out.sum()) == int(dst_v.sum())          # nothing written outside the valid area


SAO_SHAPES = [(64, 64, 64, 64), (208, 120, 64, 64), (130, 70, 64, 64), (96, 72, 32, 32), (1920, 1080, 128, 128),
              (960, 540, 64, 64)]


@pytest.mark.parametrize("w,h,cw,ch", SAO_SHAPES)
@pytest.mark.parametrize("bd,kind,full", [(10, "uniform", True), (10, "flat", False), (8, "uniform", False),
                                          (10, "extreme", True)])
def test_sao_apply(w, h, cw, ch, bd, kind, full):
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(w + 5 * h + bd)
    mx = (1 << bd) - 1
    Y = cases.rand_plane(rng, h, w, bd, kind)
    prm = cases.sao_params(rng, w, h, cw, ch, full)
    want = Y.copy()
    oracle().orc_sao_apply(p(Y), w, p(want), w, w, h, cw, ch, bd, p(prm), 0, mx)
    dst = torch.full((h, w), -1, dtype=torch.int16, device="cuda")
    ops.sao_apply(dev(Y), dst, cw, ch, bd, ops.sao_params_to_device(prm), (0, mx))
    assert np.array_equal(dst.cpu().numpy(), want)


def test_sao_apply_wide_offsets_and_narrow_clip():
    """offsets the packed 6-bit form cannot hold (the ABI takes any int16: the kernel falls back to a select chain per CTU), mixed with CTUs whose
    offsets fit; and a clipping range narrower than the sample range -- a sample that takes no offset must come through unclipped"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(77)
    w, h, c, bd = 320, 192, 64, 10
    Y = cases.rand_plane(rng, h, w, bd, "uniform")
    prm = cases.sao_params(rng, w, h, c, c, False, types=[0, 1, 2, 3, 4])
    big = rng.random(prm.size) < 0.5
    prm["offset"][big] = rng.integers(-300, 301, (int(big.sum()), 32))
    prm["offset"][~big, :5] = rng.choice(np.array([-32, -31, 31, 0, 7]), (int((~big).sum()), 5))      # the edges of the packed range
    for (cmin, cmax) in ((0, 1023), (64, 940)):
        want = Y.copy()
        oracle().orc_sao_apply(p(Y), w, p(want), w, w, h, c, c, bd, p(prm), cmin, cmax)
        dst = torch.full((h, w), -1, dtype=torch.int16, device="cuda")
        ops.sao_apply(dev(Y), dst, c, c, bd, ops.sao_params_to_device(prm), (cmin, cmax))
        assert np.array_equal(dst.cpu().numpy(), want), (cmin, cmax)


@pytest.mark.parametrize("w,h,ctu", [(416, 240, 128), (320, 192, 64), (272, 136, 32), (1936, 1096, 128), (208, 120, 64)])
def test_sao_apply_picture_strip_form_against_oracle(w, h, ctu):
    """the strip form behind vvcgpu_sao_apply_picture (a wave walks down a tile, neighbour samples by lane shifts, packed arithmetic): every type, CTUs
    whose offsets do not fit its byte table (select chain), a clipping range narrower than the sample range, tiles that span several CTUs (chroma of
    small CTUs: per-lane types) and bands cut by the last picture row -- each plane against the oracle"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(5 * w + h + ctu)
    bd = 10
    shp = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    planes = [cases.rand_plane(rng, a, b, bd, "uniform" if i == 0 else "smooth") for i, (a, b) in enumerate(shp)]
    prms = []
    for i, (a, b) in enumerate(shp):
        c = ctu if i == 0 else ctu // 2
        prm = cases.sao_params(rng, b, a, c, c, False, types=[-1, 0, 1, 2, 3, 4])
        big = rng.random(prm.size) < 0.25
        prm["offset"][big] = rng.integers(-300, 301, (int(big.sum()), 32))
        prms.append(prm)
    for (cmin, cmax) in ((0, 1023), (64, 940)):
        got = ops.sao_apply_picture([dev(x) for x in planes], [torch.full(x.shape, -1, dtype=torch.int16, device="cuda") for x in planes], ctu, bd,
                                    [ops.sao_params_to_device(q) for q in prms], (cmin, cmax))
        for i, (a, b) in enumerate(shp):
            c = ctu if i == 0 else ctu // 2
            want = planes[i].copy()
            oracle().orc_sao_apply(p(planes[i]), b, p(want), b, b, a, c, c, bd, p(prms[i]), cmin, cmax)
            assert np.array_equal(got[i].cpu().numpy(), want), (i, cmin, cmax)


def test_sao_each_type_alone():
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(3)
    w, h, c, bd = 256, 128, 64, 10
    Y = cases.rand_plane(rng, h, w, bd, "flat")
    for t in (-1, 0, 1, 2, 3, 4):
        prm = cases.sao_params(rng, w, h, c, c, True, types=[t])
        want = Y.copy()
        oracle().orc_sao_apply(p(Y), w, p(want), w, w, h, c, c, bd, p(prm), 0, 1023)
        dst = torch.full((h, w), -1, dtype=torch.int16, device="cuda")
        ops.sao_apply(dev(Y), dst, c, c, bd, ops.sao_params_to_device(prm))
        assert np.array_equal(dst.cpu().numpy(), want), t


@pytest.mark.parametrize("w,h,ctu", [(416, 240, 128), (208, 120, 64), (1920, 1080, 128), (264, 136, 64)])
@pytest.mark.parametrize("ft", [0, 1])
def test_picture_level_entry_points_equal_per_plane_ones(w, h, ctu, ft):
    """vvcgpu_sao_apply_picture / sao_stats_picture / alf_filter_picture / alf_stats_picture (three planes per launch; luma 5x5 covariances
    gathered from the 7x7 sums) give exactly what the per-plane entry points give (each of which is checked against the oracle above)"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(w + 3 * h + ft)
    bd, mx = 10, 1023
    shp = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    org = [dev(cases.rand_plane(rng, a, b, bd, "smooth")) for a, b in shp]
    rec = [dev(np.clip(o.cpu().numpy() + rng.integers(-9, 10, o.shape), 0, mx).astype(np.int16)) for o in org]
    prm = [ops.sao_params_to_device(cases.sao_params(rng, b, a, ctu if i == 0 else ctu // 2, ctu if i == 0 else ctu // 2)) for i, (a, b) in enumerate(shp)]
    # SAO apply
    want = [torch.empty_like(r) for r in rec]
    for c in range(3):
        cs = ctu if c == 0 else ctu // 2
        ops.sao_apply(rec[c], want[c], cs, cs, bd, prm[c], (0, mx))
    got = ops.sao_apply_picture(rec, [torch.empty_like(r) for r in rec], ctu, bd, prm, (0, mx))
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # SAO statistics
    ws = [ops.sao_stats(org[c], rec[c], ctu if c == 0 else ctu // 2, ctu if c == 0 else ctu // 2, bd, None, 5 if c == 0 else 3, 4 if c == 0 else 2) for c in range(3)]
    gs = ops.sao_stats_picture(org, rec, ctu, bd)
    for a, b in zip(gs, ws):
        assert torch.equal(a, b)
    # ALF filter + covariances (picture size must be a multiple of 8 for the picture-level forms)
    cls = ops.alf_classify(got[0], bd)
    lc, cc = cases.alf_coeffs(rng)
    nctu = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
    en = [dev((rng.random(nctu) < 0.7).astype(np.uint8)) for _ in range(3)]
    wf = [torch.empty_like(r) for r in rec]
    ops.alf_filter_luma(got[0], wf[0], ctu, cls, ft, lc, en[0], (0, mx))
    for c in (1, 2):
        ops.alf_filter_chroma(got[c], wf[c], ctu // 2, cc, en[c], (0, mx))
    gf = ops.alf_filter_picture(got, [torch.empty_like(r) for r in rec], ctu, cls, ft, lc, cc, en, (0, mx))
    for a, b in zip(gf, wf):
        assert torch.equal(a, b)
    a7, a5, ac = ops.alf_stats_picture(org, got, ctu, cls)
    assert torch.equal(a7, ops.alf_stats(org[0], got[0], ctu, cls, 1))
    assert torch.equal(a5, ops.alf_stats(org[0], got[0], ctu, cls, 0))
    for i, c in enumerate((1, 2)):
        if ctu >= 128:
            assert torch.equal(ac[i], ops.alf_stats(org[c], got[c], ctu // 2, None, 0))
        else:                      # the per-plane entry point has no 32-sample CTUs: the oracle directly
            o, r = org[c].cpu().numpy(), got[c].cpu().numpy()
            want = np.zeros(ac[i].numel(), np.int64)
            oracle().orc_alf_stats(p(o), o.shape[1], p(r), r.shape[1], o.shape[1], o.shape[0], ctu // 2, None, 0, p(want))
            assert np.array_equal(ac[i].cpu().numpy().ravel(), want)
    # the classifier inside the covariance launch: the classes and every record as from the two launches
    fcls, f7, f5, fc = ops.alf_classify_stats_picture(org, got, ctu, bd)
    assert torch.equal(fcls, cls) and torch.equal(f7, a7) and torch.equal(f5, a5) and torch.equal(fc[0], ac[0]) and torch.equal(fc[1], ac[1])


# (from 320 CTUs on the classes are derived inside the covariance workgroups; smaller pictures and other CTU sizes run the classifier's own launch)
@pytest.mark.parametrize("w,h,ctu", [(128, 128, 128), (136, 72, 64), (264, 136, 128), (416, 240, 128), (264, 136, 256), (1216, 1088, 64), (1240, 1096, 64)])
@pytest.mark.parametrize("bd,kind", [(10, "uniform"), (10, "extreme"), (8, "smooth"), (10, "const")])
def test_alf_classify_stats_picture_against_oracle(w, h, ctu, bd, kind):
    """vvcgpu_alf_classify_stats_picture: classes = the oracle's deriveClassification of the reconstruction, luma 7x7 records = the oracle's
    covariances over those classes (partial CTUs at the right / bottom edge, a CTU size without the CTU form, content that saturates the
    Laplacians, a constant picture where every direction ties)"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(7 * w + h + bd)
    mx = (1 << bd) - 1
    shp = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    if kind == "const":
        rec = [np.full(s_, 512 if bd == 10 else 128, np.int16) for s_ in shp]
    else:
        rec = [cases.rand_plane(rng, a, b, bd, kind) for a, b in shp]
    org = [np.clip(r.astype(np.int32) + rng.integers(-20, 21, r.shape), 0, mx).astype(np.int16) for r in rec]
    cls = np.zeros((h // 4, w // 4), np.uint16)
    oracle().orc_alf_classify(p(rec[0]), w, w, h, bd, p(cls))
    nx, ny = cases.n_ctus(w, h, ctu)
    want7 = np.zeros((nx * ny, 25, 183), np.int64)
    oracle().orc_alf_stats(p(org[0]), w, p(rec[0]), w, w, h, ctu, p(cls), 1, p(want7))
    gcls, g7, g5, gc = ops.alf_classify_stats_picture([dev(o) for o in org], [dev(r) for r in rec], ctu, bd)
    assert np.array_equal(gcls.cpu().numpy().view(np.uint16), cls)
    assert np.array_equal(g7.cpu().numpy(), want7)
    want5 = np.zeros((nx * ny, 25, 57), np.int64)
    oracle().orc_alf_stats(p(org[0]), w, p(rec[0]), w, w, h, ctu, p(cls), 0, p(want5))
    assert np.array_equal(g5.cpu().numpy(), want5)


def test_alf_classify_stats_picture_128_ctus_in_kernel_classes():
    """the same with 128-sample CTUs at a size that takes the in-kernel classifier (20 x 18 CTUs, partial ones at the right and bottom edge)"""
    test_alf_classify_stats_picture_against_oracle(2440, 2184, 128, 10, "smooth")
