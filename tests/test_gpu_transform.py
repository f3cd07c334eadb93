"""GPU parity: forward / inverse 2-D transforms and transform skip vs the CPU oracle, every (W,H) x transform pair."""
import numpy as np
import pytest
import torch

import cases
from oraclelib import oracle, p

pytestmark = pytest.mark.gpu

PAIRS = [(0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (3, 0)]


def dev(a):
    return torch.from_numpy(a).cuda()


def build(rng, bd, kind, pad=3):
    """pad = 3: odd strides (rows at odd addresses -> the library's generic stages for large TUs); pad = 4: 4-byte aligned
    rows (packed 16-bit stages).  kind 3: full int16 residuals (intermediates beyond 16 bits -> 32-bit fallback stages)."""
    from vvcsoftware_vtm_amd import ops
    mx = (1 << bd) - 1
    rows, resis = [], []
    roff = coff = 0
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            for (th, tv) in PAIRS:
                if th in (1, 2) and (w < 4 or h < 4 or w > 32 or h > 32):
                    continue
                st = w + pad
                if kind == 3:
                    r = rng.integers(-32768, 32768, (h, st))
                elif kind == 0:
                    r = rng.integers(-mx, mx + 1, (h, st))
                elif kind == 1:
                    r = rng.choice(np.array([-mx, mx]), (h, st))
                else:
                    r = rng.integers(-20, 20, (h, st))
                resis.append(r.astype(np.int16).reshape(-1))
                rows.append((roff, coff, st, w, h, th, tv, 0, 0))
                roff += h * st
                coff += w * h
    return np.array(rows, dtype=ops.TR_DESC), np.concatenate(resis), coff


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("kind,pad", [(0, 3), (1, 3), (2, 3), (0, 4), (1, 4), (2, 4), (3, 4), (3, 3)])
def test_tr_fwd_inv(bd, kind, pad):
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(bd * 10 + kind + 100 * pad)
    d, resi, ncoef = build(rng, bd, kind, pad)
    want = np.full(ncoef, 9, np.int32)
    oracle().orc_tr_fwd_batch(p(resi), p(want), p(d), len(d), bd)
    got = torch.full((ncoef,), 9, dtype=torch.int32, device="cuda")
    dd = ops.struct_to_device(d)
    ops.tr_fwd_batch(dev(resi), got, dd, len(d), bd)
    assert np.array_equal(got.cpu().numpy(), want)
    # inverse of (coarsely quantised) coefficients, plus full-range coefficients for kind 1
    cf = (want >> 3) << 3
    if kind == 1:
        cf = rng.integers(-32768, 32768, ncoef).astype(np.int32)
    if kind == 3:                                   # beyond 16 bits: exact 32-bit stages
        cf = rng.integers(-(1 << 20), 1 << 20, ncoef).astype(np.int32)
    wres = np.full(resi.size, 77, np.int16)
    oracle().orc_tr_inv_batch(p(cf), p(wres), p(d), len(d), bd)
    gres = torch.full((resi.size,), 77, dtype=torch.int16, device="cuda")
    ops.tr_inv_batch(dev(cf), gres, dd, len(d), bd)
    assert np.array_equal(gres.cpu().numpy(), wres)


def test_tr_many_tus_shuffled():
    """a long shuffled descriptor list (all sizes interleaved, several 64-descriptor batches, large-TU compaction)."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(5)
    bd = 10
    rows = []
    roff = coff = 0
    W = 256
    sizes = [2, 4, 8, 16, 32, 64]
    plane = rng.integers(-700, 700, (1024, W)).astype(np.int16)
    for n in range(700):
        w, h = int(rng.choice(sizes)), int(rng.choice(sizes))
        th, tv = PAIRS[int(rng.integers(0, len(PAIRS)))]
        if th in (1, 2) and (w < 4 or h < 4 or w > 32 or h > 32):
            th, tv = 0, 0
        x, y = int(rng.integers(0, (W - w) // 2 + 1)) * 2, int(rng.integers(0, 1024 - h + 1))
        rows.append((y * W + x, coff, W, w, h, th, tv, 0, 0))
        coff += w * h
    d = np.array(rows, dtype=ops.TR_DESC)
    want = np.zeros(coff, np.int32)
    oracle().orc_tr_fwd_batch(p(plane), p(want), p(d), len(d), bd)
    got = torch.zeros(coff, dtype=torch.int32, device="cuda")
    dd = ops.struct_to_device(d)
    ops.tr_fwd_batch(dev(plane), got, dd, len(d), bd)
    assert np.array_equal(got.cpu().numpy(), want)
    # inverse into disjoint destination blocks (one 64x64 slot per TU)
    rows2 = []
    for n, r in enumerate(d):
        rows2.append(((n // 4) * 64 * W + (n % 4) * 64, int(r["coeff_off"]), W, int(r["w"]), int(r["h"]), int(r["tr_hor"]), int(r["tr_ver"]), 0, 0))
    d2 = np.array(rows2, dtype=ops.TR_DESC)
    cf = (want >> 2) << 2
    wres = np.full(((len(d) + 3) // 4) * 64 * W, 5, np.int16)
    oracle().orc_tr_inv_batch(p(cf), p(wres), p(d2), len(d2), bd)
    gres = torch.full((wres.size,), 5, dtype=torch.int16, device="cuda")
    ops.tr_inv_batch(dev(cf), gres, ops.struct_to_device(d2), len(d2), bd)
    assert np.array_equal(gres.cpu().numpy(), wres)


def test_tr_long_lists_with_prologue_items():
    """vvcgpu_tr_fwd_batch / vvcgpu_tr_inv_batch on a picture-sized list (90 k 4x4 TUs + 16x16 / 32x32 TUs in front): the chain launch behind these
    entry points runs its prologue -- one wave copies the matrix image, the others and then the copying wave serve 4x4 items from the end of the list"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(61)
    bd, W, H = 10, 2048, 1024
    plane = rng.integers(-600, 600, (H, W)).astype(np.int16)
    rows = []
    coff = 0
    ci = 0
    for y0 in range(0, H, 64):
        for x0 in range(0, W, 64):
            s_ = (4, 4, 4, 16, 4, 32)[ci % 6]
            ci += 1
            for ty in range(0, 64, s_):
                for tx in range(0, 64, s_):
                    th, tv = PAIRS[int(rng.integers(0, len(PAIRS)))] if s_ <= 32 else (0, 0)
                    rows.append(((y0 + ty) * W + x0 + tx, coff, W, s_, s_, th, tv, 0, 0))
                    coff += s_ * s_
    d = np.array(rows, dtype=ops.TR_DESC)
    d = d[np.argsort(-d["w"].astype(np.int64), kind="stable")]
    assert ((d["w"] == 4).sum() + 15) // 16 >= 7 * 768
    want = np.zeros(coff, np.int32)
    oracle().orc_tr_fwd_batch(p(plane), p(want), p(d), len(d), bd)
    got = torch.zeros(coff, dtype=torch.int32, device="cuda")
    dd = ops.struct_to_device(d)
    ops.tr_fwd_batch(dev(plane), got, dd, len(d), bd)
    assert np.array_equal(got.cpu().numpy(), want)
    cf = (want >> 3) << 3
    wres = np.full(plane.size, 3, np.int16)
    oracle().orc_tr_inv_batch(p(cf), p(wres), p(d), len(d), bd)
    gres = torch.full((plane.size,), 3, dtype=torch.int16, device="cuda")
    ops.tr_inv_batch(dev(cf), gres, dd, len(d), bd)
    assert np.array_equal(gres.cpu().numpy(), wres)


@pytest.mark.parametrize("bd", [8, 10])
def test_dequant_tr_inv(bd):
    """N1: de-quantisation (scalar and dependent quantisation, every shape, several QPs) + inverse transform vs the oracle."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(40 + bd)
    rows = []
    roff = loff = 0
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            for (th, tv) in PAIRS:
                if th in (1, 2) and (w < 4 or h < 4 or w > 32 or h > 32):
                    continue
                for dq in (0, 1):
                    qp = int(rng.integers(0, 52 + (bd - 8) * 6))
                    st = w + 4
                    rows.append((roff, loff, st, w, h, th, tv, dq, 0, qp))
                    roff += h * st
                    loff += w * h
    d = np.array(rows, dtype=ops.DQTR_DESC)
    lv = (rng.integers(-30, 31, loff) * (rng.random(loff) < 0.35)).astype(np.int32)
    lv[::977] = rng.integers(-3000, 3000, lv[::977].size)
    wres = np.full(roff, 11, np.int16)
    wcoef = np.full(loff, 3, np.int32)
    oracle().orc_dequant_tr_inv_batch(p(lv), p(wres), p(d), len(d), bd, p(wcoef))
    gres = torch.full((roff,), 11, dtype=torch.int16, device="cuda")
    gcoef = torch.full((loff,), 3, dtype=torch.int32, device="cuda")
    ops.dequant_tr_inv_batch(dev(lv), gres, ops.struct_to_device(d), len(d), bd, gcoef)
    assert np.array_equal(gcoef.cpu().numpy(), wcoef)
    assert np.array_equal(gres.cpu().numpy(), wres)
    # without the coefficient output (they stay in LDS), and in shuffled order (mixed phases inside one workgroup's batch)
    perm = rng.permutation(len(d))
    gres2 = torch.full((roff,), 11, dtype=torch.int16, device="cuda")
    ops.dequant_tr_inv_batch(dev(lv), gres2, ops.struct_to_device(np.ascontiguousarray(d[perm])), len(d), bd, None)
    assert np.array_equal(gres2.cpu().numpy(), wres)


def test_dequant_tr_inv_long_homogeneous_batches():
    """the shapes of tools/n13_time.py (many TUs of one size: 64 / 16 / 4 descriptors per workgroup), aligned rows: vector stores of the
    matrix-core form, every transform pair, both quantisers"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(77)
    bd = 10
    for (w, h, n) in ((8, 8, 5000), (16, 16, 3000), (32, 32, 900), (64, 32, 300), (16, 64, 300), (4, 4, 9000), (32, 8, 700)):
        d = np.zeros(n, ops.DQTR_DESC)
        d["resi_off"] = d["level_off"] = np.arange(n) * w * h
        d["resi_stride"], d["w"], d["h"] = w, w, h
        pair = rng.integers(0, 3, n) if max(w, h) <= 32 and min(w, h) >= 4 else np.zeros(n, np.int64)
        d["tr_hor"] = np.where(pair == 0, 0, np.where(pair == 1, 2, 1))
        d["tr_ver"] = np.where(pair == 0, 0, np.where(pair == 1, 2, 2))
        d["dep_quant"] = rng.integers(0, 2, n)
        d["qp"] = rng.integers(22, 38, n)
        lv = (rng.integers(-12, 13, n * w * h) * (rng.random(n * w * h) < 0.35)).astype(np.int32)
        wres = np.zeros(n * w * h, np.int16)
        wcoef = np.zeros(n * w * h, np.int32)
        oracle().orc_dequant_tr_inv_batch(p(lv), p(wres), p(d), n, bd, p(wcoef))
        gres = torch.zeros(n * w * h, dtype=torch.int16, device="cuda")
        ops.dequant_tr_inv_batch(dev(lv), gres, ops.struct_to_device(d), n, bd, None)
        assert np.array_equal(gres.cpu().numpy(), wres), (w, h)


def test_scan_order_host_matches_golden():
    import ctypes as C
    import os
    from vvcsoftware_vtm_amd import capi
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dequant.npz"))
    pos = 0
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            sc = np.zeros(w * h, np.uint16)
            assert capi.lib().vvcgpu_scan_order_host(w, h, sc.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(sc, g["scan"][pos:pos + w * h]), (w, h)
            pos += w * h


def test_shipped_tables_match_golden():
    """the table inside the library == tests/golden/tr_tables.npz (dumped from the compiled reference)."""
    import ctypes as C
    import os
    from vvcsoftware_vtm_amd import capi
    lib = capi.lib()
    lib.vvcgpu_tr_matrix_host.restype = C.POINTER(C.c_int16)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tr_tables.npz"))
    for t, nm in enumerate(["DCT2", "DCT8", "DST7"]):
        for lg in range(1, 7):
            N = 1 << lg
            a = np.ctypeslib.as_array(lib.vvcgpu_tr_matrix_host(t, N), shape=(N * N,)).reshape(N, N)
            assert np.array_equal(a, g["%s_%d" % (nm, N)])


@pytest.mark.parametrize("bd", [8, 10])
def test_quant_forward(bd):
    """N1 forward: Quant::quant without RDOQ + sign bit hiding, every TU shape, all in one launch, vs the oracle; then the
    golden vectors of the compiled reference."""
    import ctypes as C
    import os
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(70 + bd)
    O = oracle()
    O.orc_quant.restype = C.c_uint32
    rows, coefs, wants, sums = [], [], [], []
    off = 0
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            for it in range(6):
                n = w * h
                qp = int(rng.integers(0, 52 + (bd - 8) * 6))
                if it == 0:
                    coef = rng.integers(-2 ** 21, 2 ** 21, n)
                elif it == 1:
                    coef = rng.integers(-3, 4, n) * 37                       # many levels of 0 / 1: the hiding corner cases
                else:
                    coef = rng.normal(0, 300 * it, n) * (rng.random(n) < 0.2 * it)
                coef = coef.astype(np.int32)
                intra, sbh = int(rng.integers(0, 2)), int(it != 5)
                lv = np.zeros(n, np.int32)
                sums.append(O.orc_quant(p(coef), p(lv), w, h, bd, qp, intra, sbh))
                rows.append((off, off, w, h, intra, sbh, 0, qp, 0))
                coefs.append(coef); wants.append(lv); off += n
    d = np.array(rows, dtype=ops.QUANT_DESC)
    level = torch.full((off,), 7, dtype=torch.int32, device="cuda")
    got_sum = ops.quant_batch(dev(np.concatenate(coefs)), level, ops.struct_to_device(d), len(d), bd)
    assert np.array_equal(got_sum.cpu().numpy().view(np.uint32), np.array(sums, np.uint32))
    assert np.array_equal(level.cpu().numpy(), np.concatenate(wants))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quant.npz"))
    sel = [r for r in g["rows"] if r[2] == bd]
    d = np.array([(r[6], r[6], r[0], r[1], r[4], r[5], 0, r[3], 0) for r in sel], dtype=ops.QUANT_DESC)
    level = torch.zeros(g["coef"].size, dtype=torch.int32, device="cuda")
    got_sum = ops.quant_batch(dev(np.ascontiguousarray(g["coef"])), level, ops.struct_to_device(d), len(d), bd)
    got = level.cpu().numpy()
    for i, r in enumerate(sel):
        n = int(r[0] * r[1])
        assert np.array_equal(got[r[6]:r[6] + n], g["level"][r[6]:r[6] + n]), tuple(r)
        assert int(got_sum.cpu().numpy().view(np.uint32)[i]) == int(r[7])


def test_depquant_trellis():
    """N1: the dependent-quantisation trellis on the device.  (1) golden vectors of the compiled reference's DepQuant::quant (luma and
    chroma, per bit depth one launch with mixed TU shapes); (2) random TUs incl. all-below-threshold blocks, huge coefficients and
    lambda extremes against the oracle, many TUs per launch (sixteen share a wavefront, each with its own first position)."""
    import ctypes as C
    from vvcsoftware_vtm_amd import ops
    from test_oracle_golden import depquant_rows
    rows = list(depquant_rows())
    g = rows[0][-1]
    rates_dev = ops.struct_to_device(np.ascontiguousarray(g["rates"]).view(ops.DQ_RATES))
    coef_dev = dev(np.ascontiguousarray(g["coef"]))
    total = g["coef"].size
    for bd in (8, 10):
        sel = [r for r in rows if r[3] == bd]
        d = np.array([(r[5], r[5], r[8], r[4], r[7], r[0], r[1], 1 - r[2], (0, 0, 0)) for r in sel], dtype=ops.DEPQUANT_DESC)
        level = torch.full((total,), 9, dtype=torch.int32, device="cuda")
        sums = ops.depquant_batch(coef_dev, level, ops.struct_to_device(d), len(d), rates_dev, total, bd).cpu().numpy().view(np.uint32)
        got = level.cpu().numpy()
        for i, r in enumerate(sel):
            n = r[0] * r[1]
            assert np.array_equal(got[r[5]:r[5] + n], g["level"][r[5]:r[5] + n]), r[:8]
            assert int(sums[i]) == r[6]
    rng = np.random.default_rng(90)
    O = oracle()
    O.orc_depquant.restype = C.c_uint32
    rates = np.ascontiguousarray(g["rates"][:6]).view(ops.DQ_RATES)
    descs, coefs, wants, sums = [], [], [], []
    off = 0
    for (w, h) in [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 32), (32, 4), (8, 64), (64, 8), (16, 32), (4, 64), (64, 4)]:
        for it in range(12 if w * h <= 256 else (5 if w * h <= 1024 else 2)):
            n = w * h
            qp = int(rng.integers(0, 63))
            lam = float([0.7, 30.0, 250.0, 4000.0][it % 4])
            yy, xx = np.mgrid[0:h, 0:w]
            decay = np.exp(-(xx / w * 2.5 + yy / h * 2.5))
            kind = it % 5
            if kind == 0:
                coef = rng.normal(0, 2500, (h, w)) * decay
            elif kind == 1:
                coef = rng.integers(-3, 4, (h, w))                          # everything below the threshold: no trellis at all
            elif kind == 2:
                coef = rng.integers(-32768, 32768, (h, w))                  # full 16-bit range everywhere
            elif kind == 3:
                coef = rng.normal(0, 9000, (h, w)) * (rng.random((h, w)) < 0.1)
            else:
                coef = np.zeros((h, w)); coef[-1, -1] = 20000                # a single level at the very end of the scan
            coef = np.ascontiguousarray(coef.astype(np.int32).reshape(-1))
            ri, luma = int(rng.integers(0, 6)), int(rng.integers(0, 2))
            lv = np.zeros(n, np.int32)
            sums.append(O.orc_depquant(p(coef), p(lv), w, h, luma, 10, qp, C.c_double(lam), C.c_void_p(rates.ctypes.data + ri * ops.DQ_RATES.itemsize)))
            descs.append((off, off, lam, qp, ri, w, h, luma, (0, 0, 0)))
            coefs.append(coef); wants.append(lv); off += n
    d = np.array(descs, dtype=ops.DEPQUANT_DESC)
    level = torch.full((off,), 9, dtype=torch.int32, device="cuda")
    got_sum = ops.depquant_batch(dev(np.concatenate(coefs)), level, ops.struct_to_device(d), len(d), ops.struct_to_device(rates), off, 10)
    assert np.array_equal(got_sum.cpu().numpy().view(np.uint32), np.array(sums, np.uint32))
    assert np.array_equal(level.cpu().numpy(), np.concatenate(wants))


def test_depquant_many_rate_tables():
    """more distinct rate tables among 64 consecutive TUs than the kernel stages in LDS at a time (16): the workgroup walks its TUs in several
    passes; every TU must still be served, with its own table"""
    import ctypes as C
    import os
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(17)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "depquant.npz"))
    rates = np.ascontiguousarray(g["rates"]).view(ops.DQ_RATES)
    nt = len(rates)
    assert nt > 40
    O = oracle()
    O.orc_depquant.restype = C.c_uint32
    descs, coefs, wants, sums = [], [], [], []
    off = 0
    shapes = [(8, 8), (4, 4), (16, 8), (8, 16), (16, 16)]
    for i in range(150):
        w, h = shapes[i % len(shapes)]
        n = w * h
        yy, xx = np.mgrid[0:h, 0:w]
        coef = np.ascontiguousarray((rng.normal(0, 1800, (h, w)) * np.exp(-(xx / w * 2 + yy / h * 2))).astype(np.int32).reshape(-1))
        ri = int((i * 37) % nt)                                           # 64 consecutive TUs name > 40 different tables, slots collide
        qp, lam, luma = int(rng.integers(10, 50)), float(rng.choice([5.0, 80.0, 900.0])), int(i % 3 != 0)
        lv = np.zeros(n, np.int32)
        sums.append(O.orc_depquant(p(coef), p(lv), w, h, luma, 10, qp, C.c_double(lam), C.c_void_p(rates.ctypes.data + ri * ops.DQ_RATES.itemsize)))
        descs.append((off, off, lam, qp, ri, w, h, luma, (0, 0, 0)))
        coefs.append(coef); wants.append(lv); off += n
    d = np.array(descs, dtype=ops.DEPQUANT_DESC)
    level = torch.full((off,), 9, dtype=torch.int32, device="cuda")
    got_sum = ops.depquant_batch(dev(np.concatenate(coefs)), level, ops.struct_to_device(d), len(d), ops.struct_to_device(rates), off, 10)
    assert np.array_equal(got_sum.cpu().numpy().view(np.uint32), np.array(sums, np.uint32))
    assert np.array_equal(level.cpu().numpy(), np.concatenate(wants))


def test_depquant_full_size():
    """bench picture size: every 16x16 TU of a 3840x2160 picture (32400 TUs, 8.3 M coefficients) in one launch.  Size-independent
    properties: abs_sum[i] == sum |level| of TU i; levels keep the sign of their coefficient; a TU whose coefficients are all
    below the trellis threshold stays zero; a random subset equals the oracle."""
    import ctypes as C
    import os
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(123)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "depquant.npz"))
    rates = np.ascontiguousarray(g["rates"][:4]).view(ops.DQ_RATES)
    B, n = 16, (3840 // 16) * (2160 // 16)
    yy, xx = np.mgrid[0:B, 0:B]
    decay = np.exp(-(xx / B * 3 + yy / B * 3)).reshape(-1)
    coef = (rng.normal(0, 1500, (n, B * B)) * decay).astype(np.int32)
    coef[::7] = rng.integers(-2, 3, (coef[::7].shape))                  # every seventh TU: nothing above the threshold
    d = np.zeros(n, ops.DEPQUANT_DESC)
    d["coeff_off"] = d["level_off"] = np.arange(n) * B * B
    d["lambda"], d["qp"], d["rates_idx"], d["w"], d["h"], d["luma"] = 60.0, 44, rng.integers(0, 4, n), B, B, 1
    level = torch.full((n * B * B,), 5, dtype=torch.int32, device="cuda")
    sums = ops.depquant_batch(dev(coef.reshape(-1)), level, ops.struct_to_device(d), n, ops.struct_to_device(rates), n * B * B, 10)
    lv = level.cpu().numpy().reshape(n, B * B)
    s = sums.cpu().numpy().view(np.uint32)
    assert np.array_equal(np.abs(lv).sum(1).astype(np.uint32), s)
    assert np.all((lv == 0) | ((lv < 0) == (coef < 0)))                   # a zero coefficient may still get a (positive) level
    assert np.all(lv[::7] == 0) and np.count_nonzero(lv) > n * 20
    O = oracle()
    O.orc_depquant.restype = C.c_uint32
    for i in rng.choice(n, 60, replace=False):
        want = np.zeros(B * B, np.int32)
        ws = O.orc_depquant(p(np.ascontiguousarray(coef[i])), p(want), B, B, 1, 10, 44, C.c_double(60.0),
                            C.c_void_p(rates.ctypes.data + int(d["rates_idx"][i]) * ops.DQ_RATES.itemsize))
        assert ws == s[i] and np.array_equal(want, lv[i]), i


def test_rdoq():
    """N1: the rate-distortion optimised quantiser on the device.  (1) golden vectors of the compiled reference's QuantRDOQ::quant (luma and
    chroma, sign hiding on and off, per bit depth one launch with mixed TU shapes, four TUs per wavefront); (2) random TUs incl. all-zero
    blocks, full-range coefficients, lambda extremes and a single level at the end of the scan against the oracle."""
    import ctypes as C
    from vvcsoftware_vtm_amd import ops
    from test_oracle_golden import rdoq_rows
    rows = list(rdoq_rows())
    g = rows[0][-1]
    rates_dev = ops.struct_to_device(np.ascontiguousarray(g["rates"]).view(ops.RDOQ_RATES))
    coef_dev = dev(np.ascontiguousarray(g["coef"]))
    total = g["coef"].size
    for bd in (8, 10):
        sel = [r for r in rows if r[3] == bd]
        d = np.array([(r[5], r[5], r[8], r[4], r[7], r[0], r[1], 1 - r[2], r[9], (0, 0)) for r in sel], dtype=ops.RDOQ_DESC)
        level = torch.full((total,), 9, dtype=torch.int32, device="cuda")
        sums = ops.rdoq_batch(coef_dev, level, ops.struct_to_device(d), len(d), rates_dev, total, bd).cpu().numpy().view(np.uint32)
        got = level.cpu().numpy()
        for i, r in enumerate(sel):
            n = r[0] * r[1]
            assert np.array_equal(got[r[5]:r[5] + n], g["level"][r[5]:r[5] + n]), r[:8]
            assert int(sums[i]) == r[6]
    rng = np.random.default_rng(91)
    O = oracle()
    O.orc_rdoq.restype = C.c_uint32
    rates = np.ascontiguousarray(g["rates"][:8]).view(ops.RDOQ_RATES)
    descs, coefs, wants, sums = [], [], [], []
    off = 0
    for (w, h) in [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 32), (32, 4), (8, 64), (64, 8), (16, 32), (4, 64), (64, 4)]:
        for it in range(15 if w * h <= 256 else (5 if w * h <= 1024 else 2)):
            n = w * h
            qp = int(rng.integers(0, 63))
            lam = float([0.7, 30.0, 250.0, 4000.0][it % 4])
            yy, xx = np.mgrid[0:h, 0:w]
            decay = np.exp(-(xx / w * 2.5 + yy / h * 2.5))
            kind = it % 5
            if kind == 0:
                coef = rng.normal(0, 2500, (h, w)) * decay
            elif kind == 1:
                coef = rng.integers(-3, 4, (h, w))                          # nothing quantises to a level at most QPs
            elif kind == 2:
                coef = rng.integers(-32768, 32768, (h, w))                  # full 16-bit range everywhere
            elif kind == 3:
                coef = rng.normal(0, 9000, (h, w)) * (rng.random((h, w)) < 0.1)
            else:
                coef = np.zeros((h, w)); coef[-1, -1] = 20000                # a single level at the very end of the scan
            coef = np.ascontiguousarray(coef.astype(np.int32).reshape(-1))
            ri, luma, sbh = int(rng.integers(0, 8)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
            lv = np.zeros(n, np.int32)
            sums.append(O.orc_rdoq(p(coef), p(lv), w, h, luma, 10, qp, C.c_double(lam), sbh, C.c_void_p(rates.ctypes.data + ri * ops.RDOQ_RATES.itemsize)))
            descs.append((off, off, lam, qp, ri, w, h, luma, sbh, (0, 0)))
            coefs.append(coef); wants.append(lv); off += n
    d = np.array(descs, dtype=ops.RDOQ_DESC)
    level = torch.full((off,), 9, dtype=torch.int32, device="cuda")
    got_sum = ops.rdoq_batch(dev(np.concatenate(coefs)), level, ops.struct_to_device(d), len(d), ops.struct_to_device(rates), off, 10)
    got = level.cpu().numpy()
    want = np.concatenate(wants)
    bad = [i for i, r in enumerate(descs) if not np.array_equal(got[r[0]:r[0] + r[5] * r[6]], want[r[0]:r[0] + r[5] * r[6]])]
    assert not bad, [descs[i][2:] for i in bad[:6]]
    assert np.array_equal(got_sum.cpu().numpy().view(np.uint32), np.array(sums, np.uint32))
