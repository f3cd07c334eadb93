"""vvcgpu_resi_chain_batch (residual -> T1 -> Quant::quant -> Quant::dequant -> T2 -> reconstruction in one pass) against the oracle's
restatement of the five separate steps (orc_pelop_batch / orc_tr_fwd_batch / orc_quant_batch / orc_dequant_tr_inv_batch / orc_pelop_batch),
each of which is pinned against the compiled reference (tests/test_oracle_golden.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
from oraclelib import oracle, p

pytestmark = pytest.mark.gpu

TR_DESC = np.dtype([("resi_off", "<i8"), ("coeff_off", "<i8"), ("resi_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                    ("tr_hor", "i1"), ("tr_ver", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
QUANT_DESC = np.dtype([("coeff_off", "<i8"), ("level_off", "<i8"), ("w", "<i2"), ("h", "<i2"), ("intra_slice", "i1"), ("sign_hiding", "i1"),
                       ("reserved", "<i2"), ("qp", "<i4"), ("reserved2", "<i4")])
DQTR_DESC = np.dtype([("resi_off", "<i8"), ("level_off", "<i8"), ("resi_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                      ("tr_hor", "i1"), ("tr_ver", "i1"), ("dep_quant", "i1"), ("reserved", "i1"), ("qp", "<i4")])
PELOP_DESC = np.dtype([("src0_off", "<i8"), ("src1_off", "<i8"), ("dst_off", "<i8"), ("src0_stride", "<i4"),
                       ("src1_stride", "<i4"), ("dst_stride", "<i4"), ("w", "<i2"), ("h", "<i2")])


class PelopCfg(C.Structure):
    _fields_ = [("scale", C.c_int32), ("shift", C.c_int32), ("offset", C.c_int32), ("clip", C.c_int32),
                ("clp_min", C.c_int32), ("clp_max", C.c_int32)]


def oracle_chain(org, pred, tus, bd, W):
    """tus: list of (x, y, w, h, tr_hor, tr_ver, qp, intra, sbh) -> (levels, abs_sum, rec)"""
    H = org.shape[0]
    mx = (1 << bd) - 1
    n = len(tus)
    tr = np.zeros(n, TR_DESC); qd = np.zeros(n, QUANT_DESC); dq = np.zeros(n, DQTR_DESC); bands = np.zeros(n, PELOP_DESC)
    coff = 0
    for i, (x, y, w, h, th, tv, qp, intra, sbh) in enumerate(tus):
        tr[i] = (y * W + x, coff, W, w, h, th, tv, 0, 0)
        qd[i] = (coff, coff, w, h, intra, sbh, 0, qp, 0)
        dq[i] = (y * W + x, coff, W, w, h, th, tv, 0, 0, qp)
        bands[i] = (y * W + x, y * W + x, y * W + x, W, W, W, w, h)
        coff += w * h
    resi = np.zeros((H, W), np.int16); resi2 = np.zeros((H, W), np.int16)
    coef = np.zeros(coff, np.int32); level = np.zeros(coff, np.int32); dqc = np.zeros(coff, np.int32); abs_sum = np.zeros(n, np.uint32)
    rec = pred.copy()
    o = oracle()
    o.orc_pelop_batch(3, p(org), p(pred), p(resi), p(bands), n, C.byref(PelopCfg(0, 0, 0, 0, 0, mx)))
    o.orc_tr_fwd_batch(p(resi), p(coef), p(tr), n, bd)
    o.orc_quant_batch(p(coef), p(level), p(qd), n, bd, p(abs_sum))
    o.orc_dequant_tr_inv_batch(p(level), p(resi2), p(dq), n, bd, p(dqc))
    o.orc_pelop_batch(1, p(pred), p(resi2), p(rec), p(bands), n, C.byref(PelopCfg(0, 0, 0, 1, 0, mx)))
    return level, abs_sum, rec, tr["coeff_off"].copy()


def gpu_chain(org, pred, tus, bd, W, coffs):
    from vvcsoftware_vtm_amd import ops
    n = len(tus)
    d = np.zeros(n, ops.RC_DESC)
    for i, (x, y, w, h, th, tv, qp, intra, sbh) in enumerate(tus):
        d[i] = (y * W + x, y * W + x, y * W + x, coffs[i], W, W, W, w, h, th, tv, intra, sbh, qp, (0, 0))
    total = int(sum(t[2] * t[3] for t in tus))
    dorg, dpred = torch.from_numpy(org).cuda(), torch.from_numpy(pred).cuda()
    drec = dpred.clone()
    dlevel = torch.full((total,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    a = ops.resi_chain_batch(dorg, dpred, drec, dlevel, ops.struct_to_device(d), n, bd, (0, (1 << bd) - 1))
    torch.cuda.synchronize()
    return dlevel.cpu().numpy(), a.cpu().numpy().view(np.uint32), drec.cpu().numpy()


def tile(W, H, shapes, rng, qps, bd, types_small=True):
    """non-overlapping TUs: the plane is cut into 64x64 cells, each cell tiled with one shape"""
    tus = []
    ci = 0
    for y0 in range(0, H, 64):
        for x0 in range(0, W, 64):
            w, h = shapes[ci % len(shapes)]
            ci += 1
            for ty in range(0, 64, h):
                for tx in range(0, 64, w):
                    th = int(rng.integers(0, 3)) if w <= 32 and w >= 4 else 0
                    tv = int(rng.integers(0, 3)) if h <= 32 and h >= 4 else 0
                    tus.append((x0 + tx, y0 + ty, w, h, th, tv, int(rng.choice(qps)) + 6 * (bd - 8), int(rng.integers(0, 2)), int(rng.integers(0, 4) != 0)))
    return tus


@pytest.mark.parametrize("bd,content", [(10, "smooth"), (10, "uniform"), (8, "smooth"), (10, "extreme")])
def test_resi_chain_squares(bd, content):
    """the shapes of the canonical workload (64 ... 4 squared, every transform pair, QP 22..37, sign hiding on / off, both slice types)"""
    rng = np.random.default_rng(bd * 5 + len(content))
    W, H = 448, 192
    org = cases.rand_plane(rng, H, W, bd, content)
    pred = cases.rand_plane(rng, H, W, bd, "smooth" if content != "extreme" else "extreme")
    if content == "smooth":                      # small residual: many zero levels, sparse coefficient groups (last-group logic)
        pred = np.clip(org + rng.integers(-6, 7, org.shape), 0, (1 << bd) - 1).astype(np.int16)
    tus = tile(W, H, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], rng, [22, 27, 32, 37, 45], bd)
    lv, asum, rec, coffs = oracle_chain(org, pred, tus, bd, W)
    glv, gsum, grec = gpu_chain(org, pred, tus, bd, W, coffs)
    assert np.array_equal(gsum, asum)
    assert np.array_equal(glv, lv)
    assert np.array_equal(grec, rec)


@pytest.mark.parametrize("W,H,shapes,per_wave", [(1536, 1024, [(4, 4), (16, 16), (4, 4), (16, 16), (32, 32)], 1),
                                                 (2048, 1024, [(4, 4), (4, 4), (4, 4), (16, 16)], 2),
                                                 (2048, 1024, [(8, 8), (8, 8), (16, 16)], 8),
                                                 (2560, 1024, [(4, 4), (4, 4), (4, 4), (16, 16)], 2)])
def test_resi_chain_prologue_items(W, H, shapes, per_wave):
    """lists long enough for the chain launch's prologue (one wave of a workgroup copies the matrix image, the other three run one / two items of the
    4x4 class -- or one of the 8x8 class -- taken from the end of that class's list): every TU served exactly once, results as the oracle's"""
    rng = np.random.default_rng(W + per_wave)
    bd = 10
    org = cases.rand_plane(rng, H, W, bd, "smooth")
    pred = np.clip(org + rng.integers(-25, 26, org.shape), 0, 1023).astype(np.int16)
    tus = tile(W, H, shapes, rng, [22, 32, 37], bd)
    items4 = (sum(1 for t in tus if t[2] == 4 and t[3] == 4) + 15) // 16
    items8 = (sum(1 for t in tus if t[2] == 8 and t[3] == 8) + 7) // 8
    if per_wave == 8:                                                              # no 4x4 TUs: one 8x8 item per wave
        assert items4 == 0 and items8 >= 3 * 768
    else:
        assert (items4 >= 6 * 768) == (per_wave == 2) and items4 >= 3 * 768      # (768 workgroups: the launch's grid)
    lv, asum, rec, coffs = oracle_chain(org, pred, tus, bd, W)
    glv, gsum, grec = gpu_chain(org, pred, tus, bd, W, coffs)
    assert np.array_equal(gsum, asum)
    assert np.array_equal(glv, lv)
    assert np.array_equal(grec, rec)


def test_resi_chain_rectangles_and_chroma_shapes():
    """every other W x H in 2..64 takes the generic path: rectangles with 2:1 ... 16:1 aspect, 2-wide chroma TUs"""
    rng = np.random.default_rng(77)
    bd, W, H = 10, 512, 128
    org = cases.rand_plane(rng, H, W, bd, "smooth")
    pred = np.clip(org + rng.integers(-40, 41, org.shape), 0, 1023).astype(np.int16)
    shapes = [(64, 32), (32, 64), (16, 32), (32, 16), (64, 16), (8, 16), (16, 8), (4, 16), (4, 64), (64, 4), (16, 4), (4, 8), (8, 4), (2, 8), (8, 2), (2, 2), (16, 64), (32, 8), (2, 32)]
    tus = tile(W, H, shapes, rng, [27, 32, 37], bd)
    lv, asum, rec, coffs = oracle_chain(org, pred, tus, bd, W)
    glv, gsum, grec = gpu_chain(org, pred, tus, bd, W, coffs)
    assert np.array_equal(gsum, asum)
    assert np.array_equal(glv, lv)
    assert np.array_equal(grec, rec)


@pytest.mark.parametrize("bd,content", [(10, "smooth"), (8, "smooth"), (10, "extreme"), (10, "uniform")])
def test_resi_chain_packed_tiles(bd, content):
    """16x8 / 8x16 / 16x4 / 4x16 (and 32x8 .. 4x32, 64x8 .. 4x64 in multi-tiles, the 64-point side with its zero-out): two or four TUs share one 16x16 matrix-core tile (rc_tile_packed*),
    every TU with its own transform pair, QP, slice type
    and sign-hiding flag; class counts that leave the last tile part-filled; a few TUs whose residual leaves +-1023 sit in tiles with ordinary
    ones and must reach the generic path alone"""
    rng = np.random.default_rng(bd * 11 + len(content))
    W, H = 896, 128
    org = cases.rand_plane(rng, H, W, bd, content)
    pred = cases.rand_plane(rng, H, W, bd, "smooth" if content != "extreme" else "extreme")
    if content == "smooth":
        pred = np.clip(org + rng.integers(-9, 10, org.shape), 0, (1 << bd) - 1).astype(np.int16)
    tus = tile(W, H, [(16, 8), (8, 16), (16, 4), (4, 16), (16, 16), (8, 8), (32, 8), (8, 32), (32, 4), (4, 32), (64, 8), (8, 64), (64, 4), (4, 64)], rng, [22, 27, 32, 37, 45], bd)
    keep = rng.random(len(tus)) > 0.07                       # odd counts per class, tiles whose TUs are not neighbours in the picture
    tus = [t for t, k in zip(tus, keep) if k]
    order = rng.permutation(len(tus))
    tus = [tus[i] for i in order]
    if content == "smooth" and bd == 10:
        org = org.copy()
        for i in rng.choice(len(tus), 25, replace=False):    # out of the matrix-core range: |org - pred| > 1023
            x, y, w, h = tus[i][:4]
            org[y + int(rng.integers(0, h)), x + int(rng.integers(0, w))] = -2500
    lv, asum, rec, coffs = oracle_chain(org, pred, tus, bd, W)
    glv, gsum, grec = gpu_chain(org, pred, tus, bd, W, coffs)
    assert np.array_equal(gsum, asum), np.nonzero(gsum != asum)[0][:8]
    assert np.array_equal(glv, lv)
    assert np.array_equal(grec, rec)


def test_resi_chain_residual_outside_range_falls_back():
    """samples outside the bit depth (|residual| > 1023) leave the matrix-core path's exactness range: those TUs are served by the generic path"""
    rng = np.random.default_rng(5)
    bd, W, H = 10, 192, 64
    org = rng.integers(-3000, 3000, (H, W)).astype(np.int16)
    pred = rng.integers(0, 1024, (H, W)).astype(np.int16)
    tus = tile(W, H, [(64, 64), (32, 32), (16, 16)], rng, [32], bd)
    lv, asum, rec, coffs = oracle_chain(org, pred, tus, bd, W)
    glv, gsum, grec = gpu_chain(org, pred, tus, bd, W, coffs)
    assert np.array_equal(gsum, asum) and np.array_equal(glv, lv) and np.array_equal(grec, rec)


def test_resi_chain_rejects_transform_skip_descriptor():
    from vvcsoftware_vtm_amd import ops
    bd, W, H = 10, 64, 64
    org = np.full((H, W), 500, np.int16)
    tus = [(0, 0, 4, 4, 3, 0, 32 + 12, 0, 1), (8, 0, 4, 4, 0, 0, 32 + 12, 0, 1)]
    d = np.zeros(2, ops.RC_DESC)
    for i, (x, y, w, h, th, tv, qp, intra, sbh) in enumerate(tus):
        d[i] = (y * W + x, y * W + x, y * W + x, 16 * i, W, W, W, w, h, th, tv, intra, sbh, qp, (0, 0))
    t = torch.from_numpy(org).cuda()
    rec = t.clone()
    lvl = torch.zeros(32, dtype=torch.int32, device="cuda")
    a = ops.resi_chain_batch(t, t.clone(), rec, lvl, ops.struct_to_device(d), 2, bd).cpu().numpy().view(np.uint32)
    assert a[0] == 0xFFFFFFFF and a[1] == 0


def test_stream_counters_survive_interleaved_entry_points():
    """vvcgpu_resi_chain_batch and vvcgpu_mc_batch take their zeroed work counters from the stream's persistent pair (lib.hip: vvcgpu_counters) and
    each call clears the other set for the next one.  Calls of both kinds in every order, with lists that leave different counts behind
    (a chain with TUs of every class and fall-back TUs, then one with a single class; MC lists with and without PUs for the generic kernel),
    must all give the oracle's results."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(4242)
    bd, W, H = 10, 256, 128
    mx = (1 << bd) - 1
    org = cases.rand_plane(rng, H, W, bd, "uniform")
    pred = np.clip(org + rng.integers(-30, 31, org.shape), 0, mx).astype(np.int16)
    wild = rng.integers(-3000, 3000, (H, W)).astype(np.int16)                       # |residual| > 1023: fall-back list of the chain
    tus_all = tile(W, H, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (16, 8)], rng, [27, 37], bd)
    tus_one = tile(W, H, [(8, 8)], rng, [32], bd)
    want = {}
    for name, (o_, tus) in {"all": (org, tus_all), "one": (org, tus_one), "wild": (wild, tus_all)}.items():
        want[name] = oracle_chain(o_, pred, tus, bd, W)

    # MC lists: ref = a padded plane, 16x16 luma PUs (fast kernel) with and without a few 32x8 PUs (generic kernel)
    M = 16
    ref = np.ascontiguousarray(np.pad(cases.rand_plane(rng, H, W, bd, "smooth"), M, mode="edge"))
    RS = W + 2 * M

    def mc_list(extra):
        rows, dst_off = [], 0
        for y in range(0, H, 16):
            for x in range(0, W // 2, 16):
                rows.append(((y + M) * RS + x + M + int(rng.integers(-3, 4)), 0, dst_off, RS, RS, 16, 16, 16, int(rng.integers(0, 4)), int(rng.integers(0, 4)), 0, 0, 1, 0, 0))
                dst_off += 256
        for k in range(extra):
            rows.append(((M + 8 * k) * RS + M + 64, 0, dst_off, RS, RS, 32, 32, 8, 1 + k % 3, 2, 0, 0, 1, 0, 0))
            dst_off += 256
        d = np.zeros(len(rows), ops.MC_DESC)
        for i, r in enumerate(rows):
            d[i] = r
        return d, dst_off

    def mc_oracle(d, total):
        out = np.zeros(total, np.int16)
        oracle().orc_mc_batch(p(ref), p(ref), p(out), p(d), d.size, bd, 0, mx)
        return out

    mc = {k: mc_list(e) for k, e in (("fast", 0), ("mixed", 5))}
    mc_want = {k: mc_oracle(*v) for k, v in mc.items()}
    dref = torch.from_numpy(ref).cuda()

    def run_chain(name):
        o_, tus = {"all": (org, tus_all), "one": (org, tus_one), "wild": (wild, tus_all)}[name]
        lv, asum, rec, coffs = want[name]
        glv, gsum, grec = gpu_chain(o_, pred, tus, bd, W, coffs)
        assert np.array_equal(gsum, asum) and np.array_equal(glv, lv) and np.array_equal(grec, rec), name

    def run_mc(name):
        d, total = mc[name]
        dst = torch.zeros(total, dtype=torch.int16, device="cuda")
        ops.mc_batch(dref, dref, dst, ops.struct_to_device(d), d.size, bd, (0, mx))
        assert np.array_equal(dst.cpu().numpy(), mc_want[name]), name

    for step in ["all", "mixed", "one", "wild", "wild", "fast", "mixed", "mixed", "all", "one", "fast", "all"]:
        (run_mc if step in mc else run_chain)(step)


@pytest.mark.parametrize("bd,content", [(10, "smooth"), (8, "uniform")])
def test_resi_chain_runs_entry_equals_the_classified_entry(bd, content):
    """vvcgpu_resi_chain_runs_batch: descriptors grouped by shape, runs (w, h, count) from the host -- no classification launch.  Every shape the chain
    has a body for (squares, rectangles, packed tiles), in an order of the caller's choosing, a few TUs outside the matrix-core range (fall-back list), twice on
    one stream with different lists (the identity array and the counter sets persist); equal to the oracle, i.e. to vvcgpu_resi_chain_batch.  A list
    that holds a 2-wide shape is served through the classified path."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(bd * 3 + len(content))
    W, H = 768, 192
    org = cases.rand_plane(rng, H, W, bd, content)
    pred = np.clip(org + rng.integers(-25, 26, org.shape), 0, (1 << bd) - 1).astype(np.int16)
    shapes_all = [(8, 8), (64, 64), (4, 4), (32, 32), (16, 16), (64, 32), (32, 64), (16, 8), (8, 16), (4, 8), (8, 4), (32, 8), (4, 32), (64, 16), (16, 64), (32, 16), (16, 32), (64, 4), (8, 64)]
    for rnd, shapes in enumerate([shapes_all, shapes_all[3:9], shapes_all + [(2, 8)]]):
        tus = tile(W, H, shapes, rng, [22, 32, 37], bd)
        keep = rng.random(len(tus)) > 0.1
        tus = [t for t, k in zip(tus, keep) if k]
        order = sorted(range(len(tus)), key=lambda i: shapes.index((tus[i][2], tus[i][3])))      # grouped by shape, groups in the caller's order
        tus = [tus[i] for i in order]
        runs = []
        for t in tus:
            if runs and runs[-1][0] == t[2] and runs[-1][1] == t[3]:
                runs[-1][2] += 1
            else:
                runs.append([t[2], t[3], 1])
        o2 = org.copy()
        if rnd == 0 and bd == 10:
            for i in rng.choice(len(tus), 9, replace=False):
                x, y = tus[i][:2]
                o2[y, x] = -2500
        lv, asum, rec, coffs = oracle_chain(o2, pred, tus, bd, W)
        n = len(tus)
        d = np.zeros(n, ops.RC_DESC)
        for i, (x, y, w, h, th, tv, qp, intra, sbh) in enumerate(tus):
            d[i] = (y * W + x, y * W + x, y * W + x, coffs[i], W, W, W, w, h, th, tv, intra, sbh, qp, (0, 0))
        dorg, dpred = torch.from_numpy(o2).cuda(), torch.from_numpy(pred).cuda()
        drec = dpred.clone()
        dlevel = torch.full((int(sum(t[2] * t[3] for t in tus)),), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        a = ops.resi_chain_runs_batch(dorg, dpred, drec, dlevel, ops.struct_to_device(d), n, runs, bd, (0, (1 << bd) - 1))
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy().view(np.uint32), asum), rnd
        assert np.array_equal(dlevel.cpu().numpy(), lv), rnd
        assert np.array_equal(drec.cpu().numpy(), rec), rnd


def test_resi_chain_runs_entry_rejects_inconsistent_runs():
    from vvcsoftware_vtm_amd import ops, capi
    t = torch.zeros((64, 64), dtype=torch.int16, device="cuda")
    lvl = torch.zeros(4096, dtype=torch.int32, device="cuda")
    d = np.zeros(4, ops.RC_DESC)
    d["w"] = d["h"] = 8
    dd = ops.struct_to_device(d)
    for runs in ([(8, 8, 3)], [(8, 8, 2), (8, 8, 2)], [(8, 3, 4)]):
        with pytest.raises(capi.VvcGpuError):
            ops.resi_chain_runs_batch(t, t, t.clone(), lvl, dd, 4, runs, 10)
