"""GPU parity: intra sample prediction (next row N4, vvcgpu_intra_pred_batch) vs the CPU oracle and the golden vectors of
the compiled reference's own IntraPrediction::predIntraAng."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oraclelib import oracle, p

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
SHAPES = [(w, h) for w in (4, 8, 16, 32, 64) for h in (4, 8, 16, 32, 64)]


def dev(a):
    return torch.from_numpy(a).cuda()


def test_intra_pred_golden():
    from vvcsoftware_vtm_amd import ops
    g = np.load(os.path.join(G, "intra.npz"))
    rows, refs_all, want = g["rows"], np.ascontiguousarray(g["refs"]), g["pred"]
    for bd in (8, 10):
        sel = [r for r in rows if r[3] == bd]
        d = np.zeros(len(sel), ops.INTRA_DESC)
        for i, (w, h, mode, _, filt, T, L, ro, po) in enumerate(sel):
            assert ops.intra_ref_lengths(int(w), int(h)) == (T, L)
            d[i] = (ro, po, w, w, h, mode, filt, 0, 0)
        out = torch.zeros(want.size, dtype=torch.int16, device="cuda")
        ops.intra_pred_batch(dev(refs_all), out, ops.struct_to_device(d), len(d), clp=(0, (1 << bd) - 1))
        got = out.cpu().numpy()
        for (w, h, mode, _, filt, T, L, ro, po) in sel:
            assert np.array_equal(got[po:po + w * h], want[po:po + w * h]), (w, h, mode, bd, filt)


@pytest.mark.parametrize("bd,kind", [(8, "uniform"), (10, "ramp"), (10, "extreme")])
def test_intra_pred_all_shapes_modes(bd, kind):
    """all 25 block shapes x 67 modes x {unfiltered, filtered} in ONE launch, destination blocks scattered in a plane with
    a stride wider than the block, a narrowed clip range (PDPC clips, the plain predictors do not)."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(bd * 31 + len(kind))
    O = oracle()
    mx = (1 << bd) - 1
    clp = (16, mx - 20)
    descs, refs_all, wants = [], [], []
    roff = poff = 0
    for (w, h) in SHAPES:
        T, L = ops.intra_ref_lengths(w, h)
        for mode in range(67):
            for filt in (0, 1):
                if kind == "uniform":
                    refs = rng.integers(0, mx + 1, T + L + 1).astype(np.int16)
                elif kind == "ramp":
                    refs = np.clip(np.cumsum(rng.integers(-5, 8, T + L + 1)) + mx // 3, 0, mx).astype(np.int16)
                else:
                    refs = rng.choice(np.array([0, mx], np.int16), T + L + 1)
                src = refs
                if filt:
                    src = np.zeros_like(refs)
                    O.orc_intra_filter_refs(p(refs), p(src), w, h)
                stride = w + 8
                want = np.full((h, stride), -1, np.int16)
                O.orc_intra_pred(p(src), p(want), stride, w, h, mode, clp[0], clp[1])
                descs.append((roff, poff, stride, w, h, mode, filt, 0, 0))
                refs_all.append(refs); wants.append(want.reshape(-1))
                roff += refs.size; poff += want.size
    d = np.array(descs, ops.INTRA_DESC)
    want = np.concatenate(wants)
    out = torch.full((want.size,), -1, dtype=torch.int16, device="cuda")
    ops.intra_pred_batch(dev(np.concatenate(refs_all)), out, ops.struct_to_device(d), len(d), clp=clp)
    got = out.cpu().numpy()
    if not np.array_equal(got, want):
        bad = np.nonzero(got != want)[0][0]
        k = int(np.searchsorted(d["dst_off"], bad, side="right")) - 1
        raise AssertionError("first mismatch in desc %s" % (d[k],))


def test_intra_pred_full_picture_property():
    """size-independent property at bench scale: 3840x2160 tiled with 16x16 blocks whose reference samples are all equal to
    one value c per block -> every mode must reproduce c exactly (planar, DC, angular, PDPC are all affine combinations with
    weights summing to one), 32400 blocks x a mode each in one launch."""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(12)
    W, H, B = 3840, 2160, 16
    T, L = ops.intra_ref_lengths(B, B)
    nb = (W // B) * (H // B)
    vals = rng.integers(0, 1024, nb).astype(np.int16)
    refs = np.repeat(vals, T + L + 1)
    d = np.zeros(nb, ops.INTRA_DESC)
    bx, by = np.meshgrid(np.arange(W // B), np.arange(H // B))
    d["ref_off"] = np.arange(nb) * (T + L + 1)
    d["dst_off"] = (by.ravel() * B) * W + bx.ravel() * B
    d["dst_stride"], d["w"], d["h"] = W, B, B
    d["mode"] = rng.integers(0, 67, nb)
    d["filter_refs"] = rng.integers(0, 2, nb)
    out = torch.zeros((H, W), dtype=torch.int16, device="cuda")
    ops.intra_pred_batch(dev(refs), out, ops.struct_to_device(d), nb)
    got = out.cpu().numpy()
    want = np.repeat(np.repeat(vals.reshape(H // B, W // B), B, axis=0), B, axis=1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_satd_batch_equals_predict_then_hadamard(bd):
    """intra mode pre-selection (IntraSearch.cpp:397-480): the fused entry against the oracle's predIntraAng followed by its xGetHADs, every block
    shape 4..64 (squares and rectangles), every mode, filtered and unfiltered references"""
    from vvcsoftware_vtm_amd import ops
    rng = np.random.default_rng(90 + bd)
    mx = (1 << bd) - 1
    shapes = [(w, h) for w in (4, 8, 16, 32, 64) for h in (4, 8, 16, 32, 64)]
    rows, refs_all, roff, ooff = [], [], 0, 0
    for (w, h) in shapes:
        T, L = ops.intra_ref_lengths(w, h)
        for mode in list(range(0, 67, 3)) + [1, 2, 18, 34, 50, 66]:
            base = int(rng.integers(0, mx))
            refs_all.append(np.clip(base + rng.integers(-60, 61, T + L + 1), 0, mx).astype(np.int16))
            rows.append((roff, ooff, w + 8, w, h, mode, int(rng.integers(0, 2)), 0, 0))
            roff += T + L + 1
            ooff += (w + 8) * h
    sd = np.array(rows, dtype=ops.INTRA_SATD_DESC)
    refs = np.concatenate(refs_all)
    org = rng.integers(0, mx + 1, ooff).astype(np.int16)
    # oracle: predict every candidate into its own buffer, then the Hadamard distortion of (org, pred)
    O = oracle()
    pred = np.zeros(ooff, np.int16)
    for i, r in enumerate(sd):
        w, h = int(r["w"]), int(r["h"])
        src = refs_all[i]
        if r["filter_refs"]:
            src = np.zeros_like(refs_all[i])
            O.orc_intra_filter_refs(p(refs_all[i]), p(src), w, h)
        blk = np.zeros((h, w + 8), np.int16)
        O.orc_intra_pred(p(src), p(blk), w + 8, w, h, int(r["mode"]), 0, mx)
        pred[int(r["org_off"]):int(r["org_off"]) + blk.size] = blk.reshape(-1)
    dd = np.zeros(len(sd), ops.DIST_DESC)
    dd["org_off"] = dd["cur_off"] = sd["org_off"]
    dd["org_stride"] = dd["cur_stride"] = sd["org_stride"]
    dd["w"], dd["h"] = sd["w"], sd["h"]
    want = np.zeros(len(sd), np.uint64)
    oracle().orc_dist_batch(1, p(org), p(pred), p(dd), len(dd), p(want))
    got = ops.intra_satd_batch(dev(refs), dev(org), ops.struct_to_device(sd), len(sd), clp=(0, mx))
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want)


def test_cclm_golden_and_random():
    """CCLM (vvcgpu_cclm_pred_batch): (1) the 938 blocks captured from the reference's own predIntraChromaLM, all in one launch;
    (2) random luma / neighbours for every chroma shape x availability combination against the oracle."""
    from vvcsoftware_vtm_amd import ops
    from test_oracle_golden import cclm_records
    for bd in (8, 10):
        recs = [r for r in cclm_records() if r[5] == bd]
        assert recs
        d = np.zeros(len(recs), ops.CCLM_DESC)
        lo = no = po = 0
        wins, nbs, wants = [], [], []
        for i, (w, h, above, left, bdl, bdc, cmin, cmax, lw, lh, win, nb, want) in enumerate(recs):
            d[i] = (lo + 2 * lw + 3, no, po, lw, w, w, h, above, left, 0, (0, 0))
            wins.append(win); nbs.append(nb); wants.append(want)
            lo += win.size; no += nb.size; po += want.size
            clp = (cmin, cmax)
        out = torch.zeros(po, dtype=torch.int16, device="cuda")
        ops.cclm_pred_batch(dev(np.concatenate(wins)), dev(np.concatenate(nbs)), out, ops.struct_to_device(d), len(d), bd, bd, clp)
        assert np.array_equal(out.cpu().numpy(), np.concatenate(wants))
    rng = np.random.default_rng(8)
    O = oracle()
    bd, clp = 10, (4, 1000)
    descs, lum, nbs, wants = [], [], [], []
    lo = no = po = 0
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            for above in (0, 1):
                for left in (0, 1):
                    lw, lh = 2 * w + 3, 2 * h + 2
                    kind = (w + h + above) % 3
                    if kind == 0:
                        win = rng.integers(0, 1024, lw * lh).astype(np.int16)
                        nb = rng.integers(0, 1024, w + h).astype(np.int16)
                    elif kind == 1:                       # correlated chroma = 0.6 luma + 200: a real linear model
                        win = np.clip(rng.normal(500, 120, lw * lh), 0, 1023).astype(np.int16)
                        nb = np.clip(0.6 * rng.normal(500, 120, w + h) + 200, 0, 1023).astype(np.int16)
                    else:
                        win = rng.choice(np.array([0, 1023], np.int16), lw * lh)
                        nb = rng.choice(np.array([0, 1023], np.int16), w + h)
                    want = np.zeros((h, w), np.int16)
                    O.orc_cclm_pred(C.c_void_p(win.ctypes.data + (2 * lw + 3) * 2), lw, p(nb), C.c_void_p(nb.ctypes.data + 2 * w), p(want), w, w, h,
                                    above, left, bd, bd, clp[0], clp[1])
                    descs.append((lo + 2 * lw + 3, no, po, lw, w, w, h, above, left, 0, (0, 0)))
                    lum.append(win); nbs.append(nb); wants.append(want.reshape(-1))
                    lo += win.size; no += nb.size; po += want.size
    d = np.array(descs, ops.CCLM_DESC)
    out = torch.zeros(po, dtype=torch.int16, device="cuda")
    ops.cclm_pred_batch(dev(np.concatenate(lum)), dev(np.concatenate(nbs)), out, ops.struct_to_device(d), len(d), bd, bd, clp)
    got, want = out.cpu().numpy(), np.concatenate(wants)
    if not np.array_equal(got, want):
        bad = np.nonzero(got != want)[0][0]
        k = int(np.searchsorted(d["dst_off"], bad, side="right")) - 1
        raise AssertionError("first mismatch in desc %s" % (d[k],))


def test_intra_fill_refs_golden_and_random():
    """xFillReferenceSamples on the device: (1) the 1584 calls captured from the reference, one launch; (2) random availability
    patterns (none, all, random runs) for every block shape and both unit sizes against the oracle; (3) chained with
    vvcgpu_intra_pred_batch: reconstruction + flags -> prediction, equal to the oracle's fill -> (filter) -> predict."""
    from vvcsoftware_vtm_amd import ops
    from test_oracle_golden import intra_fill_records
    O = oracle()
    for bd in (8, 10):
        recs = [r for r in intra_fill_records() if r[4] == bd]
        d = np.zeros(len(recs), ops.INTRA_FILL_DESC)
        planes, flags_all, wants = [], [], []
        po = fo = ro = 0
        for i, (w, h, uw, uh, _, T, L, flags, plane, want) in enumerate(recs):
            d[i] = (po + plane.shape[1] + 1, fo, ro, plane.shape[1], w, h, uw, uh, 0, 0)
            planes.append(plane.reshape(-1)); flags_all.append(flags); wants.append(want)
            po += plane.size; fo += flags.size; ro += want.size
        out = torch.zeros(ro, dtype=torch.int16, device="cuda")
        ops.intra_fill_refs_batch(dev(np.concatenate(planes)), dev(np.concatenate(flags_all)), out, ops.struct_to_device(d), len(d), bd)
        assert np.array_equal(out.cpu().numpy(), np.concatenate(wants))

    rng = np.random.default_rng(21)
    bd = 10
    W, H = 512, 384
    rec = rng.integers(0, 1024, (H, W)).astype(np.int16)
    fdescs, flags_all, wants, pdescs, pwants = [], [], [], [], []
    fo = ro = po = 0
    for (w, h) in SHAPES:
        T, L = ops.intra_ref_lengths(w, h)
        for unit in (4, 2):
            aboveUnits, leftUnits = (T + unit - 1) // unit, (L + unit - 1) // unit
            total = aboveUnits + leftUnits + 1
            for pat in range(6):
                if pat == 0:
                    flags = np.zeros(total, np.uint8)
                elif pat == 1:
                    flags = np.ones(total, np.uint8)
                elif pat == 2:                               # only something far along the chain is available
                    flags = np.zeros(total, np.uint8); flags[int(rng.integers(total // 2, total))] = 1
                else:                                        # random runs
                    flags = np.repeat(rng.integers(0, 2, total).astype(np.uint8), int(rng.integers(1, 5)))[:total].copy()
                x0, y0 = int(rng.integers(140, W - 200)), int(rng.integers(140, H - 200))
                want = np.zeros(T + L + 1, np.int16)
                O.orc_intra_fill_refs(C.c_void_p(rec.ctypes.data + (y0 * W + x0) * 2), W, p(flags), p(want), w, h, unit, unit, bd)
                fdescs.append((y0 * W + x0, fo, ro, W, w, h, unit, unit, 0, 0))
                mode, filt = int(rng.integers(0, 67)), int(rng.integers(0, 2))
                src = want
                if filt:
                    src = np.zeros_like(want); O.orc_intra_filter_refs(p(want), p(src), w, h)
                pw = np.zeros((h, w), np.int16)
                O.orc_intra_pred(p(src), p(pw), w, w, h, mode, 0, 1023)
                pdescs.append((ro, po, w, w, h, mode, filt, 0, 0))
                flags_all.append(flags); wants.append(want); pwants.append(pw.reshape(-1))
                fo += total; ro += want.size; po += pw.size
    fd, pd = np.array(fdescs, ops.INTRA_FILL_DESC), np.array(pdescs, ops.INTRA_DESC)
    refs = torch.zeros(ro, dtype=torch.int16, device="cuda")
    pred = torch.zeros(po, dtype=torch.int16, device="cuda")
    ops.intra_fill_refs_batch(dev(rec), dev(np.concatenate(flags_all)), refs, ops.struct_to_device(fd), len(fd), bd)
    ops.intra_pred_batch(refs, pred, ops.struct_to_device(pd), len(pd))
    assert np.array_equal(refs.cpu().numpy(), np.concatenate(wants))
    assert np.array_equal(pred.cpu().numpy(), np.concatenate(pwants))
