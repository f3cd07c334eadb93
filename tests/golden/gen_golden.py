#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs + the outputs of the COMPILED REFERENCE (oracle/_ref/libvtmref.so, built
from /root/reference by oracle/Makefile) for every hot-path row.  Fixtures are data only (inputs, parameters, expected
outputs).  Run in the build container; the fixtures travel with the repo and pin the CPU oracle (tests/test_oracle_golden.py)
and, through it, the HIP kernels."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oraclelib import ref, p, SAO_DTYPE  # noqa: E402

R = ref()
R.vtmref_dist.restype = C.c_uint64
R.vtmref_mvcost.restype = C.c_uint64


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(name, os.path.getsize(path), "bytes")


def gen_alf():
    rng = np.random.default_rng(1001)
    out = {}
    for ci, (w, h, ctu, bd, kind) in enumerate([(136, 72, 64, 10, "smooth"), (64, 64, 32, 8, "uniform"), (128, 64, 128, 10, "uniform")]):
        Y, Cb, Cr = (cases.rand_plane(rng, h, w, bd, kind), cases.rand_plane(rng, h // 2, w // 2, bd, kind),
                     cases.rand_plane(rng, h // 2, w // 2, bd, kind))
        lc, cc = cases.alf_coeffs(rng)
        nx, ny = cases.n_ctus(w, h, ctu)
        en = [rng.integers(0, 2, nx * ny).astype(np.uint8) for _ in range(3)]
        en[0][0] = 1
        for ft in (0, 1):
            dY, dCb, dCr = Y.copy(), Cb.copy(), Cr.copy()
            cls = np.zeros((h // 4, w // 4), np.uint16)
            R.vtmref_alf_picture(1, p(Y), p(Cb), p(Cr), p(dY), p(dCb), p(dCr), w, h, ctu, bd, ft, p(lc), p(cc), p(en[0]), p(en[1]),
                                 p(en[2]), p(cls))
            k = "c%d_f%d_" % (ci, ft)
            out.update({k + "dY": dY, k + "dCb": dCb, k + "dCr": dCr, k + "cls": cls})
        # classification for the whole picture (all CTUs on)
        cls = np.zeros((h // 4, w // 4), np.uint16)
        dY, dCb, dCr = Y.copy(), Cb.copy(), Cr.copy()
        R.vtmref_alf_picture(0, p(Y), p(Cb), p(Cr), p(dY), p(dCb), p(dCr), w, h, ctu, bd, 1, p(lc), p(cc), None, None, None, p(cls))
        k = "c%d_" % ci
        out.update({k + "Y": Y, k + "Cb": Cb, k + "Cr": Cr, k + "lc": lc, k + "cc": cc, k + "enY": en[0], k + "enCb": en[1],
                    k + "enCr": en[2], k + "cls_all": cls, k + "meta": np.array([w, h, ctu, bd])})
        # statistics (scalar reference path; exact integers)
        org = cases.rand_plane(rng, h, w, bd, kind)
        if ctu >= 64:
            for ft in (0, 1):
                N = 13 if ft else 7
                st = np.zeros((nx * ny, 25, N * N + N + 1), np.int64)
                R.vtmref_alf_stats(p(org), w, p(Y), w, h, ctu, p(cls), ft, p(st))
                out[k + "stats_f%d" % ft] = st
            out[k + "org"] = org
    save("alf", **out)


def gen_sao():
    rng = np.random.default_rng(1002)
    out = {}
    for ci, (w, h, cw, bd, kind, full) in enumerate([(136, 72, 64, 10, "flat", True), (96, 72, 32, 8, "uniform", False),
                                                     (130, 70, 64, 10, "extreme", False)]):
        mx = (1 << bd) - 1
        Y = cases.rand_plane(rng, h, w, bd, kind)
        prm = cases.sao_params(rng, w, h, cw, cw, full)
        d = Y.copy()
        R.vtmref_sao_apply(p(Y), w, p(d), w, w, h, cw, cw, bd, p(prm), 0, mx)
        org = cases.rand_plane(rng, h, w, bd, kind)
        nx, ny = cases.n_ctus(w, h, cw)
        st = np.zeros((nx * ny, 5, 2, 32), np.int64)
        R.vtmref_sao_stats(0, p(org), w, p(Y), w, w, h, cw, cw, bd, None, 5, 4, p(st))
        k = "c%d_" % ci
        out.update({k + "Y": Y, k + "prm": prm.view(np.uint8), k + "out": d, k + "org": org, k + "stats": st,
                    k + "meta": np.array([w, h, cw, bd])})
    save("sao", **out)


def gen_dist():
    rng = np.random.default_rng(1003)
    rows = []
    W, H = 160, 144
    out = {}
    for bd in (8, 10):
        org = cases.rand_plane(rng, H, W, bd, "smooth")
        cur = cases.rand_plane(rng, H, W, bd, "smooth")
        out["org%d" % bd], out["cur%d" % bd] = org, cur
        for w in (4, 8, 12, 16, 24, 32, 48, 64, 128):
            for h in (4, 8, 16, 32, 64, 128):
                ox, oy, cx, cy = (int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)),
                                  int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)))
                po = C.c_void_p(org.ctypes.data + 2 * (oy * W + ox))
                pc = C.c_void_p(cur.ctypes.data + 2 * (cy * W + cx))
                for ss in range(0, 4):
                    if (h >> ss) < 2 or (w == 4 and h == 4 and ss):
                        continue
                    rows.append((bd, 0, ox, oy, cx, cy, w, h, ss, R.vtmref_dist(0, 1, po, W, pc, W, w, h, bd, ss)))
                    rows.append((bd, 3, ox, oy, cx, cy, w, h, ss, R.vtmref_dist(3, 1, po, W, pc, W, w, h, bd, ss)))     # D4: MR-SAD table entry
                rows.append((bd, 1, ox, oy, cx, cy, w, h, 0, R.vtmref_dist(1, 1, po, W, pc, W, w, h, bd, 0)))
                rows.append((bd, 2, ox, oy, cx, cy, w, h, 0, R.vtmref_dist(2, 1, po, W, pc, W, w, h, bd, 0)))
                rows.append((bd, 4, ox, oy, cx, cy, w, h, 0, R.vtmref_dist(4, 1, po, W, pc, W, w, h, bd, 0)))          # D4: MR-SATD
    out["rows"] = np.array(rows, dtype=np.int64)

    class MV(C.Structure):
        _fields_ = [("l", C.c_double), ("ph", C.c_int32), ("pv", C.c_int32), ("cs", C.c_int32), ("imv", C.c_int32)]
    mvrows = []
    for _ in range(300):
        m = MV(float(rng.uniform(0.5, 200)), int(rng.integers(-2000, 2000)), int(rng.integers(-2000, 2000)),
               int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        x, y = int(rng.integers(-500, 500)), int(rng.integers(-500, 500))
        mvrows.append((m.l, m.ph, m.pv, m.cs, m.imv, x, y, R.vtmref_mvcost(C.byref(m), x, y)))
    out["mvcost"] = np.array(mvrows, dtype=np.float64)
    save("dist", **out)


def gen_interp():
    rng = np.random.default_rng(1004)
    O_luma = np.array([[0, 0, 0, 64, 0, 0, 0, 0], [0, 1, -3, 63, 4, -2, 1, 0], [-1, 2, -5, 62, 8, -3, 1, 0], [-1, 3, -8, 60, 13, -4, 1, 0],
                       [-1, 4, -10, 58, 17, -5, 1, 0], [-1, 4, -11, 52, 26, -8, 3, -1], [-1, 3, -9, 47, 31, -10, 4, -1],
                       [-1, 4, -11, 45, 34, -10, 4, -1], [-1, 4, -11, 40, 40, -11, 4, -1]], np.int16)
    out = {}
    W, H, M = 96, 80, 8
    for bd in (8, 10):
        mx = (1 << bd) - 1
        ref_ = cases.rand_plane(rng, H, W, bd, "smooth")
        out["ref%d" % bd] = ref_
        rows, outs = [], []
        for (w, h) in [(4, 4), (8, 8), (16, 8), (12, 16), (32, 32), (17, 9)]:
            for luma in (1, 0):
                nf = 16 if luma else 32
                for (fx, fy) in [(0, 0), (5 % nf, 0), (0, 9 % nf), (3, 7), (nf - 1, nf - 3)]:
                    for rnd in (0, 1):
                        x, y = int(rng.integers(M, W - w - M)), int(rng.integers(M, H - h - M))
                        d = np.zeros((h, w), np.int16)
                        R.vtmref_pred_blk(1, C.c_void_p(ref_.ctypes.data + 2 * (y * W + x)), W, p(d), w, w, h, fx, fy, luma, rnd, bd, 0, mx)
                        rows.append((x, y, w, h, luma, fx, fy, rnd, len(outs)))
                        outs.append(d.reshape(-1))
        out["pred_rows%d" % bd] = np.array(rows, np.int32)
        out["pred_out%d" % bd] = np.concatenate(outs)
        out["pred_off%d" % bd] = np.cumsum([0] + [o.size for o in outs]).astype(np.int64)
        # table slots incl. second-stage inputs (14-bit intermediates)
        inter = rng.integers(-8192, 8192 + mx * 16, (H, W)).astype(np.int16)
        out["inter%d" % bd] = inter
        rows, outs = [], []
        for (w, h) in [(4, 4), (8, 8), (24, 8), (17, 9)]:
            for N in (0, 8, 4, 2):
                for isV in (0, 1):
                    for isF in (0, 1):
                        for isL in (0, 1):
                            if N and not isV and not isF:
                                continue
                            cf = np.zeros(8, np.int16)
                            if N == 8:
                                cf[:] = O_luma[int(rng.integers(1, 9))]
                            elif N == 4:
                                cf[:4] = [-4, 36, 36, -4]
                            elif N == 2:
                                cf[:2] = [40, 24]
                            src = ref_ if isF else inter
                            x, y = int(rng.integers(M, W - w - M)), int(rng.integers(M, H - h - M))
                            d = np.zeros((h, w), np.int16)
                            R.vtmref_if_call(1, N, isV, isF, isL, C.c_void_p(src.ctypes.data + 2 * (y * W + x)), W, p(d), w, w, h, p(cf), bd, 0, mx)
                            rows.append([x, y, w, h, N, isV, isF, isL] + list(cf))
                            outs.append(d.reshape(-1))
        out["if_rows%d" % bd] = np.array(rows, np.int32)
        out["if_out%d" % bd] = np.concatenate(outs)
    save("interp", **out)


def gen_transform():
    rng = np.random.default_rng(1005)
    out = {}
    for bd in (8, 10):
        mx = (1 << bd) - 1
        rows, resis, coefs, invs = [], [], [], []
        for w in (2, 4, 8, 16, 32, 64):
            for h in (2, 4, 8, 16, 32, 64):
                for (th, tv) in [(0, 0), (1, 1), (1, 2), (2, 1), (2, 2)]:
                    if th and (w < 4 or h < 4 or w > 32 or h > 32):
                        continue
                    if w * h > 1024 and (th, tv) not in [(0, 0), (2, 2)]:
                        continue
                    r = rng.integers(-mx, mx + 1, (h, w)).astype(np.int16)
                    c = np.zeros((h, w), np.int32)
                    R.vtmref_fwd_tr2d(bd, p(r), w, p(c), w, h, th, tv)
                    q = ((c >> 4) << 4).astype(np.int32)
                    ri = np.zeros((h, w), np.int16)
                    R.vtmref_inv_tr2d(bd, p(q), p(ri), w, w, h, th, tv)
                    rows.append((w, h, th, tv))
                    resis.append(r.reshape(-1)); coefs.append(c.reshape(-1)); invs.append(ri.reshape(-1))
        out["rows%d" % bd] = np.array(rows, np.int32)
        out["resi%d" % bd] = np.concatenate(resis)
        out["coef%d" % bd] = np.concatenate(coefs)
        out["inv%d" % bd] = np.concatenate(invs)
    save("transform", **out)


def gen_tskip():
    """transform skip through the reference's own TrQuant::xTransformSkip / xITransformSkip (private members, entered by
    oracle/ref_wrap_kernels.h:vtmref_transform_skip): every W x H in 2..64 incl. the sqrt(2)-scaled rectangular shapes."""
    rng = np.random.default_rng(1007)
    out = {}
    for bd in (8, 10):
        mx = (1 << bd) - 1
        rows, resis, coefs, cins, invs = [], [], [], [], []
        for w in (2, 4, 8, 16, 32, 64):
            for h in (2, 4, 8, 16, 32, 64):
                r = rng.integers(-mx, mx + 1, (h, w)).astype(np.int16)
                c = np.zeros((h, w), np.int32)
                R.vtmref_transform_skip(0, bd, p(r), w, p(c), w, h)
                q = rng.integers(-32768, 32768, (h, w)).astype(np.int32)
                ri = np.zeros((h, w), np.int16)
                R.vtmref_transform_skip(1, bd, p(ri), w, p(q), w, h)
                rows.append((w, h))
                resis.append(r.reshape(-1)); coefs.append(c.reshape(-1)); cins.append(q.reshape(-1)); invs.append(ri.reshape(-1))
        out["rows%d" % bd] = np.array(rows, np.int32)
        out["resi%d" % bd] = np.concatenate(resis)
        out["coef%d" % bd] = np.concatenate(coefs)
        out["cin%d" % bd] = np.concatenate(cins)
        out["inv%d" % bd] = np.concatenate(invs)
    save("tskip", **out)


def gen_dequant():
    """next row N1: Quant::dequant and the dependent-quantisation state machine of the compiled reference (DepQuant::dequant),
    every W x H, several QPs, sparse / dense / full-range levels; plus the coefficient scans themselves."""
    rng = np.random.default_rng(1009)
    out = {}
    scans = []
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            sc = np.zeros(w * h, np.uint32)
            assert R.vtmref_scan_order(w, h, p(sc)) == 0
            scans.append(sc.astype(np.uint16))
    out["scan"] = np.concatenate(scans)
    for bd in (8, 10):
        rows, lvs, outs = [], [], []
        for w in (2, 4, 8, 16, 32, 64):
            for h in (2, 4, 8, 16, 32, 64):
                for qp in (1, 17, 22 + (bd - 8) * 6, 37, 51 + (bd - 8) * 6):
                    for dq in (0, 1):
                        kind = int(rng.integers(0, 3))
                        if kind == 0:
                            lv = rng.integers(-40, 41, w * h)
                        elif kind == 1:
                            lv = rng.integers(-6, 7, w * h) * (rng.random(w * h) < 0.2)
                        else:
                            lv = rng.integers(-32768, 32768, w * h)
                        lv = lv.astype(np.int32)
                        o = np.zeros(w * h, np.int32)
                        R.vtmref_dequant(dq, bd, qp, 0, p(lv), p(o), w, h)
                        rows.append((w, h, qp, dq))
                        lvs.append(lv); outs.append(o)
        out["rows%d" % bd] = np.array(rows, np.int32)
        out["level%d" % bd] = np.concatenate(lvs)
        out["coef%d" % bd] = np.concatenate(outs)
    save("dequant", **out)


def gen_rdpcm():
    """residual DPCM (row T3) through the reference's own TrQuant::applyForwardRDPCM / invRdpcmNxN (oracle/ref_wrap_kernels.h:vtmref_rdpcm):
    every mode x lossless x rotation (4-wide) x slice type at 8 and 10 bit, small and saturating residuals."""
    rng = np.random.default_rng(1021)
    rows, resis, coefs, sums, invin, invout = [], [], [], [], [], []
    for bd in (8, 10):
        for (w, h) in [(4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (4, 16), (16, 4), (32, 8)]:
            for mode in (0, 1, 2):
                for lossless in (0, 1):
                    for rot in ((0, 1) if w == 4 else (0,)):
                        for intra in (0, 1):
                            qp = int(rng.integers(12, 45)) + 6 * (bd - 8)
                            amp = int(rng.choice([3, 40, 600, 1023]))
                            r = rng.integers(-amp, amp + 1, (h, w)).astype(np.int16)
                            c = np.zeros(w * h, np.int32)
                            sm = np.zeros(1, np.uint32)
                            rin = r.copy()                         # (a named array: the pointer must outlive the call)
                            R.vtmref_rdpcm(0, bd, qp, mode, lossless, rot, intra, p(rin), w, w, h, p(c), p(sm))
                            a = rng.integers(-3000, 3000, (h, w)).astype(np.int16)
                            b = a.copy()
                            if rot == 0:
                                cdum, sdum = np.zeros(w * h, np.int32), np.zeros(1, np.uint32)
                                R.vtmref_rdpcm(1, bd, qp, mode, lossless, 0, intra, p(b), w, w, h, p(cdum), p(sdum))
                            rows.append((bd, w, h, mode, lossless, rot, intra, qp))
                            resis.append(r.reshape(-1)); coefs.append(c); sums.append(int(sm[0])); invin.append(a.reshape(-1)); invout.append(b.reshape(-1))
    save("rdpcm", rows=np.array(rows, np.int32), resi=np.concatenate(resis), coef=np.concatenate(coefs), abs_sum=np.array(sums, np.uint32),
         inv_in=np.concatenate(invin), inv_out=np.concatenate(invout))


def gen_affine_mv():
    """affine sub-block vectors (row I3), pinned end to end: the reference's own InterPrediction::xPredAffineBlk (vtmref_affine_pred) predicts
    luma, Cb and Cr of PUs with 4- and 6-parameter models from seeded planes; the tests derive the sub-block descriptors and run the
    (separately pinned) block interpolation on them -- equal predictions pin the vectors, their rounding, clipping and the phase split."""
    rng = np.random.default_rng(1022)
    W, H, bd = 256, 128, 10
    Y = rng.integers(0, 1024, (H, W)).astype(np.int16)
    Cb = rng.integers(0, 1024, (H // 2, W // 2)).astype(np.int16)
    Cr = rng.integers(0, 1024, (H // 2, W // 2)).astype(np.int16)
    rows, preds = [], []
    for trial in range(120):
        w, h = int(rng.choice([8, 16, 32, 64, 128])), int(rng.choice([8, 16, 32, 64, 128]))
        px, py = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4
        six = int(rng.integers(0, 2))
        amp = int(rng.choice([8, 64, 400, 3000]))
        mv = rng.integers(-amp, amp + 1, (3, 2)).astype(np.int32)
        if trial % 7 == 0:
            mv[1] = mv[0]; mv[2] = mv[0]
        if trial % 11 == 0:
            mv[0] = (-4000, 3000); mv[1] = (-4007, 3001); mv[2] = (-3990, 3005)
        rows.append((px, py, w, h, six) + tuple(int(v) for v in mv.reshape(-1)))
        for comp in range(3):
            c = 1 if comp else 0
            want = np.zeros((h >> c, w >> c), np.int16)
            R.vtmref_affine_pred(comp, W, H, bd, p(Y), p(Cb), p(Cr), px, py, w, h, p(np.ascontiguousarray(mv.reshape(-1))), six, 0, p(want), w >> c)
            preds.append(want.reshape(-1))
    save("affine_mv", Y=Y, Cb=Cb, Cr=Cr, rows=np.array(rows, np.int32), pred=np.concatenate(preds))


def gen_affine():
    """next row N3: Sobel derivative planes and equal-coefficient sums from the compiled reference's SIMD table slots."""
    rng = np.random.default_rng(1011)
    out = {}
    rows, preds, gxs, gys, resis, eqs = [], [], [], [], [], []
    for (w, h) in [(16, 16), (16, 32), (32, 16), (64, 64), (128, 64), (16, 8), (8, 16), (128, 128), (32, 32)]:
        pred = rng.integers(0, 1024, (h, w)).astype(np.int16)
        gx = np.zeros((h, w), np.int32)
        gy = np.zeros((h, w), np.int32)
        R.vtmref_affine_sobel(1, 0, p(pred), w, p(gx), w, w, h)
        R.vtmref_affine_sobel(1, 1, p(pred), w, p(gy), w, w, h)
        resi = rng.integers(-1023, 1024, (h, w)).astype(np.int16)
        for six in (0, 1):
            eq = np.zeros(49, np.int64)
            R.vtmref_affine_equal_coeff(1, p(resi), p(gx), p(gy), w, w, h, six, p(eq))
            eqs.append(eq)
        rows.append((w, h))
        preds.append(pred.reshape(-1)); gxs.append(gx.reshape(-1)); gys.append(gy.reshape(-1)); resis.append(resi.reshape(-1))
    out["rows"] = np.array(rows, np.int32)
    out["pred"] = np.concatenate(preds); out["gx"] = np.concatenate(gxs); out["gy"] = np.concatenate(gys)
    out["resi"] = np.concatenate(resis); out["eq"] = np.concatenate(eqs)
    save("affine", **out)


def gen_frac():
    rng = np.random.default_rng(1006)
    FB = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4")])
    FR = np.dtype([("half_x", "<i4"), ("half_y", "<i4"), ("qter_x", "<i4"), ("qter_y", "<i4"), ("cost_half", "<u8"), ("cost", "<u8")])

    class MV(C.Structure):
        _fields_ = [("l", C.c_double), ("ph", C.c_int32), ("pv", C.c_int32), ("cs", C.c_int32), ("imv", C.c_int32)]
    out = {}
    W, H, M = 128, 96, 16
    for bd in (8, 10):
        mx = (1 << bd) - 1
        ref_ = cases.rand_plane(rng, H + 2 * M, W + 2 * M, bd, "smooth")
        org = np.clip(ref_[M + 1:M + 1 + H, M + 2:M + 2 + W].astype(np.int32) + rng.integers(-6, 7, (H, W)), 0, mx).astype(np.int16)
        org = np.ascontiguousarray(org)
        out["ref%d" % bd], out["org%d" % bd] = ref_, org
        rows = []
        for (w, h) in [(4, 4), (8, 8), (16, 16), (16, 8), (8, 16), (32, 32), (64, 64), (8, 4), (32, 16)]:
            for had in (1, 0):
                nb = 4
                blk = np.zeros(nb, FB)
                for i in range(nb):
                    x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
                    mvx, mvy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
                    blk[i] = (x, y, M + x + mvx, M + y + mvy, mvx, mvy)
                m = MV(float(rng.uniform(2, 40)), int(rng.integers(-20, 20)), int(rng.integers(-20, 20)), 0, 0)
                res = np.zeros(nb, FR)
                R.vtmref_frac_refine(p(org), W, p(ref_), W + 2 * M, p(blk), nb, w, h, bd, 0, mx, had, C.byref(m), p(res))
                for i in range(nb):
                    rows.append([bd, w, h, had, m.ph, m.pv] + [int(v) for v in blk[i]] + [int(res[i][k]) for k in FR.names] + [m.l])
        out["rows%d" % bd] = np.array(rows, dtype=np.float64)
    save("frac", **out)


def gen_tzsearch():
    """next row N2: the reference's own InterSearch::xTZSearch on blob-texture planes with a true displacement."""
    rng = np.random.default_rng(1007)
    out = {}
    W, H, M = 192, 128, 160
    sizes = [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (8, 16), (32, 64), (4, 8), (64, 16), (128, 64), (12, 16), (24, 32)]
    for k, (bd, motion) in enumerate([(8, (7, -5)), (10, (-21, 13))]):
        org, ref_ = cases.tz_planes(rng, W, H, M, bd, motion)
        out["org%d" % k], out["ref%d" % k], out["bd%d" % k] = org, ref_, np.array(bd)
        for j, (rng_, stop) in enumerate([(64, 0), (96, 1), (8, 0)]):
            n = 40
            pus = cases.tz_pus(rng, n, W, H, M, sizes)
            cfg = cases.tz_cfg(W, H, M, float(rng.uniform(4, 60)), search_range=rng_, first_stop=stop)
            res = np.zeros(n, cases.BEST)
            assert R.vtmref_tz_search(p(org), W, p(ref_), W + 2 * M, p(pus), n, p(cfg), bd, p(res)) == 0
            out["pus%d_%d" % (k, j)], out["cfg%d_%d" % (k, j)], out["res%d_%d" % (k, j)] = pus, cfg, res
    save("tzsearch", **out)


def gen_picture():
    """next row N4, picture-level passes: Picture::extendPicBorder on a real Picture; compCRC / compChecksum."""
    rng = np.random.default_rng(1008)
    out = {}
    for k, (w, h, bd, margin) in enumerate([(64, 32, 8, 16), (200, 120, 10, 80), (136, 72, 8, 144), (208, 120, 10, 144)]):
        mx = (1 << bd) - 1
        planes = [rng.integers(0, mx + 1, (h >> (c > 0), w >> (c > 0))).astype(np.int16) for c in range(3)]
        outs = [np.zeros((pl.shape[0] + 2 * (margin >> (c > 0)), pl.shape[1] + 2 * (margin >> (c > 0))), np.int16) for c, pl in enumerate(planes)]
        R.vtmref_extend_border(p(planes[0]), p(planes[1]), p(planes[2]), w, h, 128, margin, p(outs[0]), p(outs[1]), p(outs[2]))
        out["meta%d" % k] = np.array([w, h, bd, margin])
        for c in range(3):
            out["in%d_%d" % (k, c)] = planes[c]
            # the padded result is np.pad(mode="edge") of the input iff the reference did what the restatement says; keep a digest
            out["padsum%d_%d" % (k, c)] = np.array([int(outs[c].astype(np.int64).sum()), int((outs[c].astype(np.int64) * np.arange(outs[c].size).reshape(outs[c].shape) % 65521).sum())])
            assert np.array_equal(outs[c], np.pad(planes[c], margin >> (c > 0), mode="edge"))
            out["crc%d_%d" % (k, c)] = np.array(R.vtmref_crc(bd, p(planes[c]), planes[c].shape[1], planes[c].shape[1], planes[c].shape[0]) & 0xffffffff, np.uint32)
            out["sum%d_%d" % (k, c)] = np.array(R.vtmref_checksum(bd, p(planes[c]), planes[c].shape[1], planes[c].shape[1], planes[c].shape[0]) & 0xffffffff, np.uint32)
    save("picture", **out)


def gen_intra():
    """next row N4: the reference's own IntraPrediction::predIntraAng (+ xFilterReferenceSamples) per block shape x mode."""
    rng = np.random.default_rng(1009)
    out = {}
    shapes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16), (4, 64), (64, 4), (32, 16)]
    rows, refs_all, pred_all = [], [], []
    for si, (w, h) in enumerate(shapes):
        t, l = C.c_int(), C.c_int()
        R.vtmref_intra_ref_lengths(w, h, C.byref(t), C.byref(l))
        T, L = t.value, l.value
        modes = range(67) if w * h < 4096 else [0, 1, 2, 3, 10, 18, 19, 34, 49, 50, 58, 66]
        for mode in modes:
            bd = 8 if (mode + si) % 3 == 0 else 10
            mx = (1 << bd) - 1
            kind = (mode + si) % 3
            if kind == 0:
                refs = rng.integers(0, mx + 1, T + L + 1).astype(np.int16)
            elif kind == 1:
                refs = np.clip(np.cumsum(rng.integers(-6, 7, T + L + 1)) + mx // 2, 0, mx).astype(np.int16)
            else:
                refs = rng.choice(np.array([0, mx], np.int16), T + L + 1)
            filt = (mode + si) & 1
            pred = np.zeros((h, w), np.int16)
            R.vtmref_intra_pred(p(refs), p(pred), w, w, h, mode, 0, mx, bd, filt, None)
            rows.append((w, h, mode, bd, filt, T, L, sum(len(r) for r in refs_all), sum(x.size for x in pred_all)))
            refs_all.append(refs); pred_all.append(pred.reshape(-1))
    out["rows"] = np.array(rows, np.int64)
    out["refs"] = np.concatenate(refs_all); out["pred"] = np.concatenate(pred_all)
    save("intra", **out)


def gen_imv():
    """next row N2 (AMVR): the reference's own InterSearch::xPatternSearchIntRefine."""
    rng = np.random.default_rng(1010)
    out = {}
    W, H, M = 192, 128, 160
    sizes = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 64), (4, 8), (64, 16), (128, 64), (8, 4), (4, 4)]
    org, ref_ = cases.tz_planes(rng, W, H, M, 10, (6, -9))
    out["org"], out["ref"] = org, ref_
    for j, (sh, had, wgt) in enumerate([(2, 1, 1.0), (4, 1, 0.5), (2, 0, 1.37), (4, 0, 1.0)]):
        n = 60
        pus = cases.imv_pus(rng, n, W, H, M, sizes, sh)
        cfg = cases.tz_cfg(W, H, M, float(rng.uniform(4, 60)), imv_shift=sh)
        res = np.zeros(n, cases.IMV_RESULT)
        R.vtmref_imv_refine(p(org), W, p(ref_), W + 2 * M, p(pus), n, p(cfg), 10, had, C.c_double(wgt), p(res))
        out["pus%d" % j], out["cfg%d" % j], out["res%d" % j], out["par%d" % j] = pus, cfg, res, np.array([sh, had, wgt])
    save("imv", **out)


def gen_quant():
    """next row N1 (forward, no RDOQ): the reference's own Quant::quant incl. sign bit hiding."""
    rng = np.random.default_rng(1011)
    R.vtmref_quant.restype = C.c_uint32
    rows, coefs, levels = [], [], []
    off = 0
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            for it in range(4):
                bd = 8 if it % 2 else 10
                qp = int(rng.integers(4, 50 + (bd - 8) * 6))
                n = w * h
                coef = (rng.normal(0, 400 * (1 + 3 * it), n) * (rng.random(n) < (0.25 + 0.25 * it))).astype(np.int32)
                intra, sbh = int(rng.integers(0, 2)), int(it != 3)
                lv = np.zeros(n, np.int32)
                s = R.vtmref_quant(p(coef), p(lv), w, h, bd, qp, intra, sbh)
                rows.append((w, h, bd, qp, intra, sbh, off, s))
                coefs.append(coef); levels.append(lv); off += n
    save("quant", rows=np.array(rows, np.int64), coef=np.concatenate(coefs), level=np.concatenate(levels))


def gen_depquant():
    """next row N1: the reference's own DepQuant::quant (dependent-quantisation trellis), luma and chroma TUs, with the rate tables
    the reference derived from its CABAC contexts (re-derived through Ctx's public FracBitsAccess by the wrapper)."""
    rng = np.random.default_rng(1012)
    R.vtmref_depquant.restype = C.c_uint32
    RATES = np.dtype([("last_x", "<i4", (64,)), ("last_y", "<i4", (64,)), ("sig_sbb", "<i4", (2, 2)), ("sig", "<i4", (3, 18, 2)), ("gtx", "<i4", (21, 7))])
    rows, coefs, levels, rates = [], [], [], []
    off = 0
    shapes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (16, 4), (4, 16), (32, 8), (8, 32), (64, 16), (16, 64), (32, 64), (16, 8)]
    for (w, h) in shapes:
        for comp in (0, 1):
            if comp and max(w, h) > 32:
                continue
            for it in range(4 if w * h <= 1024 else 2):
                bd = 8 if it % 2 else 10
                qp = int(rng.integers(10, 46 + (bd - 8) * 6))
                n = w * h
                yy, xx = np.mgrid[0:h, 0:w]
                decay = np.exp(-(xx / w * 3 + yy / h * 3))
                kind = it % 3
                coef = rng.normal(0, [4000, 600, 15000][kind], (h, w)) * decay * (1 if kind < 2 else (rng.random((h, w)) < 0.2))
                coef = np.ascontiguousarray(coef.astype(np.int32).reshape(-1))
                lam = float(rng.uniform(5, 400))
                cq, init = int(rng.integers(20, 45)), int(rng.integers(0, 3))
                rt = np.zeros(1, RATES)
                lv = np.zeros(n, np.int32)
                s = R.vtmref_depquant(p(coef), p(lv), w, h, comp, bd, qp, C.c_double(lam), cq, init, p(rt))
                rows.append((w, h, comp, bd, qp, off, s, len(rates)))
                rates.append(rt[0]); coefs.append(coef); levels.append(lv); off += n
                rows[-1] = rows[-1] + (lam,)
    save("depquant", rows=np.array(rows, np.float64), coef=np.concatenate(coefs), level=np.concatenate(levels), rates=np.array(rates, RATES))


def gen_rdoq():
    """next row N1: the reference's own QuantRDOQ::quant (xRateDistOptQuant), luma and chroma TUs of inter and intra CUs, sign hiding on
    and off, transform-skip flagged 4x4 blocks, with the fractional-bit tables gathered from the same CABAC context object."""
    rng = np.random.default_rng(1013)
    R.vtmref_rdoq.restype = C.c_uint32
    RATES = np.dtype([("sig", "<i4", (18, 2)), ("par", "<i4", (21, 2)), ("gt1", "<i4", (21, 2)), ("gt2", "<i4", (21, 2)), ("sig_group", "<i4", (2, 2)),
                      ("last_x", "<i4", (14,)), ("last_y", "<i4", (14,)), ("cbf", "<i4", (2,))])
    assert RATES.itemsize == 784
    rows, coefs, levels, rates = [], [], [], []
    off = 0
    shapes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (16, 4), (4, 16), (32, 8), (8, 32), (64, 16), (16, 64), (32, 64), (16, 8)]
    for (w, h) in shapes:
        for comp in (0, 1):
            if comp and max(w, h) > 32:
                continue
            for it in range(6 if w * h <= 1024 else 3):
                bd = 8 if it % 2 else 10
                qp = int(rng.integers(10, 46 + (bd - 8) * 6))
                n = w * h
                yy, xx = np.mgrid[0:h, 0:w]
                decay = np.exp(-(xx / w * 3 + yy / h * 3))
                kind = it % 3
                coef = rng.normal(0, [4000, 600, 15000][kind], (h, w)) * decay * (1 if kind < 2 else (rng.random((h, w)) < 0.2))
                ts = int(w == 4 and h == 4 and it >= 4)
                if ts:
                    coef = rng.normal(0, 300, (h, w))            # transform-skip residuals have no decay
                coef = np.ascontiguousarray(coef.astype(np.int32).reshape(-1))
                lam = float(rng.uniform(5, 400))
                cq, init = int(rng.integers(20, 45)), int(rng.integers(0, 3))
                intra, sbh = int(rng.integers(0, 2)), int(it % 3 != 1)
                rt = np.zeros(1, RATES)
                lv = np.zeros(n, np.int32)
                s = R.vtmref_rdoq(p(coef), p(lv), w, h, comp, bd, qp, C.c_double(lam), cq, init, intra, sbh, ts, p(rt))
                rows.append((w, h, comp, bd, qp, off, s, len(rates), lam, sbh))
                rates.append(rt[0]); coefs.append(coef); levels.append(lv); off += n
    save("rdoq", rows=np.array(rows, np.float64), coef=np.concatenate(coefs), level=np.concatenate(levels), rates=np.array(rates, RATES))


def gen_pelop():
    """B1-B4: the reference's own PelBufferOps table (addAvg / reco / linTf, scalar and the SIMD set it installs itself) and the
    AreaBuf arithmetic that is not in the table (subtract, removeHighFreq, copyClip) -- every op x widths x 8 / 10 bit x clip on / off.  Rows:
    (op, w, h, bd, clip, scale, shift, offset, clp_min, clp_max, x0, y0, x1, y1, out offset); sources are windows of two shared planes."""
    rng = np.random.default_rng(1014)
    out = {}
    H, W = 160, 192
    for bd in (8, 10):
        mx = (1 << bd) - 1
        pel = cases.rand_plane(rng, H, W, bd, "uniform")                                   # picture samples
        inter = rng.integers(-8192, 8192 + mx * 16, (H, W)).astype(np.int16)               # 14-bit first-stage values (addAvg inputs)
        resi = rng.integers(-mx, mx + 1, (H, W)).astype(np.int16)                          # residual samples
        out.update({"pel%d" % bd: pel, "inter%d" % bd: inter, "resi%d" % bd: resi})
        rows, outs = [], []
        pos = 0
        for op in range(6):
            widths = [4, 8, 12, 16, 24, 32, 64, 128] if op < 3 else [2, 4, 6, 8, 12, 16, 24, 32, 64, 128]
            for w in widths:
                for h in (2, 4, 8, 16, 128) if w < 64 else (4, 64, 128):
                    for clip in (0, 1):
                        if op in (0, 1, 3, 5) and clip == 0 and op != 3:
                            continue                                                       # addAvg / reco / copyClip always clip
                        if op == 3 and clip == 1:
                            continue                                                       # subtract never clips
                        x0, y0 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
                        x1, y1 = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
                        cmin, cmax = (0, mx) if rng.random() < 0.7 else (int(rng.integers(0, 40)), mx - int(rng.integers(0, 40)))
                        scale = shift = offset = 0
                        if op == 0:
                            a, b = inter, inter
                            shift = max(2, 14 - bd) + 1
                            offset = (1 << (shift - 1)) + 2 * 8192
                        elif op == 1:
                            a, b = pel, resi
                        elif op == 2:
                            a, b = pel, pel
                            scale, shift, offset = int(rng.integers(-40, 41)), int(rng.integers(0, 7)), int(rng.integers(-64, 65))
                            if not clip:                                                   # an unclipped result beyond int16 is outside the domain: the reference's
                                scale = int(rng.integers(-30, 31))                         # SIMD form saturates there, its scalar form wraps (measured: gen asserts equality)
                        elif op == 3:
                            a, b = pel, pel
                        elif op == 4:
                            a, b = pel, pel
                        else:
                            a, b = resi, resi                                              # copyClip of out-of-range samples
                        d = np.full((h, w), -31000, np.int16)
                        a0 = C.c_void_p(a.ctypes.data + 2 * (y0 * W + x0))
                        b0 = C.c_void_p(b.ctypes.data + 2 * (y1 * W + x1))
                        if op < 3:
                            d2 = d.copy()
                            R.vtmref_pelop(1, op, a0, W, b0, W, p(d), w, w, h, scale, shift, offset, clip, bd, cmin, cmax)
                            R.vtmref_pelop(0, op, a0, W, b0, W, p(d2), w, w, h, scale, shift, offset, clip, bd, cmin, cmax)
                            assert np.array_equal(d, d2), ("reference SIMD != scalar", op, w, h, bd)
                        else:
                            R.vtmref_pelop_area(op, a0, W, b0, W, p(d), w, w, h, clip, bd, cmin, cmax)
                        rows.append((op, w, h, bd, clip, scale, shift, offset, cmin, cmax, x0, y0, x1, y1, pos))
                        outs.append(d.reshape(-1))
                        pos += d.size
        out["rows%d" % bd] = np.array(rows, np.int32)
        out["out%d" % bd] = np.concatenate(outs)
    save("pelop", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (gen_alf, gen_sao, gen_dist, gen_interp, gen_transform, gen_tskip, gen_dequant, gen_affine, gen_rdpcm, gen_affine_mv, gen_frac, gen_tzsearch, gen_picture, gen_intra, gen_imv, gen_quant, gen_depquant, gen_rdoq, gen_pelop):
        if not only or fn.__name__[4:] in only:
            fn()
