"""CPU leg of the canonical per-picture workload: the CHECKER side (test infrastructure).

`run_cpu(workload, lib, kind)` drives oracle/liboracle.so (kind 'port') or oracle/_ref/libvtmref.so (kind 'reference')
over the same seeded `vvcsoftware_vtm_amd.workload.Workload` object the HIP path runs.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under vvcsoftware_vtm_amd/ does."""
import ctypes as C
import time

import numpy as np

from vvcsoftware_vtm_amd.workload import CTU, FRAC_RESULT, SEARCH_BEST


def run_cpu(wl, lib, kind="port"):
    """One step on the host through a checker library.  kind 'port': oracle/liboracle.so (orc_* names);
    kind 'reference': oracle/_ref/libvtmref.so for every stage that has a reference entry point (deblocking and the
    element-wise plane ops have none and use `port_lib`).  Returns (outputs dict of numpy arrays, seconds per stage)."""
    P = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    port, refl = (lib, None) if kind == "port" else (lib[0], lib[1])
    w, h, bd, mx = wl.w, wl.h, wl.bd, wl.mx
    out, secs = {}, {}

    def timed(name, fn):
        t0 = time.perf_counter()
        fn()
        secs[name] = secs.get(name, 0.0) + time.perf_counter() - t0
    # me
    for s in sorted(wl.me):
        blk = wl.me[s]
        for (dx0, dy0, nx, ny, sx, sy) in wl.me_grids:
            sad = np.zeros((blk.size, ny, nx), np.uint32)
            best = np.zeros(blk.size, SEARCH_BEST)
            if refl is not None:
                timed("me", lambda: refl.vtmref_sad_search(P(wl.org[0]), w, P(wl.ref0_pad[0]), wl.pw, P(blk), blk.size, s, s, 1,
                                                            dx0, dy0, nx, ny, sx, sy, bd, P(sad)))
                best = None
            else:
                timed("me", lambda: port.orc_sad_search(P(wl.org[0]), w, P(wl.ref0_pad[0]), wl.pw, P(blk), blk.size, s, s, 1,
                                                         dx0, dy0, nx, ny, sx, sy, P(sad), C.byref(wl.mvcost), P(best)))
            out["me_sad_%d_%d" % (s, nx)] = sad
            out["me_best_%d_%d" % (s, nx)] = best
    # frac
    fres = np.zeros(wl.frac.size, FRAC_RESULT)
    if refl is not None:
        timed("frac", lambda: refl.vtmref_frac_refine(P(wl.org[0]), w, P(wl.ref0_pad[0]), wl.pw, P(wl.frac), wl.frac.size, 16, 16, bd, 0, mx, 1,
                                                      C.byref(wl.frac_mvcost), P(fres)))
    else:
        timed("frac", lambda: port.orc_frac_refine(P(wl.org[0]), w, P(wl.ref0_pad[0]), wl.pw, P(wl.frac), wl.frac.size, 16, 16, bd, 0, mx, 1,
                                                   C.byref(wl.frac_mvcost), P(fres)))
    out["frac"] = fres
    # mc
    pred = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
    f_mc = (lambda *a: refl.vtmref_mc_batch(*a)) if refl is not None else (lambda *a: port.orc_mc_batch(*a))
    timed("mc", lambda: f_mc(P(wl.ref0_pad[0]), P(wl.ref1_pad[0]), P(pred[0]), P(wl.mc_luma), wl.mc_luma.size, bd, 0, mx))
    for c in (1, 2):
        timed("mc", lambda: f_mc(P(wl.ref0_pad[c]), P(wl.ref1_pad[c]), P(pred[c]), P(wl.mc_chroma), wl.mc_chroma.size, bd, 0, mx))
    # residual / transform
    resi = np.zeros((h, w), np.int16)
    resi2 = np.zeros((h, w), np.int16)
    coef = np.zeros(wl.n_coef, np.int32)
    timed("resi", lambda: port.orc_pelop_batch(3, P(wl.org[0]), P(pred[0]), P(resi), P(wl.bands_luma), wl.bands_luma.size, C.byref(wl.cfg_sub)))
    if refl is not None:
        timed("resi", lambda: refl.vtmref_tr_fwd_batch(P(resi), P(coef), P(wl.tr), wl.tr.size, bd))
    else:
        timed("resi", lambda: port.orc_tr_fwd_batch(P(resi), P(coef), P(wl.tr), wl.tr.size, bd))
    level, dqcoef, abs_sum = np.zeros(wl.n_coef, np.int32), np.zeros(wl.n_coef, np.int32), np.zeros(wl.tr.size, np.uint32)
    if getattr(wl, "depquant", False):
        # the `with_depquant` leg: DepQuant::quant TU by TU (restatement; orc_depquant is pinned by tests/golden/depquant.npz), then the de-quantiser in its
        # dependent-quantisation form (dqtr.dep_quant = 1)
        port.orc_depquant.restype = C.c_uint32

        def dq_all():
            for i in range(wl.tr.size):
                d = wl.dq[i]
                n = int(d["w"]) * int(d["h"])
                off = int(d["coeff_off"])
                abs_sum[i] = port.orc_depquant(P(coef[off:off + n]), P(level[off:off + n]), int(d["w"]), int(d["h"]), int(d["luma"]), bd, int(d["qp"]),
                                               C.c_double(float(d["lambda"])), C.c_void_p(wl.dq_rates.ctypes.data + int(d["rates_idx"]) * wl.dq_rates.dtype.itemsize))
        timed("resi", dq_all)
        timed("resi", lambda: port.orc_dequant_tr_inv_batch(P(level), P(resi2), P(wl.dqtr), wl.tr.size, bd, P(dqcoef)))
    elif refl is not None:
        timed("resi", lambda: refl.vtmref_quant_batch(P(coef), P(level), P(wl.quant), wl.tr.size, bd, P(abs_sum)))
        timed("resi", lambda: refl.vtmref_dequant_tr_inv_batch(P(level), P(resi2), P(wl.dqtr), wl.tr.size, bd, P(dqcoef)))
    else:
        timed("resi", lambda: port.orc_quant_batch(P(coef), P(level), P(wl.quant), wl.tr.size, bd, P(abs_sum)))
        timed("resi", lambda: port.orc_dequant_tr_inv_batch(P(level), P(resi2), P(wl.dqtr), wl.tr.size, bd, P(dqcoef)))
    coef = level
    out["abs_sum"] = abs_sum
    rec = [np.zeros((h, w), np.int16), pred[1].copy(), pred[2].copy()]
    timed("resi", lambda: port.orc_pelop_batch(1, P(pred[0]), P(resi2), P(rec[0]), P(wl.bands_luma), wl.bands_luma.size, C.byref(wl.cfg_reco)))
    out["coef"] = coef
    # deblock (no reference entry point: port)
    timed("dbk", lambda: port.orc_deblock(P(rec[0]), w, P(rec[1]), P(rec[2]), w // 2, w, h, P(wl.edge_ver), P(wl.edge_hor),
                                          P(wl.qp_luma), P(wl.qp_chroma), C.byref(wl.dbk_cfg)))
    # sao
    sao_stats, sao_out = [], []
    for c in range(3):
        cs = CTU if c == 0 else CTU // 2
        pw_, ph_ = (w, h) if c == 0 else (w // 2, h // 2)
        stt = np.zeros((wl.nctu_x * wl.nctu_y, 5, 2, 32), np.int64)
        if refl is not None:
            timed("sao", lambda: refl.vtmref_sao_stats(c, P(wl.org[c]), pw_, P(rec[c]), pw_, pw_, ph_, cs, cs, bd, None, 5 if c == 0 else 3, 4 if c == 0 else 2, P(stt)))
        else:
            timed("sao", lambda: port.orc_sao_stats(P(wl.org[c]), pw_, P(rec[c]), pw_, pw_, ph_, cs, cs, bd, None, 5 if c == 0 else 3, 4 if c == 0 else 2, P(stt)))
        sao_stats.append(stt)
        so = rec[c].copy()
        f = refl.vtmref_sao_apply if refl is not None else port.orc_sao_apply
        timed("sao", lambda: f(P(rec[c]), pw_, P(so), pw_, pw_, ph_, cs, cs, bd, P(wl.sao[c]), 0, mx))
        sao_out.append(so)
    out["sao_stats"] = sao_stats
    # alf
    cls = np.zeros((h // 4, w // 4), np.uint16)
    alf_out = [p.copy() for p in sao_out]
    if refl is not None:
        timed("alf", lambda: refl.vtmref_alf_picture(1, P(sao_out[0]), P(sao_out[1]), P(sao_out[2]), P(alf_out[0]), P(alf_out[1]), P(alf_out[2]),
                                                     w, h, CTU, bd, 1, P(wl.alf_luma_coeff), P(wl.alf_chroma_coeff),
                                                     P(wl.alf_enable[0]), P(wl.alf_enable[1]), P(wl.alf_enable[2]), None))
        timed("alf", lambda: port.orc_alf_classify(P(sao_out[0]), w, w, h, bd, P(cls)))   # picture-wide classifier for the stats
    else:
        timed("alf", lambda: port.orc_alf_classify(P(sao_out[0]), w, w, h, bd, P(cls)))
        timed("alf", lambda: port.orc_alf_filter_luma(P(sao_out[0]), w, P(alf_out[0]), w, w, h, CTU, P(cls), 1, P(wl.alf_luma_coeff), P(wl.alf_enable[0]), 0, mx))
        for c in (1, 2):
            timed("alf", lambda: port.orc_alf_filter_chroma(P(sao_out[c]), w // 2, P(alf_out[c]), w // 2, w // 2, h // 2, CTU // 2, P(wl.alf_chroma_coeff), P(wl.alf_enable[c]), 0, mx))
    nct = wl.nctu_x * wl.nctu_y
    a7 = np.zeros((nct, 25, 183), np.int64)
    a5 = np.zeros((nct, 25, 57), np.int64)
    ac = [np.zeros((nct, 1, 57), np.int64) for _ in range(2)]
    if refl is not None:
        timed("alf", lambda: refl.vtmref_alf_stats(P(wl.org[0]), w, P(sao_out[0]), w, h, CTU, P(cls), 1, P(a7)))
        timed("alf", lambda: refl.vtmref_alf_stats(P(wl.org[0]), w, P(sao_out[0]), w, h, CTU, P(cls), 0, P(a5)))
        for i, c in enumerate((1, 2)):
            timed("alf", lambda: refl.vtmref_alf_stats(P(wl.org[c]), w // 2, P(sao_out[c]), w // 2, h // 2, CTU // 2, None, 0, P(ac[i])))
    else:
        timed("alf", lambda: port.orc_alf_stats(P(wl.org[0]), w, P(sao_out[0]), w, w, h, CTU, P(cls), 1, P(a7)))
        timed("alf", lambda: port.orc_alf_stats(P(wl.org[0]), w, P(sao_out[0]), w, w, h, CTU, P(cls), 0, P(a5)))
        for i, c in enumerate((1, 2)):
            timed("alf", lambda: port.orc_alf_stats(P(wl.org[c]), w // 2, P(sao_out[c]), w // 2, w // 2, h // 2, CTU // 2, None, 0, P(ac[i])))
    out.update({"cls": cls, "alf_stats7": a7, "alf_stats5": a5, "alf_stats_c": ac, "final": alf_out, "pred": pred})
    return out, secs


def _expgolomb_bits(v):
    """RdCost::xGetExpGolombNumberOfBits (RdCost.h:172-184), vectorised."""
    v = np.asarray(v, np.int64)
    t = np.where(v <= 0, (-v << 1) + 1, v << 1).astype(np.int64)
    length = np.ones_like(t)
    while True:
        big = t > 128
        if not big.any():
            break
        length = np.where(big, length + 14, length)
        t = np.where(big, t >> 7, t)
    return length + (np.floor(np.log2(t)).astype(np.int64) << 1)


def best_from_surface(sad, grid, mvcost):
    """arg-min of SAD + motion-vector cost over a SAD surface [nblocks, ny, nx] in scan order with strict `<`
    (xTZSearchHelp / xPatternSearch keep the FIRST minimum, InterSearch.cpp:269-343, 1909-1925): -> SEARCH_BEST records.
    Used when the checker is the compiled reference, whose entry point returns the surface only."""
    dx0, dy0, nx, ny, sx, sy = grid
    xs = dx0 + np.arange(nx) * sx
    ys = dy0 + np.arange(ny) * sy
    bx = _expgolomb_bits(((xs << mvcost.cost_scale) - mvcost.pred_hor) >> mvcost.imv_shift)
    by = _expgolomb_bits(((ys << mvcost.cost_scale) - mvcost.pred_ver) >> mvcost.imv_shift)
    bits = by[:, None] + bx[None, :]
    mvc = (mvcost.lambda_ * bits.astype(np.float64)).astype(np.uint64)          # Distortion(m_motionLambda * bits): truncation
    cost = sad.astype(np.uint64) + mvc[None]
    flat = cost.reshape(cost.shape[0], -1)
    idx = flat.argmin(axis=1)                                                    # first minimum in scan order
    best = np.zeros(cost.shape[0], SEARCH_BEST)
    j, i = idx // nx, idx % nx
    best["x"], best["y"] = xs[i], ys[j]
    best["cost"] = flat[np.arange(flat.shape[0]), idx]
    best["sad"] = sad.reshape(sad.shape[0], -1)[np.arange(flat.shape[0]), idx]
    return best
