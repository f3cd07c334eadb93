"""Host-side operator mirrors of the reference call sites, over torch CUDA tensors (torch is plumbing:
device memory + streams).  Every function goes through the C ABI (capi) -- no CPU fallback exists.

Planes are 2-D int16 CUDA tensors (rows may be strided views into a padded picture buffer)."""
import ctypes as C
import numpy as np
import torch

from . import capi


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _plane(t, name="plane"):
    assert t.is_cuda and t.dtype == torch.int16 and t.dim() == 2 and t.stride(1) == 1, "%s: need 2-D int16 CUDA plane" % name
    return capi.ptr(t), t.stride(0), t.shape[1], t.shape[0]


# ---- ALF (AdaptiveLoopFilter.cpp) ---------------------------------------------------------------
def alf_classify(src, bit_depth):
    """A1: returns (H/4, W/4) int16 tensor holding classIdx | transposeIdx << 8 (uint16 bit pattern)."""
    p, st, w, h = _plane(src)
    cls = torch.empty((h // 4, w // 4), dtype=torch.int16, device=src.device)
    capi.call("vvcgpu_alf_classify", p, st, w, h, bit_depth, capi.ptr(cls), _stream())
    return cls


def alf_filter_luma(src, dst, ctu, cls, filter_type, coeff, ctu_enable=None, clp=(0, 1023)):
    p, st, w, h = _plane(src)
    q, dt, w2, h2 = _plane(dst, "dst")
    assert (w, h) == (w2, h2)
    cf = np.ascontiguousarray(coeff, dtype=np.int16)
    assert cf.size == 25 * 13
    capi.call("vvcgpu_alf_filter_luma", p, st, q, dt, w, h, ctu, capi.ptr(cls), filter_type,
              C.c_void_p(cf.ctypes.data), capi.ptr(ctu_enable), clp[0], clp[1], _stream())
    return dst


def alf_filter_chroma(src, dst, ctu_c, coeff, ctu_enable=None, clp=(0, 1023)):
    p, st, w, h = _plane(src)
    q, dt, w2, h2 = _plane(dst, "dst")
    assert (w, h) == (w2, h2)
    cf = np.ascontiguousarray(coeff, dtype=np.int16)
    assert cf.size == 7
    capi.call("vvcgpu_alf_filter_chroma", p, st, q, dt, w, h, ctu_c, C.c_void_p(cf.ctypes.data),
              capi.ptr(ctu_enable), clp[0], clp[1], _stream())
    return dst


# ---- SAO (SampleAdaptiveOffset.cpp) -------------------------------------------------------------
SAO_DTYPE = np.dtype([("type", "i1"), ("avail", "u1"), ("offset", "<i2", (32,))])


def sao_params_to_device(params, device="cuda"):
    """params: numpy structured array of SAO_DTYPE (one per CTU, raster order)."""
    assert params.dtype == SAO_DTYPE
    return torch.from_numpy(params.view(np.uint8).copy()).to(device)


def sao_apply(src, dst, ctu_w, ctu_h, bit_depth, params_dev, clp=(0, 1023)):
    p, st, w, h = _plane(src)
    q, dt, w2, h2 = _plane(dst, "dst")
    assert (w, h) == (w2, h2)
    capi.call("vvcgpu_sao_apply", p, st, q, dt, w, h, ctu_w, ctu_h, bit_depth, capi.ptr(params_dev),
              clp[0], clp[1], _stream())
    return dst


# ---- Deblocking (LoopFilter.cpp) ----------------------------------------------------------------
class DeblockCfg(C.Structure):
    """vvcgpu_deblock_cfg"""
    _fields_ = [("bit_depth_luma", C.c_int32), ("bit_depth_chroma", C.c_int32),
                ("beta_offset_div2", C.c_int32), ("tc_offset_div2", C.c_int32),
                ("cb_qp_offset", C.c_int32), ("cr_qp_offset", C.c_int32),
                ("clp_min", C.c_int32 * 3), ("clp_max", C.c_int32 * 3)]


def deblock_cfg(bd=10, beta_off=0, tc_off=0, cb_off=0, cr_off=0):
    mx = (1 << bd) - 1
    return DeblockCfg(bd, bd, beta_off, tc_off, cb_off, cr_off, (C.c_int32 * 3)(0, 0, 0), (C.c_int32 * 3)(mx, mx, mx))


def deblock(Y, Cb, Cr, edge_ver, edge_hor, qp_luma, qp_chroma, cfg):
    """In place.  Maps are uint8/int8 CUDA tensors of shape (H/4, W/4)."""
    p, st, w, h = _plane(Y)
    if Cb is not None:
        pb, sc, _, _ = _plane(Cb)
        pr, sc2, _, _ = _plane(Cr)
        assert sc == sc2
    else:
        pb = pr = None
        sc = 0
    capi.call("vvcgpu_deblock", p, st, pb, pr, sc, w, h, capi.ptr(edge_ver), capi.ptr(edge_hor),
              capi.ptr(qp_luma), capi.ptr(qp_chroma), C.byref(cfg), _stream())


# ---- encoder-side statistics ----------------------------------------------------------------------
def sao_stats(org, rec, ctu_w, ctu_h, bit_depth, avail=None, skip_r=5, skip_b=4):
    """S2: returns int64 tensor (nCtu, 5 types, 2 {diff,count}, 32 classes)."""
    po, so, w, h = _plane(org, "org")
    pr, sr, w2, h2 = _plane(rec, "rec")
    assert (w, h) == (w2, h2)
    n = ((w + ctu_w - 1) // ctu_w) * ((h + ctu_h - 1) // ctu_h)
    out = torch.empty((n, 5, 2, 32), dtype=torch.int64, device=org.device)
    capi.call("vvcgpu_sao_stats", po, so, pr, sr, w, h, ctu_w, ctu_h, bit_depth, capi.ptr(avail), skip_r, skip_b,
              capi.ptr(out), _stream())
    return out


def alf_stats(org, rec, ctu, cls, filter_type):
    """A3: returns int64 tensor (nCtu, nClasses, N*N+N+1)."""
    po, so, w, h = _plane(org, "org")
    pr, sr, w2, h2 = _plane(rec, "rec")
    assert (w, h) == (w2, h2)
    n = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
    N = 13 if filter_type else 7
    ncls = 25 if cls is not None else 1
    out = torch.empty((n, ncls, N * N + N + 1), dtype=torch.int64, device=org.device)
    capi.call("vvcgpu_alf_stats", po, so, pr, sr, w, h, ctu, capi.ptr(cls), filter_type, capi.ptr(out), _stream())
    return out


# ---- block distortion (RdCost) ---------------------------------------------------------------------
DIST_DESC = np.dtype([("org_off", "<i8"), ("cur_off", "<i8"), ("org_stride", "<i4"), ("cur_stride", "<i4"),
                      ("w", "<i2"), ("h", "<i2"), ("sub_shift", "<i2"), ("reserved", "<i2")])
SEARCH_BLK = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4")])
SEARCH_BEST = np.dtype([("x", "<i4"), ("y", "<i4"), ("cost", "<u8"), ("sad", "<u8")])
SAD, HAD, SSE, MRSAD, MRHAD = 0, 1, 2, 3, 4


class MvCost(C.Structure):
    """vvcgpu_mvcost"""
    _fields_ = [("lambda_", C.c_double), ("pred_hor", C.c_int32), ("pred_ver", C.c_int32),
                ("cost_scale", C.c_int32), ("imv_shift", C.c_int32)]


def struct_to_device(arr, device="cuda"):
    return torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(device)


def dist_batch(kind, org_base, cur_base, descs_dev, n, bit_depth=10):
    """D1-D3: org_base / cur_base are int16 CUDA tensors (any shape, offsets are in elements from element 0)."""
    out = torch.empty(n, dtype=torch.int64, device=org_base.device)
    capi.call("vvcgpu_dist_batch", kind, capi.ptr(org_base), capi.ptr(cur_base), capi.ptr(descs_dev), n, bit_depth,
              capi.ptr(out), _stream())
    return out


def sad_search(org, ref, blocks_dev, nblocks, w, h, sub_shift, dx0, dy0, nx, ny, sx, sy, mvcost=None, want_sad=True):
    """D1 search form.  org/ref: 2-D int16 planes (ref may be a view into a padded buffer; positions are relative to
    the view's origin and may be negative as long as they stay inside the allocation).  Returns (sad[nblocks,ny,nx]
    int32 tensor, best uint8 tensor or None).  want_sad=False asks for the best candidate only (sad_out = NULL in the
    C ABI, what xPatternSearch / xTZSearch keep); the first return value is then None."""
    po, so, _, _ = _plane(org, "org")
    pr, sr, _, _ = _plane(ref, "ref")
    sad = torch.empty((nblocks, ny, nx), dtype=torch.int32, device=org.device) if want_sad else None
    best = None
    mv = None
    if mvcost is not None:
        best = torch.empty(nblocks * SEARCH_BEST.itemsize, dtype=torch.uint8, device=org.device)
        mv = C.byref(mvcost)
    capi.call("vvcgpu_sad_search", po, so, pr, sr, capi.ptr(blocks_dev), nblocks, w, h, sub_shift, dx0, dy0, nx, ny,
              sx, sy, capi.ptr(sad), mv, capi.ptr(best), _stream())
    return sad, best


class MeHierCfg(C.Structure):
    _fields_ = [("org_x", C.c_int32), ("org_y", C.c_int32), ("ref_x", C.c_int32), ("ref_y", C.c_int32), ("n16x", C.c_int32), ("n16y", C.c_int32),
                ("sub_shift", C.c_int32), ("raster_range", C.c_int32), ("raster_step", C.c_int32), ("dense_range", C.c_int32)]


def me_hier_search(org, ref, org_xy, ref_xy, n16x, n16y, sub_shift, raster_range, dense_range, mvcost, raster_step=5):
    """D1 + D5 hierarchical form (vvcgpu_me_hier_search): the step-5 raster and the +-dense_range grid of every 16x16 / 32x32 / 64x64 block of the grid
    in one launch.  Returns (raster, dense): two lists of three SEARCH_BEST uint8 tensors (16, 32, 64; None where the grid has no block of the size,
    dense = None with dense_range 0)."""
    po, so, _, _ = _plane(org, "org")
    pr, sr, _, _ = _plane(ref, "ref")
    cfg = MeHierCfg(org_xy[0], org_xy[1], ref_xy[0], ref_xy[1], n16x, n16y, sub_shift, raster_range, raster_step, dense_range)
    counts = [n16x * n16y, (n16x // 2) * (n16y // 2), (n16x // 4) * (n16y // 4)]
    mk = lambda: [torch.empty(n * SEARCH_BEST.itemsize, dtype=torch.uint8, device=org.device) if n else None for n in counts]
    raster, dense = mk(), (mk() if dense_range else None)
    arr = lambda ts: (C.c_void_p * 3)(*[capi.ptr(t) for t in ts])
    capi.call("vvcgpu_me_hier_search", po, so, pr, sr, C.byref(cfg), C.byref(mvcost), arr(raster), arr(dense) if dense else None, _stream())
    return raster, dense


# ---- N2: integer TZ search of whole PUs (InterSearch::xTZSearch) ---------------------------------------------
TZ_PU = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4"), ("start_x", "<i4"), ("start_y", "<i4"),
                  ("pred2_x", "<i4"), ("pred2_y", "<i4"), ("pos_x", "<i4"), ("pos_y", "<i4"), ("pred_hor", "<i4"), ("pred_ver", "<i4"),
                  ("w", "<i2"), ("h", "<i2"), ("sub_shift", "<i2"), ("flags", "<i2"), ("reserved", "<i4", (2,))])
TZ_CFG = np.dtype([("lambda", "<f8"), ("cost_scale", "<i4"), ("imv_shift", "<i4"), ("search_range", "<i4"), ("first_search_stop", "<i4"),
                   ("pic_w", "<i4"), ("pic_h", "<i4"), ("max_cu_w", "<i4"), ("max_cu_h", "<i4"),
                   ("ref_x0", "<i4"), ("ref_y0", "<i4"), ("ref_x1", "<i4"), ("ref_y1", "<i4"), ("wg_per_pu", "<i4"), ("reserved", "<i4")])   # "reserved" = vvcgpu_tz_cfg.uniform_pu (the field keeps its name: the golden fixtures store this dtype)
assert TZ_PU.itemsize == 64 and TZ_CFG.itemsize == 64
TZ_PRED2, TZ_EXTENDED, TZ_FAST = 1, 2, 4


def tz_search_batch(org, ref, pus_dev, n, cfg):
    """N2: xTZSearch for n PUs (one wavefront each).  org/ref: 2-D int16 planes (ref = the whole padded reference plane,
    PU positions in plane coordinates); pus_dev: TZ_PU records on the device; cfg: one-element TZ_CFG numpy record (host).
    Returns SEARCH_BEST records as a uint8 tensor: x, y (integer MV), cost (uiBestSad), sad (ruiSAD)."""
    po, so, _, _ = _plane(org, "org")
    pr, sr, _, _ = _plane(ref, "ref")
    cfg = np.ascontiguousarray(cfg)
    assert cfg.dtype == TZ_CFG and cfg.size == 1
    best = torch.empty(n * SEARCH_BEST.itemsize, dtype=torch.uint8, device=org.device)
    capi.call("vvcgpu_tz_search_batch", po, so, pr, sr, capi.ptr(pus_dev), n, C.c_void_p(cfg.ctypes.data), capi.ptr(best), _stream())
    return best


def me_batch(org, ref, pus_dev, n, w, h, cfg, bit_depth, use_hadamard=True):
    """N2 chained: TZ search + fused fractional refinement of the same PUs (xPatternSearchFast -> xPatternSearchFracDIF).
    Returns (SEARCH_BEST uint8 tensor, FRAC_RESULT uint8 tensor)."""
    po, so, _, _ = _plane(org, "org")
    pr, sr, _, _ = _plane(ref, "ref")
    cfg = np.ascontiguousarray(cfg)
    assert cfg.dtype == TZ_CFG and cfg.size == 1
    best = torch.empty(n * SEARCH_BEST.itemsize, dtype=torch.uint8, device=org.device)
    frac = torch.empty(n * FRAC_RESULT.itemsize, dtype=torch.uint8, device=org.device)
    capi.call("vvcgpu_me_batch", po, so, pr, sr, capi.ptr(pus_dev), n, w, h, C.c_void_p(cfg.ctypes.data), bit_depth, 0, (1 << bit_depth) - 1,
              1 if use_hadamard else 0, capi.ptr(best), capi.ptr(frac), _stream())
    return best, frac


# ---- N4 picture-level passes: border extension, picture hash -------------------------------------------------
HASH_CRC, HASH_CHECKSUM = 1, 2


def extend_border(padded, margin_x, margin_y):
    """Picture::extendPicBorder for one plane.  padded: 2-D int16 tensor holding the picture with its margins."""
    assert padded.dim() == 2 and padded.dtype == torch.int16 and padded.stride(1) == 1
    h, w = padded.shape[0] - 2 * margin_y, padded.shape[1] - 2 * margin_x
    origin = padded.data_ptr() + (margin_y * padded.stride(0) + margin_x) * 2
    capi.call("vvcgpu_extend_border", C.c_void_p(origin), padded.stride(0), w, h, margin_x, margin_y, _stream())


def picture_hash(method, plane, bit_depth):
    """compCRC (method 1) / compChecksum (method 2) of one plane -> 1-element int32 tensor (device), bits as uint32."""
    pp, sp, w, h = _plane(plane, "plane")
    out = torch.zeros(1, dtype=torch.int32, device=plane.device)
    capi.call("vvcgpu_picture_hash", method, pp, sp, w, h, bit_depth, capi.ptr(out), _stream())
    return out


# ---- N4 intra sample prediction ----------------------------------------------------------------------------
INTRA_DESC = np.dtype([("ref_off", "<i8"), ("dst_off", "<i8"), ("dst_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("mode", "i1"),
                       ("filter_refs", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
assert INTRA_DESC.itemsize == 32


def intra_ref_lengths(w, h):
    t, l = C.c_int(), C.c_int()
    capi.call("vvcgpu_intra_ref_lengths", w, h, C.byref(t), C.byref(l))
    return t.value, l.value


INTRA_SATD_DESC = np.dtype([("ref_off", "<i8"), ("org_off", "<i8"), ("org_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("mode", "i1"),
                            ("filter_refs", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
assert INTRA_SATD_DESC.itemsize == 32


def intra_satd_batch(refs_base, org_base, descs_dev, n, clp=(0, 1023)):
    """N4: intra mode pre-selection -- Hadamard distortion of predIntraAng(mode) against the original, one value per (block, mode)."""
    out = torch.empty(n, dtype=torch.int64, device=org_base.device)
    capi.call("vvcgpu_intra_satd_batch", capi.ptr(refs_base), capi.ptr(org_base), capi.ptr(descs_dev), n, clp[0], clp[1], capi.ptr(out), _stream())
    return out


def intra_pred_batch(refs_base, dst_base, descs_dev, n, clp=(0, 1023)):
    """N4: IntraPrediction::predIntraAng for n blocks (packed reference samples in, prediction blocks out)."""
    capi.call("vvcgpu_intra_pred_batch", capi.ptr(refs_base), capi.ptr(dst_base), capi.ptr(descs_dev), n, clp[0], clp[1], _stream())


CCLM_DESC = np.dtype([("luma_off", "<i8"), ("nb_off", "<i8"), ("dst_off", "<i8"), ("luma_stride", "<i4"), ("dst_stride", "<i4"), ("w", "<i2"),
                      ("h", "<i2"), ("above_avail", "i1"), ("left_avail", "i1"), ("reserved", "<i2"), ("reserved2", "<i4", (2,))])
assert CCLM_DESC.itemsize == 48


def cclm_pred_batch(luma_base, nb_base, dst_base, descs_dev, n, bd_luma=10, bd_chroma=10, clp=(0, 1023)):
    """N4: CCLM chroma prediction (xGetLumaRecPixels + xGetLMParameters + predIntraChromaLM) for n chroma blocks."""
    capi.call("vvcgpu_cclm_pred_batch", capi.ptr(luma_base), capi.ptr(nb_base), capi.ptr(dst_base), capi.ptr(descs_dev), n, bd_luma, bd_chroma,
              clp[0], clp[1], _stream())


INTRA_FILL_DESC = np.dtype([("rec_off", "<i8"), ("flags_off", "<i8"), ("ref_off", "<i8"), ("rec_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                            ("unit_w", "i1"), ("unit_h", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
assert INTRA_FILL_DESC.itemsize == 40


def intra_fill_refs_batch(rec_base, flags_base, refs_base, descs_dev, n, bit_depth=10):
    """N4: xFillReferenceSamples for n blocks: reconstruction + unit availability flags -> packed reference samples."""
    capi.call("vvcgpu_intra_fill_refs_batch", capi.ptr(rec_base), capi.ptr(flags_base), capi.ptr(refs_base), capi.ptr(descs_dev), n, bit_depth, _stream())


IMV_PU = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4"),
                   ("cand_x", "<i4", (2,)), ("cand_y", "<i4", (2,)), ("pos_x", "<i4"), ("pos_y", "<i4"), ("idx_cost", "<u4", (2,)), ("bits", "<u4"),
                   ("w", "<i2"), ("h", "<i2"), ("num_cand", "i1"), ("mvp_idx", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
IMV_RESULT = np.dtype([("mv_x", "<i4"), ("mv_y", "<i4"), ("mvp_idx", "<i4"), ("bits", "<u4"), ("cost", "<u8")])
assert IMV_PU.itemsize == 72 and IMV_RESULT.itemsize == 24


def imv_refine_batch(org, ref, pus_dev, n, cfg, use_hadamard=True, weight=1.0):
    """N2 (AMVR): xPatternSearchIntRefine for n PUs -> IMV_RESULT records (uint8 tensor)."""
    po, so, _, _ = _plane(org, "org")
    pr, sr, _, _ = _plane(ref, "ref")
    cfg = np.ascontiguousarray(cfg)
    assert cfg.dtype == TZ_CFG and cfg.size == 1
    out = torch.empty(n * IMV_RESULT.itemsize, dtype=torch.uint8, device=org.device)
    capi.call("vvcgpu_imv_refine_batch", po, so, pr, sr, capi.ptr(pus_dev), n, C.c_void_p(cfg.ctypes.data), 1 if use_hadamard else 0, C.c_double(weight),
              capi.ptr(out), _stream())
    return out


QUANT_DESC = np.dtype([("coeff_off", "<i8"), ("level_off", "<i8"), ("w", "<i2"), ("h", "<i2"), ("intra_slice", "i1"), ("sign_hiding", "i1"),
                       ("reserved", "<i2"), ("qp", "<i4"), ("reserved2", "<i4")])
assert QUANT_DESC.itemsize == 32


def quant_batch(coeff_base, level_base, descs_dev, n, bit_depth=10):
    """N1 forward: Quant::quant without RDOQ (+ sign bit hiding) for n TUs -> abs-sum int32 tensor [n] (bits as uint32)."""
    out = torch.zeros(n, dtype=torch.int32, device=coeff_base.device)
    capi.call("vvcgpu_quant_batch", capi.ptr(coeff_base), capi.ptr(level_base), capi.ptr(descs_dev), n, bit_depth, capi.ptr(out), _stream())
    return out


DQ_RATES = np.dtype([("last_x", "<i4", (64,)), ("last_y", "<i4", (64,)), ("sig_sbb", "<i4", (2, 2)), ("sig", "<i4", (3, 18, 2)), ("gtx", "<i4", (21, 7))])
DEPQUANT_DESC = np.dtype([("coeff_off", "<i8"), ("level_off", "<i8"), ("lambda", "<f8"), ("qp", "<i4"), ("rates_idx", "<i4"), ("w", "<i2"), ("h", "<i2"),
                          ("luma", "i1"), ("reserved", "i1", (3,))])
assert DEPQUANT_DESC.itemsize == 40 and DQ_RATES.itemsize == 4 * (128 + 4 + 108 + 147)


def depquant_batch(coeff_base, level_base, descs_dev, n, rates_dev, total_coeffs, bit_depth=10):
    """N1: dependent-quantisation trellis (DQIntern::DepQuant::quant) for n TUs -> abs-sum int32 tensor [n] (bits as uint32)."""
    lib = capi.lib()
    lib.vvcgpu_depquant_workspace_bytes.restype = C.c_size_t
    nbytes = int(lib.vvcgpu_depquant_workspace_bytes(C.c_size_t(total_coeffs), n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=coeff_base.device)
    out = torch.zeros(n, dtype=torch.int32, device=coeff_base.device)
    capi.call("vvcgpu_depquant_batch", capi.ptr(coeff_base), capi.ptr(level_base), capi.ptr(descs_dev), n, capi.ptr(rates_dev), bit_depth,
              capi.ptr(out), C.c_size_t(total_coeffs), capi.ptr(ws), C.c_size_t(nbytes), _stream())
    return out


RDOQ_RATES = np.dtype([("sig", "<i4", (18, 2)), ("par", "<i4", (21, 2)), ("gt1", "<i4", (21, 2)), ("gt2", "<i4", (21, 2)), ("sig_group", "<i4", (2, 2)),
                       ("last_x", "<i4", (14,)), ("last_y", "<i4", (14,)), ("cbf", "<i4", (2,))])
RDOQ_DESC = np.dtype([("coeff_off", "<i8"), ("level_off", "<i8"), ("lambda", "<f8"), ("qp", "<i4"), ("rates_idx", "<i4"), ("w", "<i2"), ("h", "<i2"),
                      ("luma", "i1"), ("sign_hiding", "i1"), ("reserved", "i1", (2,))])
assert RDOQ_DESC.itemsize == 40 and RDOQ_RATES.itemsize == 784


def rdoq_batch(coeff_base, level_base, descs_dev, n, rates_dev, total_coeffs, bit_depth=10):
    """N1: rate-distortion optimised quantiser (QuantRDOQ::xRateDistOptQuant) for n TUs -> abs-sum int32 tensor [n] (bits as uint32)."""
    lib = capi.lib()
    lib.vvcgpu_rdoq_workspace_bytes.restype = C.c_size_t
    nbytes = int(lib.vvcgpu_rdoq_workspace_bytes(C.c_size_t(total_coeffs), n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=coeff_base.device)
    out = torch.zeros(n, dtype=torch.int32, device=coeff_base.device)
    capi.call("vvcgpu_rdoq_batch", capi.ptr(coeff_base), capi.ptr(level_base), capi.ptr(descs_dev), n, capi.ptr(rates_dev), bit_depth,
              capi.ptr(out), C.c_size_t(total_coeffs), capi.ptr(ws), C.c_size_t(nbytes), _stream())
    return out


# ---- interpolation / MC / PelBuffer ops -------------------------------------------------------------
IF_DESC = np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("src_stride", "<i4"), ("dst_stride", "<i4"),
                    ("w", "<i2"), ("h", "<i2"), ("taps", "i1"), ("is_vertical", "i1"), ("is_first", "i1"),
                    ("is_last", "i1"), ("coeff", "<i2", (8,)), ("reserved", "<i2", (4,))])
MC_DESC = np.dtype([("ref0_off", "<i8"), ("ref1_off", "<i8"), ("dst_off", "<i8"), ("ref0_stride", "<i4"),
                    ("ref1_stride", "<i4"), ("dst_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("frac_x0", "i1"),
                    ("frac_y0", "i1"), ("frac_x1", "i1"), ("frac_y1", "i1"), ("is_luma", "i1"), ("bi", "i1"),
                    ("reserved", "<i2")])
PELOP_DESC = np.dtype([("src0_off", "<i8"), ("src1_off", "<i8"), ("dst_off", "<i8"), ("src0_stride", "<i4"),
                       ("src1_stride", "<i4"), ("dst_stride", "<i4"), ("w", "<i2"), ("h", "<i2")])
assert IF_DESC.itemsize == 56 and MC_DESC.itemsize == 48 and PELOP_DESC.itemsize == 40


class PelopCfg(C.Structure):
    _fields_ = [("scale", C.c_int32), ("shift", C.c_int32), ("offset", C.c_int32), ("clip", C.c_int32),
                ("clp_min", C.c_int32), ("clp_max", C.c_int32)]


def if_batch(src_base, dst_base, descs_dev, n, bit_depth=10, clp=(0, 1023)):
    capi.call("vvcgpu_if_batch", capi.ptr(src_base), capi.ptr(dst_base), capi.ptr(descs_dev), n, bit_depth, clp[0], clp[1], _stream())


def mc_batch(ref0_base, ref1_base, dst_base, descs_dev, n, bit_depth=10, clp=(0, 1023)):
    capi.call("vvcgpu_mc_batch", capi.ptr(ref0_base), capi.ptr(ref1_base), capi.ptr(dst_base), capi.ptr(descs_dev), n,
              bit_depth, clp[0], clp[1], _stream())


def mc_picture_batch(ref0_base, ref1_base, dst_base, descs_dev, n, bit_depth=10, clp=(0, 1023)):
    """mc_batch for a picture's list of (mostly) 16x16 luma / 8x8 chroma PUs: one launch (vvcgpu_mc_picture_batch)"""
    capi.call("vvcgpu_mc_picture_batch", capi.ptr(ref0_base), capi.ptr(ref1_base), capi.ptr(dst_base), capi.ptr(descs_dev), n,
              bit_depth, clp[0], clp[1], _stream())


def mc_dist_batch(kind, ref0_base, ref1_base, org_base, descs_dev, n, bit_depth=10, clp=(0, 1023)):
    """predict a candidate (descriptors as mc_batch, dst_off / dst_stride = the original block, reserved = SAD row sub-sampling shift) and return
    its distortion against the original: int64 tensor [n]."""
    out = torch.empty(n, dtype=torch.int64, device=org_base.device)
    capi.call("vvcgpu_mc_dist_batch", kind, capi.ptr(ref0_base), capi.ptr(ref1_base), capi.ptr(org_base), capi.ptr(descs_dev), n, bit_depth,
              clp[0], clp[1], capi.ptr(out), _stream())
    return out


def pelop_batch(op, src0_base, src1_base, dst_base, descs_dev, n, cfg):
    capi.call("vvcgpu_pelop_batch", op, capi.ptr(src0_base), capi.ptr(src1_base), capi.ptr(dst_base), capi.ptr(descs_dev), n,
              C.byref(cfg), _stream())


# ---- transforms (TrQuant) ----------------------------------------------------------------------------
TR_DESC = np.dtype([("resi_off", "<i8"), ("coeff_off", "<i8"), ("resi_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                    ("tr_hor", "i1"), ("tr_ver", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
assert TR_DESC.itemsize == 32
DCT2, DCT8, DST7, TSKIP = 0, 1, 2, 3


def tr_fwd_batch(resi_base, coeff_base, descs_dev, n, bit_depth=10):
    capi.call("vvcgpu_tr_fwd_batch", capi.ptr(resi_base), capi.ptr(coeff_base), capi.ptr(descs_dev), n, bit_depth, _stream())


def tr_inv_batch(coeff_base, resi_base, descs_dev, n, bit_depth=10):
    capi.call("vvcgpu_tr_inv_batch", capi.ptr(coeff_base), capi.ptr(resi_base), capi.ptr(descs_dev), n, bit_depth, _stream())


# ---- fused fractional refinement (InterSearch::xPatternSearchFracDIF) ---------------------------------------
FRAC_BLK = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4")])
FRAC_RESULT = np.dtype([("half_x", "<i4"), ("half_y", "<i4"), ("qter_x", "<i4"), ("qter_y", "<i4"), ("cost_half", "<u8"), ("cost", "<u8")])


AFG_DESC = np.dtype([("pred_off", "<i8"), ("deriv_off", "<i8"), ("pred_stride", "<i4"), ("deriv_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                     ("reserved", "<i4")])
AFE_DESC = np.dtype([("resi_off", "<i8"), ("deriv_off", "<i8"), ("deriv_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("six_param", "<i4"),
                     ("reserved", "<i4")])
assert AFG_DESC.itemsize == 32 and AFE_DESC.itemsize == 32


def affine_sobel_batch(vertical, pred_base, deriv_base, descs_dev, n):
    """N3: Sobel derivative planes of prediction blocks (table slots m_HorizontalSobelFilter / m_VerticalSobelFilter)."""
    capi.call("vvcgpu_affine_sobel_batch", int(vertical), capi.ptr(pred_base), capi.ptr(deriv_base), capi.ptr(descs_dev), n, _stream())


def affine_equal_coeff_batch(resi_base, gx_base, gy_base, descs_dev, n):
    """N3: normal-equation sums of the affine model (table slot m_EqualCoeffComputer) -> int64 tensor [n, 7, 7]."""
    out = torch.empty((n, 7, 7), dtype=torch.int64, device=resi_base.device)
    capi.call("vvcgpu_affine_equal_coeff_batch", capi.ptr(resi_base), capi.ptr(gx_base), capi.ptr(gy_base), capi.ptr(descs_dev), n,
              capi.ptr(out), _stream())
    return out


DQTR_DESC = np.dtype([("resi_off", "<i8"), ("level_off", "<i8"), ("resi_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                      ("tr_hor", "i1"), ("tr_ver", "i1"), ("dep_quant", "i1"), ("reserved", "i1"), ("qp", "<i4")])
assert DQTR_DESC.itemsize == 32


def dequant_tr_inv_batch(level_base, resi_base, descs_dev, n, bit_depth, coeff_out=None):
    """N1: de-quantisation + inverse transform in one launch.  coeff_out (optional): int32, same offsets as level_base, receives the de-quantised
    coefficients (the reference's m_plTempCoeff); None: they never leave the chip."""
    capi.call("vvcgpu_dequant_tr_inv_batch", capi.ptr(level_base), capi.ptr(resi_base), capi.ptr(descs_dev), n, bit_depth,
              capi.ptr(coeff_out) if coeff_out is not None else None, _stream())


def frac_refine(org, ref, blocks_dev, nblocks, w, h, bit_depth, mvcost, use_hadamard=True, clp=(0, 1023)):
    """I2+D2+D5: returns a uint8 CUDA tensor holding nblocks FRAC_RESULT records."""
    po, so, _, _ = _plane(org, "org")
    pr, sr, _, _ = _plane(ref, "ref")
    res = torch.empty(nblocks * FRAC_RESULT.itemsize, dtype=torch.uint8, device=org.device)
    capi.call("vvcgpu_frac_refine", po, so, pr, sr, capi.ptr(blocks_dev), nblocks, w, h, bit_depth, clp[0], clp[1],
              1 if use_hadamard else 0, C.byref(mvcost), capi.ptr(res), _stream())
    return res


# ---- fused residual chain (InterSearch::xEstimateInterResidualQT per TU) ------------------------------------
RC_DESC = np.dtype([("org_off", "<i8"), ("pred_off", "<i8"), ("rec_off", "<i8"), ("level_off", "<i8"), ("org_stride", "<i4"), ("pred_stride", "<i4"),
                    ("rec_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("tr_hor", "i1"), ("tr_ver", "i1"), ("intra_slice", "i1"), ("sign_hiding", "i1"),
                    ("qp", "<i4"), ("reserved", "<i4", (2,))])
assert RC_DESC.itemsize == 64


def resi_chain_batch(org_base, pred_base, rec_base, level_base, descs_dev, n, bit_depth=10, clp=(0, 1023)):
    """subtract -> forward transform -> Quant::quant -> Quant::dequant -> inverse transform -> reconstruction of n TUs in one pass.
    Returns the abs-sum int32 tensor [n] (bits as uint32; 0xFFFFFFFF marks a TU outside the entry point's preconditions)."""
    out = torch.empty(n, dtype=torch.int32, device=org_base.device)       # every entry is written by the library
    capi.call("vvcgpu_resi_chain_batch", capi.ptr(org_base), capi.ptr(pred_base), capi.ptr(rec_base), capi.ptr(level_base), capi.ptr(descs_dev), n,
              bit_depth, clp[0], clp[1], capi.ptr(out), _stream())
    return out


def resi_chain_runs_batch(org_base, pred_base, rec_base, level_base, descs_dev, n, runs, bit_depth=10, clp=(0, 1023)):
    """the same for descriptors grouped by shape: runs = [(w, h, count), ...] in the order of the list (vvcgpu_resi_chain_runs_batch)"""
    r = np.ascontiguousarray(np.asarray(runs, dtype=np.int32).reshape(-1, 3))
    out = torch.empty(n, dtype=torch.int32, device=org_base.device)
    capi.call("vvcgpu_resi_chain_runs_batch", capi.ptr(org_base), capi.ptr(pred_base), capi.ptr(rec_base), capi.ptr(level_base), capi.ptr(descs_dev), n,
              C.c_void_p(r.ctypes.data), int(r.shape[0]), bit_depth, clp[0], clp[1], capi.ptr(out), _stream())
    return out


# ---- picture-level forms of the in-loop entry points (three planes, one launch each) ------------------------
class Planes(C.Structure):
    """vvcgpu_planes"""
    _fields_ = [("p", C.c_void_p * 3), ("stride", C.c_int32 * 3)]


def planes(ts):
    """three 2-D int16 CUDA tensors (Y, Cb, Cr) -> vvcgpu_planes"""
    pl = Planes()
    for i, t in enumerate(ts):
        assert t.is_cuda and t.dtype == torch.int16 and t.dim() == 2 and t.stride(1) == 1
        pl.p[i] = t.data_ptr()
        pl.stride[i] = t.stride(0)
    return pl


def sao_apply_picture(src, dst, ctu, bit_depth, params_dev, clp=(0, 1023)):
    h, w = src[0].shape
    capi.call("vvcgpu_sao_apply_picture", C.byref(planes(src)), C.byref(planes(dst)), w, h, ctu, bit_depth, capi.ptr(params_dev[0]),
              capi.ptr(params_dev[1]), capi.ptr(params_dev[2]), clp[0], clp[1], _stream())
    return dst


def sao_stats_picture(org, rec, ctu, bit_depth, avail=None, skip_luma=(5, 4), skip_chroma=(3, 2)):
    """-> [int64 tensor (nCtu, 5, 2, 32)] x 3"""
    h, w = org[0].shape
    n = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
    outs = [torch.empty((n, 5, 2, 32), dtype=torch.int64, device=org[0].device) for _ in range(3)]
    capi.call("vvcgpu_sao_stats_picture", C.byref(planes(org)), C.byref(planes(rec)), w, h, ctu, bit_depth, capi.ptr(avail), skip_luma[0], skip_luma[1],
              skip_chroma[0], skip_chroma[1], capi.ptr(outs[0]), capi.ptr(outs[1]), capi.ptr(outs[2]), _stream())
    return outs


def alf_filter_picture(src, dst, ctu, cls, filter_type, luma_coeff, chroma_coeff, enable=(None, None, None), clp=(0, 1023)):
    h, w = src[0].shape
    lc = np.ascontiguousarray(luma_coeff, dtype=np.int16)
    cc = np.ascontiguousarray(chroma_coeff, dtype=np.int16)
    assert lc.size == 25 * 13 and cc.size == 7
    capi.call("vvcgpu_alf_filter_picture", C.byref(planes(src)), C.byref(planes(dst)), w, h, ctu, capi.ptr(cls), filter_type,
              C.c_void_p(lc.ctypes.data), C.c_void_p(cc.ctypes.data), capi.ptr(enable[0]), capi.ptr(enable[1]), capi.ptr(enable[2]), clp[0], clp[1], _stream())
    return dst


def alf_stats_picture(org, rec, ctu, cls):
    """-> (luma 7x7 (nCtu, 25, 183), luma 5x5 (nCtu, 25, 57), [Cb (nCtu, 1, 57), Cr (nCtu, 1, 57)]) int64 tensors"""
    h, w = org[0].shape
    n = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
    dev = org[0].device
    a7 = torch.empty((n, 25, 183), dtype=torch.int64, device=dev)
    a5 = torch.empty((n, 25, 57), dtype=torch.int64, device=dev)
    ac = [torch.empty((n, 1, 57), dtype=torch.int64, device=dev) for _ in range(2)]
    capi.call("vvcgpu_alf_stats_picture", C.byref(planes(org)), C.byref(planes(rec)), w, h, ctu, capi.ptr(cls), capi.ptr(a7), capi.ptr(a5),
              capi.ptr(ac[0]), capi.ptr(ac[1]), _stream())
    return a7, a5, ac


def alf_classify_stats_picture(org, rec, ctu, bit_depth):
    """The encoder's ALF front end in one launch (vvcgpu_alf_classify_stats_picture): -> (cls as alf_classify returns it, then alf_stats_picture's tuple)"""
    h, w = org[0].shape
    n = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
    dev = org[0].device
    cls = torch.empty((h // 4, w // 4), dtype=torch.int16, device=dev)
    a7 = torch.empty((n, 25, 183), dtype=torch.int64, device=dev)
    a5 = torch.empty((n, 25, 57), dtype=torch.int64, device=dev)
    ac = [torch.empty((n, 1, 57), dtype=torch.int64, device=dev) for _ in range(2)]
    capi.call("vvcgpu_alf_classify_stats_picture", C.byref(planes(org)), C.byref(planes(rec)), w, h, ctu, bit_depth, capi.ptr(cls), capi.ptr(a7), capi.ptr(a5),
              capi.ptr(ac[0]), capi.ptr(ac[1]), _stream())
    return cls, a7, a5, ac


# ---- T3 residual DPCM, I3 affine sub-block vectors -------------------------------------------------------------
RDPCM_DESC = np.dtype([("resi_off", "<i8"), ("coeff_off", "<i8"), ("resi_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("mode", "i1"), ("lossless", "i1"),
                       ("rotate", "i1"), ("intra_slice", "i1"), ("qp", "<i4"), ("reserved", "<i4"), ("pad", "<i4")])
AFFINE_PU = np.dtype([("pos_x", "<i4"), ("pos_y", "<i4"), ("w", "<i2"), ("h", "<i2"), ("six_param", "<i2"), ("bi", "<i2"), ("mv", "<i4", (2, 3, 2)),
                      ("dst_off", "<i8"), ("dst_stride", "<i4"), ("first_desc", "<i4")])
MC_DESC = np.dtype([("ref0_off", "<i8"), ("ref1_off", "<i8"), ("dst_off", "<i8"), ("ref0_stride", "<i4"), ("ref1_stride", "<i4"), ("dst_stride", "<i4"),
                    ("w", "<i2"), ("h", "<i2"), ("frac_x0", "i1"), ("frac_y0", "i1"), ("frac_x1", "i1"), ("frac_y1", "i1"), ("is_luma", "i1"), ("bi", "i1"),
                    ("reserved", "<i2")])
assert RDPCM_DESC.itemsize == 40 and AFFINE_PU.itemsize == 80 and MC_DESC.itemsize == 48


def rdpcm_fwd_batch(resi_base, coeff_base, descs_dev, n, bit_depth=10):
    """TrQuant::applyForwardRDPCM for n TUs -> abs-sum int32 tensor [n] (bits as uint32)"""
    out = torch.zeros(n, dtype=torch.int32, device=resi_base.device)
    capi.call("vvcgpu_rdpcm_fwd_batch", capi.ptr(resi_base), capi.ptr(coeff_base), capi.ptr(descs_dev), n, bit_depth, capi.ptr(out), _stream())
    return out


def rdpcm_inv_batch(resi_base, descs_dev, n):
    """TrQuant::invRdpcmNxN, in place"""
    capi.call("vvcgpu_rdpcm_inv_batch", capi.ptr(resi_base), capi.ptr(descs_dev), n, _stream())


AFFINE_ITER = np.dtype([("pu", AFFINE_PU), ("org_off", "<i8"), ("org_stride", "<i4"), ("reserved", "<i4")])
assert AFFINE_ITER.itemsize == 96


def affine_me_iter_batch(org_base, ref_base, pred_base, items_dev, n, n_subblocks, dist_kind, pic_w, pic_h, ref_origin, ref_stride,
                         bit_depth=10, clp=(0, 1023), max_cu=128, want_dist=True):
    """one iteration of the affine gradient search behind its prediction (loop body of xAffineMotionEstimation): the prediction is left in
    pred_base; -> (equation sums int64 [n, 7, 7], distortion int64 [n] or None)"""
    ws = torch.empty(n_subblocks * MC_DESC.itemsize, dtype=torch.uint8, device=org_base.device)
    coeff = torch.empty((n, 7, 7), dtype=torch.int64, device=org_base.device)
    dist = torch.empty(n, dtype=torch.int64, device=org_base.device) if want_dist else None
    capi.call("vvcgpu_affine_me_iter_batch", capi.ptr(org_base), capi.ptr(ref_base), capi.ptr(pred_base), capi.ptr(items_dev), n, n_subblocks,
              capi.ptr(ws), dist_kind, pic_w, pic_h, max_cu, max_cu, ref_origin[0], ref_origin[1], ref_stride, bit_depth, clp[0], clp[1],
              capi.ptr(coeff), capi.ptr(dist) if want_dist else None, _stream())
    return coeff, dist


def affine_pred_batch(ref0_base, ref1_base, dst_base, pus_dev, n, n_subblocks, comp, pic_w, pic_h, ref_origin, ref0_stride, ref1_stride,
                      bit_depth=10, clp=(0, 1023), max_cu=128):
    """xPredAffineBlk for a list of PUs: sub-block vectors + their interpolation in one call (luma: four 4x4 sub-blocks per wavefront)"""
    ws = torch.empty(n_subblocks * MC_DESC.itemsize, dtype=torch.uint8, device=dst_base.device)
    capi.call("vvcgpu_affine_pred_batch", capi.ptr(ref0_base), capi.ptr(ref1_base), capi.ptr(dst_base), capi.ptr(pus_dev), n, n_subblocks, capi.ptr(ws),
              comp, pic_w, pic_h, max_cu, max_cu, ref_origin[0], ref_origin[1], ref0_stride, ref1_stride, bit_depth, clp[0], clp[1], _stream())


def affine_subblock_descs(pus_dev, n, n_descs, comp, pic_w, pic_h, ref_origin, ref0_stride, ref1_stride, max_cu=128):
    """sub-block MC descriptors (uint8 tensor of n_descs vvcgpu_mc_desc) of n affine PUs, on the device"""
    out = torch.zeros(n_descs * MC_DESC.itemsize, dtype=torch.uint8, device=pus_dev.device)
    capi.call("vvcgpu_affine_subblock_descs", capi.ptr(pus_dev), n, comp, pic_w, pic_h, max_cu, max_cu, ref_origin[0], ref_origin[1], ref0_stride,
              ref1_stride, capi.ptr(out), _stream())
    return out
