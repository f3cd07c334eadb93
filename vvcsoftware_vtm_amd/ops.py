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
