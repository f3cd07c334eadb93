"""ctypes binding of include/vvcgpu.h.  Fails loudly when the HIP library is missing or a call fails."""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "lib", "libvvcgpu.so")
HEADER = os.path.join(ROOT, "include", "vvcgpu.h")

_lib = None


class VvcGpuError(RuntimeError):
    pass


class SaoCtu(C.Structure):
    """vvcgpu_sao_ctu"""
    _fields_ = [("type", C.c_int8), ("avail", C.c_uint8), ("offset", C.c_int16 * 32)]


def declared_symbols():
    """Every function name include/vvcgpu.h declares (used by the ABI test)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vvcgpu_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VvcGpuError(
                "HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.vvcgpu_last_error.restype = C.c_char_p
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise VvcGpuError("%s failed (%d): %s" % (what, rc, lib().vvcgpu_last_error().decode()))


def call(name, *args):
    fn = getattr(lib(), name)
    check(fn(*args), name)


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
