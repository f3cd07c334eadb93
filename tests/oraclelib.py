"""ctypes access to the CHECKERS: oracle/liboracle.so (CPU restatement) and, when built,
oracle/_ref/libvtmref.so (the compiled reference).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class SaoCtu(C.Structure):
    _fields_ = [("type", C.c_int8), ("avail", C.c_uint8), ("offset", C.c_int16 * 32)]


SAO_DTYPE = np.dtype([("type", "i1"), ("avail", "u1"), ("offset", "<i2", (32,))])
assert SAO_DTYPE.itemsize == C.sizeof(SaoCtu) == 66

_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, os.environ.get("ORACLE_SO", "liboracle.so"))     # ORACLE_SO: the sanitizer build (tests/test_oracle_asan.py)
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "restate"], stdout=subprocess.DEVNULL)
        _oracle = C.CDLL(so)
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libvtmref.so"))


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libvtmref.so"))
    return _ref


def p(a):
    """pointer to a numpy array's data (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"] or a.ndim == 0 or a.strides[-1] == a.itemsize
    return C.c_void_p(a.ctypes.data)
