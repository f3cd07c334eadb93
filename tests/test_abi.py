"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/vvcgpu.h declares (no compute calls -- there is no GPU here)."""
import ctypes as C
import os
import subprocess

from vvcsoftware_vtm_amd import capi


def _ensure_built():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_header_symbols_exported():
    _ensure_built()
    lib = capi.lib()
    names = capi.declared_symbols()
    assert len(names) >= 8
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in vvcgpu.h but not exported: %s" % missing


def test_exports_are_c_abi_only():
    """no C++-mangled entry points leak out under the vvcgpu_ prefix and no torch symbol is needed."""
    _ensure_built()
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert all(not e.startswith("_Z") or "vvcgpu" not in e for e in exported if e.startswith("vvcgpu"))
    und = subprocess.check_output(["nm", "-D", "--undefined-only", capi.LIB_PATH], text=True)
    assert "torch" not in und and "c10" not in und


def test_version_and_error_text():
    _ensure_built()
    lib = capi.lib()
    assert lib.vvcgpu_version() == 1
    # argument validation happens before any device work, so it can be exercised without a GPU
    rc = lib.vvcgpu_alf_classify(None, 0, 0, 0, 10, None, None)
    assert rc == -1
    assert b"alf_classify" in lib.vvcgpu_last_error()
    rc = lib.vvcgpu_sao_apply(C.c_void_p(16), 8, C.c_void_p(16), 8, 8, 8, 8, 8, 10, C.c_void_p(16), 0, 1023, None)
    assert rc == -1 and b"alias" in lib.vvcgpu_last_error()


def test_struct_layouts_match_python_bindings():
    """numpy / ctypes mirrors used by the host code have exactly the C sizes (no GPU needed; torch is imported lazily)."""
    _ensure_built()
    import numpy as np
    from vvcsoftware_vtm_amd import ops
    lib = capi.lib()
    want = {0: ops.SAO_DTYPE.itemsize, 1: C.sizeof(ops.DeblockCfg), 2: ops.DIST_DESC.itemsize, 3: ops.SEARCH_BLK.itemsize,
            4: C.sizeof(ops.MvCost), 5: ops.SEARCH_BEST.itemsize, 6: ops.IF_DESC.itemsize, 7: ops.MC_DESC.itemsize,
            8: ops.PELOP_DESC.itemsize, 9: C.sizeof(ops.PelopCfg), 10: ops.TR_DESC.itemsize, 13: ops.DQTR_DESC.itemsize, 14: ops.AFG_DESC.itemsize, 15: ops.AFE_DESC.itemsize,
            11: ops.FRAC_BLK.itemsize, 12: ops.FRAC_RESULT.itemsize}
    for k, v in want.items():
        assert lib.vvcgpu_sizeof(k) == v, (k, lib.vvcgpu_sizeof(k), v)
    assert lib.vvcgpu_sizeof(99) == -1
