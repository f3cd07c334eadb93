"""integration/vtm-2.1-hip.patch (the SIMD=HIP selector + source hooks a maintainer adds to VTM 2.1): regenerated from the reference tree it must be the
committed file, contain nothing but added lines, and apply cleanly.  CPU only; skipped where /root/reference does not exist (GPU box)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "vtm-2.1-hip.patch")


def test_patch_holds_only_added_lines():
    body = [l for l in open(PATCH).read().splitlines() if not l.startswith(("--- ", "+++ ", "@@ "))]
    assert body and all(l.startswith("+") for l in body)              # zero context, nothing removed: no reference text in the repository
    assert sum("vvcHipEnter(" in l for l in body) == 16 and sum("vvcshim_" in l for l in body) == 13


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "source", "Lib")), reason="reference tree not present")
def test_patch_is_reproducible_and_applies(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "integration", "make_patch.py"), REF], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(PATCH).read()
    files = sorted({l.split()[1][2:] for l in r.stdout.splitlines() if l.startswith("--- ")})
    assert len(files) == 14
    for f in files:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    a = subprocess.run(["patch", "-s", "-p1", "-d", str(tmp_path)], stdin=open(PATCH), capture_output=True, text=True)
    assert a.returncode == 0, a.stdout + a.stderr
    hooked = open(tmp_path / "source/Lib/CommonLib/x86/InitX86.cpp").read()
    assert hooked.count("vvcHipEnter(") == 5 and '#include "hip/InitHIP.h"' in hooked
