"""CPU: the oracle restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's sanitizer builds; GPU
sanitizers are not available on the pool, so the CPU-side code is what gets this check).  oracle/liboracle_asan.so is built by
`make -C oracle asan`; the golden-vector tests then run against it in a child interpreter with the sanitizer runtimes preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_golden_vectors_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ORACLE_SO="liboracle_asan.so", LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=24")
    # a subset that touches every restatement file and finishes in about a minute under the sanitizers
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
