"""The ONE behaviour switch of the library, VVCGPU_NO_MFMA=1 (csrc/common.h): interpolation, Hadamard refinement and the transform entries stay off the
matrix cores and their vector-pipe bodies -- which otherwise only serve flagged PUs / TUs -- take every PU / TU.  The parity cases of those bodies run in ONE child process with it set (every non-default path has an oracle test)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vector_pipe_bodies_match_the_oracle():
    env = dict(os.environ, VVCGPU_NO_MFMA="1")
    cases = ["tests/test_gpu_frac.py::test_frac16_every_alignment", "tests/test_gpu_frac.py::test_frac16_hadamard_at_the_int16_bound",
             "tests/test_gpu_interp.py::test_mc_every_phase", "tests/test_gpu_interp.py::test_mc_batch_few_pus",
             "tests/test_gpu_transform.py::test_tr_fwd_inv", "tests/test_gpu_transform.py::test_tr_many_tus_shuffled",
             "tests/test_gpu_transform.py::test_dequant_tr_inv", "tests/test_gpu_transform.py::test_dequant_tr_inv_long_homogeneous_batches"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + cases, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
