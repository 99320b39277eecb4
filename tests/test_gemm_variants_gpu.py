"""The 8-wave GEMM picks its m-tiles per wave (MT = 2 / 3 / 4) from the grid; the variant is latched per process
(GC_GEMM_MT), so every variant is forced over the whole linear / conv / GEGLU parity suite in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"GC_GEMM_MT": "2"}, {"GC_GEMM_MT": "3"}, {"GC_GEMM_MT": "4"}, {"GC_GEMM8": "0"}, {"GC_ATTN_SAFE": "1"}])
def test_forced_kernel_variant(env):
    e = dict(os.environ); e.update(env)
    sel = "attention" if "GC_ATTN_SAFE" in env else "linear or geglu or conv"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_denoise_kernels_gpu.py"), "-x", "-q", "-k", sel],
                       cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
