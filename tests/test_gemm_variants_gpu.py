"""The 8-wave GEMM picks its m-tiles per wave (MT = 2 / 3 / 4) from the grid; the variant is latched per process
(GC_GEMM_MT), so every variant is forced over the whole linear / conv / GEGLU parity suite in a child process.  The children run
CONCURRENTLY (they are light on the GPU and mostly wait for their CPU references): the test costs the slowest child, not the sum."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [{"GC_GEMM_MT": "2"}, {"GC_GEMM_MT": "3"}, {"GC_GEMM_MT": "4"}, {"GC_GEMM8": "0"}, {"GC_ATTN_SAFE": "1"},
            {"GC_GEMM_DBG": "16"}]          # (16 = kernel_variant 0x1000: the 3 x 3 convolutions on the tap-outer k order of rounds 1-5; default since round 6: tap-inner)


def test_forced_kernel_variants():
    procs = []
    for env in VARIANTS:
        e = dict(os.environ); e.update(env)
        e["OMP_NUM_THREADS"] = e["MKL_NUM_THREADS"] = "8"          # the children share the host
        sel = "attention" if "GC_ATTN_SAFE" in env else ("conv" if "GC_GEMM_DBG" in env else "linear or geglu or conv")
        log = tempfile.TemporaryFile(mode="w+")        # (a file, not a pipe: nobody drains five pipes at once)
        procs.append((env, log, subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_denoise_kernels_gpu.py"), "-x", "-q",
                                                  "-p", "no:cacheprovider", "-k", sel], cwd=ROOT, env=e, stdout=log, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for env, log, p in procs:
        try:
            p.wait(timeout=900)
            note = ""
        except subprocess.TimeoutExpired:
            p.kill(); p.wait()
            note = "\n[timed out]"
        log.seek(0)
        out = log.read() + note
        log.close()
        tail = out.strip().splitlines()[-1] if out.strip() else ""
        print(f"{env}: rc {p.returncode}  {tail}")
        if p.returncode != 0:
            failed.append((env, out[-3000:]))
    assert not failed, failed
