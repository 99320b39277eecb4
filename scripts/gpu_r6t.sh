#!/bin/bash
# round 6, call T: the new tests (grouped VAE decode, tap-outer conv variant child, 32-bit operand guard)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6t
timeout 1200 python -m pytest tests/test_denoise_model_gpu.py tests/test_gemm_variants_gpu.py tests/test_denoise_kernels_gpu.py -m gpu -q -x -k "decode or variants or refuses" > gpurun_out/r6t/tests.txt 2>&1
tail -5 gpurun_out/r6t/tests.txt
