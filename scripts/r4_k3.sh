cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_gemm_variants_gpu.py -q -x 2>&1 | grep -v "^  x0" | tail -40
