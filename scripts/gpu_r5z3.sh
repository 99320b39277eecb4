set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5z3}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_plugin_gpu.py -m gpu -q -k "attention" 2>&1 | grep -v "^  x0" | tail -30 > $O/tests_a.log; tail -6 $O/tests_a.log
