set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_raster_views_gpu.py -m gpu -q -x 2>&1 | grep -v "^  x0" | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-230
