cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dist_gpu.py -q -x -k allgather 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids\|diffusion weights" | head -80
