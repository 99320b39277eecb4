#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
timeout 600 python scripts/attn_variant_scan_sets.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6s/attn_variant_scan.txt
