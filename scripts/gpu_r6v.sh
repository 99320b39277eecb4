#!/bin/bash
mkdir -p gpurun_out/r6v
timeout 300 scripts/ubench/attn_pack 2>&1 | tee gpurun_out/r6v/attn_pack.txt
