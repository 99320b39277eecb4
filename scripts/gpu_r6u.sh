#!/bin/bash
# round 6, call U: attention schedule microbenchmark (interleaved lockstep vs X | Y segments in phase vs phase-shifted)
mkdir -p gpurun_out/r6u
timeout 300 scripts/ubench/attn_pingpong 2>&1 | tee gpurun_out/r6u/attn_pingpong.txt
