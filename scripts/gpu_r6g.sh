#!/bin/bash
# round 6, call G: GEMM loop microbenchmark with the 3x3-conv access pattern: tap-outer (product order) vs tap-inner k order, W through L1 or not
mkdir -p gpurun_out/r6g
timeout 300 scripts/ubench/gemm_loop 2880 conv > gpurun_out/r6g/gemm_loop_conv_k2880.txt 2>&1
cut -c1-250 gpurun_out/r6g/gemm_loop_conv_k2880.txt
timeout 300 scripts/ubench/gemm_loop 5760 conv > gpurun_out/r6g/gemm_loop_conv_k5760.txt 2>&1
cut -c1-250 gpurun_out/r6g/gemm_loop_conv_k5760.txt
