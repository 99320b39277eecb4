#!/bin/bash
# round 6, call E: GEMM k-loop microbenchmark -- wave tile / wave count sweep with DMA / fragment-read ablations
mkdir -p gpurun_out/r6e
timeout 300 scripts/ubench/gemm_loop 5760 > gpurun_out/r6e/gemm_loop_k5760.txt 2>&1
cat gpurun_out/r6e/gemm_loop_k5760.txt | cut -c1-250
timeout 300 scripts/ubench/gemm_loop 2880 > gpurun_out/r6e/gemm_loop_k2880.txt 2>&1
cat gpurun_out/r6e/gemm_loop_k2880.txt | cut -c1-250
