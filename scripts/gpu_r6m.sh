#!/bin/bash
# round 6, call M: s_setprio 1 for waves 4..7 of k_gemm8 (GC_GEMM_DBG=32 = kernel_variant 0x2000) -- same-box A/B, twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m
mkdir -p $O
for v in 0 32 0 32; do
  GC_GEMM_DBG=$v timeout 400 python bench.py --steps 28 --warmup 14 --no-secondary --no-cpu-baseline > $O/bench_dbg$v.json 2> $O/bench_dbg$v.err
  python -c "
import json
d=json.loads([l for l in open('$O/bench_dbg$v.json') if l.startswith('{')][-1])
print('GC_GEMM_DBG=$v:', d['value'], 'views/s', d['ms_per_step'], d['mfma_util_step'])
for k,x in d['roofline']['other'].items():
    if 'gemm' in k and x['ms']>10: print('   ', k, x)
"
done
