#!/bin/bash
# round 6, call S2: 8 x 8-map convs on 256-row k-sliced tiles -- GEMM tests, then same-box A/B (GC_GEMM_SPLIT_MT=3 = before)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s
mkdir -p $O
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_random_shapes_gpu.py -m gpu -q -x -k "conv or linear or geglu or statistics or partials or random" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for v in 3 0 3 0; do
  GC_GEMM_SPLIT_MT=$v GC_BENCH_SHAPES=1 timeout 400 python bench.py --steps 28 --warmup 14 --no-secondary --no-cpu-baseline > $O/bench_smt$v.json 2> $O/shapes_smt$v.txt
  python -c "
import json
d=json.loads([l for l in open('$O/bench_smt$v.json') if l.startswith('{')][-1])
print('GC_GEMM_SPLIT_MT=$v:', d['value'], 'views/s', d['ms_per_step'], [ (k[:28], x['ms']) for k,x in d['roofline']['other'].items() if 'conv3x3,BN=128' in k])
"
  grep "hw=8x8" $O/shapes_smt$v.txt | head -2 | cut -c1-150
done
