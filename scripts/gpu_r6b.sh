#!/bin/bash
# round 6, call B: the raster test files with the knife-edge-aware gradient check; per-shape GEMM table at cobatch 2
mkdir -p gpurun_out/r6b
python -m pytest tests/test_raster_views_gpu.py tests/test_raster_gpu.py tests/test_plugin_gpu.py -m gpu -q > gpurun_out/r6b/tests.txt 2>&1
tail -3 gpurun_out/r6b/tests.txt
GC_BENCH_SHAPES=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/shapes.txt
grep "^# shape" gpurun_out/r6b/shapes.txt | head -70
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6b/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['mfma_util_step'], d['roofline']['frac'], d['roofline_raster'])
"
