#!/bin/bash
# round 6, call H: tap-inner k order of the fast 3x3 convs -- kernel tests, then same-box A/B (GC_GEMM_DBG=16 = kernel_variant 0x1000 = tap-outer, rounds 1-5) at cobatch 4
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_gemm_variants_gpu.py -m gpu -q -x > gpurun_out/r6h/tests.txt 2>&1
tail -3 gpurun_out/r6h/tests.txt
for v in 16 0 16 0; do
  GC_GEMM_DBG=$v GC_BENCH_SHAPES=1 timeout 400 python bench.py --steps 28 --warmup 14 --no-secondary --no-cpu-baseline > gpurun_out/r6h/bench_dbg$v.json 2> gpurun_out/r6h/shapes_dbg$v.txt
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6h/bench_dbg$v.json") if l.startswith("{")][-1])
print("GC_GEMM_DBG=$v:", d["value"], "views/s", d["ms_per_step"], "ms/step", "mfma_util", d["mfma_util_step"])
for k,v in d["roofline"]["other"].items():
    if "conv" in k: print("   ", k, v)
PY
done
grep "^# shape conv" gpurun_out/r6h/shapes_dbg16.txt | head -12 | cut -c1-160
grep "^# shape conv" gpurun_out/r6h/shapes_dbg0.txt | head -12 | cut -c1-160
