#!/bin/bash
# round 6, call C: the gather-free depth order -- parity tests, then raster-only A/B (pair chain vs sorted-boxes chain) at 1 M and 4 M on one box
mkdir -p gpurun_out/r6c
python -m pytest tests/test_raster_views_gpu.py tests/test_raster_gpu.py -m gpu -q -x > gpurun_out/r6c/tests.txt 2>&1
tail -3 gpurun_out/r6c/tests.txt
for n in 1000000 4000000; do
  for sb in 0 1; do
    GC_RASTER_SORTED_BOXES=$sb python bench.py --workload raster --gaussians $n --steps 32 --warmup 2 --no-cpu-baseline > gpurun_out/r6c/raster_${n}_sb$sb.json 2> gpurun_out/r6c/raster_${n}_sb$sb.err
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6c/raster_${n}_sb$sb.json") if l.startswith("{")][-1])
r=d["roofline"]
print("N=$n sorted_boxes=$sb:", d["value"], "views/s; chain us/view", r["chain"]["kernel_us_per_view"], "frac", r["chain"]["frac"], "8d", r["chain"]["frac_8d_unbatched"])
print("   ", {k:v["avg_us"] for k,v in r["stages"].items()})
PY
  done
done
