#!/bin/bash
# round 6, call I: fp8 conv kernel with the tap-inner k order (tests + same-box A/B), PMC traffic of the sorted-boxes raster chain (1 M / 4 M, 8 views per launch set),
# raster-only bench lines that read it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i
mkdir -p $O
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_gemm_variants_gpu.py -m gpu -q -x -k "fp8 or e4m3 or conv" > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for v in 16 0; do
  GC_GEMM_DBG=$v timeout 400 python bench.py --dtype fp8 --steps 28 --warmup 14 --no-secondary --no-cpu-baseline > $O/bench_fp8_dbg$v.json 2> $O/bench_fp8_dbg$v.err
  python -c "
import json
d=json.loads([l for l in open('$O/bench_fp8_dbg$v.json') if l.startswith('{')][-1])
print('fp8 GC_GEMM_DBG=$v:', d['value'], 'views/s', d['ms_per_step'])
for k,x in d['roofline']['other'].items():
    if 'conv' in k: print('   ', k, x)
"
done
timeout 900 python scripts/pmc_traffic.py 1000000 16 $O/raster_traffic_views8_1m.json 8 > $O/raster_traffic_1m.log 2>&1; tail -4 $O/raster_traffic_1m.log
timeout 900 python scripts/pmc_traffic.py 4000000 16 $O/raster_traffic_views8_4m.json 8 > $O/raster_traffic_4m.log 2>&1; tail -4 $O/raster_traffic_4m.log
python - <<PY
import json
out={}
for f in ("$O/raster_traffic_views8_1m.json","$O/raster_traffic_views8_4m.json"):
    try:
        d=json.load(open(f))
        for k,v in d.items():
            if k=="_detail": out.setdefault("_detail",{}).update(v)
            else: out[k]=v
    except Exception as e: print("missing", f, e)
json.dump(out, open("profiles/r06_raster_traffic_views8.json","w"))
json.dump(out, open("$O/r06_raster_traffic_views8.json","w"))
PY
for NG in 1000000 4000000; do
  timeout 600 python bench.py --workload raster --gaussians $NG --steps 32 --warmup 2 --no-cpu-baseline > $O/raster_${NG}.json 2> $O/raster_${NG}.err
  python -c "
import json; d=json.loads(open('$O/raster_${NG}.json').read().strip().splitlines()[-1]); c=d['roofline']['chain']; print('$NG', d['value'], c['kernel_us_per_view'], 'frac', c['frac'], 'counters', c['frac_counters'], 'ratio', c['traffic_ratio'])
print({k:(v['avg_us'], v.get('traffic_MB')) for k,v in d['roofline']['stages'].items()})"
done
