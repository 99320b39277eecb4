#!/bin/bash
# round 6, call J: PMC traffic of the sorted-boxes raster chain (1 M / 4 M, 8 views per launch set; calibration copies in the same passes), raster-only bench lines that read it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j
mkdir -p $O
timeout 900 python scripts/pmc_traffic.py 1000000 16 $O/raster_traffic_views8_1m.json 8 > $O/raster_traffic_1m.log 2>&1
timeout 900 python scripts/pmc_traffic.py 4000000 16 $O/raster_traffic_views8_4m.json 8 > $O/raster_traffic_4m.log 2>&1
python - <<PY
import json
out={}
for f in ("$O/raster_traffic_views8_1m.json","$O/raster_traffic_views8_4m.json"):
    d=json.load(open(f))
    n=[k for k in d if k.isdigit()][0]
    out[n]=d[n]
    out.setdefault("_detail",{})[n]={k:v for k,v in d.items() if k!=n and k!="_detail"}
    print(n, d[n], d.get("_correction"))
json.dump(out, open("profiles/r06_raster_traffic_views8.json","w"))
json.dump(out, open("$O/r06_raster_traffic_views8.json","w"))
PY
for NG in 1000000 4000000; do
  timeout 600 python bench.py --workload raster --gaussians $NG --steps 32 --warmup 2 --no-cpu-baseline > $O/raster_${NG}.json 2> $O/raster_${NG}.err
  python -c "
import json; d=json.loads(open('$O/raster_${NG}.json').read().strip().splitlines()[-1]); c=d['roofline']['chain']; print('$NG', d['value'], c['kernel_us_per_view'], 'frac', c['frac'], 'counters', c['frac_counters'], 'ratio', c['traffic_ratio'])
print({k:(v['avg_us'], v.get('traffic_MB'), v.get('traffic_GBps')) for k,v in d['roofline']['stages'].items()})"
done
