#!/bin/bash
# round 6, call L: counters of a launch set cleared by ONE kernel instead of 27 memsets -- raster tests, raster-only lines at 1 M / 4 M (compare r6j, same command), default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l
mkdir -p $O
timeout 900 python -m pytest tests/test_raster_views_gpu.py tests/test_raster_gpu.py -m gpu -q -x > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for NG in 1000000 4000000; do
  timeout 600 python bench.py --workload raster --gaussians $NG --steps 32 --warmup 2 --no-cpu-baseline > $O/raster_${NG}.json 2> $O/raster_${NG}.err
  python -c "
import json; d=json.loads(open('$O/raster_${NG}.json').read().strip().splitlines()[-1]); c=d['roofline']['chain']; print('$NG', d['value'], c['kernel_us_per_view'], 'frac', c['frac'], 'counters', c['frac_counters'], 'ratio', c['traffic_ratio'])
print({k:v['avg_us'] for k,v in d['roofline']['stages'].items()})"
done
timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 28 --warmup 14 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['mfma_util_step'], d['roofline_raster']['kernel_us_per_view'], d['roofline_raster']['frac'])"
