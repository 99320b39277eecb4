# Round 5 evidence pass (build with the GEMM scheduling work): suite, smoke, default bench, rocprofv3 kernel summaries bf16 + fp8, PMC traffic, raster-only lines
# (1) whole -m gpu suite with margins; (2) default bench line (bf16 + f16 / fp8 secondaries + cpu_baseline); (3) rocprofv3 kernel summaries of the
# bf16 and fp8 runs; (4) PMC traffic of k_attn5 and of the batched rasterizer (separate FETCH_SIZE / WRITE_SIZE passes); (5) raster-only lines.
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5v}
mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -1 > $O/box.txt; hostname >> $O/box.txt
GC_TEST_MARGINS=$R/$O/margins.jsonl timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/tests.log
tail -2 $O/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
tail -1 $O/bench_bf16.json | cut -c1-400
for DT in bf16 fp8; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$DT -o bench -- python $R/bench.py --dtype $DT --no-cpu-baseline --no-secondary > $R/$O/bench_${DT}_profiled.json 2> $R/$O/bench_${DT}_profiled.err)
  DB=$(find $O/prof_$DT -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats_$DT.txt
  rm -rf $O/prof_$DT
  head -14 $O/bench_kernel_stats_$DT.txt
done
PYTHONPATH=$R timeout 600 python scripts/pmc_kernel_traffic.py 'k_attn5' $O/attn_traffic.json -- python $R/scripts/attn5_check.py time 0 > $O/attn_traffic.log 2>&1; tail -12 $O/attn_traffic.log
timeout 900 python scripts/pmc_traffic.py 1000000 16 $O/raster_traffic_views8_1m.json 8 > $O/raster_traffic_1m.log 2>&1; tail -5 $O/raster_traffic_1m.log
timeout 900 python scripts/pmc_traffic.py 4000000 16 $O/raster_traffic_views8_4m.json 8 > $O/raster_traffic_4m.log 2>&1; tail -5 $O/raster_traffic_4m.log
for NG in 1000000 4000000; do
  timeout 600 python bench.py --workload raster --gaussians $NG --steps 24 --warmup 2 > $O/raster_${NG}.json 2> $O/raster_${NG}.err
  python -c "
import json; d=json.loads(open('$O/raster_${NG}.json').read().strip().splitlines()[-1]); c=d['roofline']['chain']; print('$NG', d['value'], c['kernel_us_per_view'], c['frac'], c['frac_processed_pairs'], c['traffic_ratio'])"
done
