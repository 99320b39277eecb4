# default bench line + the same command under rocprofv3 --kernel-trace --stats, final build (one box, one call)
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5zf}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err
for DT in bf16 fp8; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$DT -o bench -- python $R/bench.py --dtype $DT --no-cpu-baseline --no-secondary > $R/$O/bench_${DT}_profiled.json 2> $R/$O/bench_${DT}_profiled.err)
  DB=$(find $O/prof_$DT -name "*.db" | head -1)
  python scripts/rocpd_stats.py $DB 70 > $O/bench_kernel_stats_$DT.txt
  rm -rf $O/prof_$DT
  head -3 $O/bench_kernel_stats_$DT.txt | cut -c1-150
done
python -c "
import json
d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['mfma_util_step'], r['frac'], r['avg_launch_us'], 'f16', d['secondary']['value'], 'fp8', d['secondary_fp8']['value'])
for k in ('bf16','fp8'):
    p=json.loads(open('$O/bench_'+k+'_profiled.json').read().strip().splitlines()[-1]); print(k, 'profiled', p['value'], p['roofline']['avg_launch_us'])
"
