set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5zg}
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 900 python bench.py --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r['avg_launch_us'], d['secondary']['value'], d['secondary_fp8']['value'], d['secondary_fp8']['roofline']['avg_launch_us'])"
done
