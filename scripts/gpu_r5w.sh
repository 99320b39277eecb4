# tile prefetch in the online-softmax attention kernel (head size 160 / short key streams): tests, microbench, bench A/B vs previous build
set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=gpurun_out/${1:-r5w}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise_kernels_gpu.py tests/test_plugin_gpu.py -m gpu -q -x -k "attention" 2>&1 | grep -v "^  x0" | tail -4 > $O/tests_a.log; tail -3 $O/tests_a.log
for L in prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python scripts/bench_kernels.py attn > $O/ubench_attn_$L.txt 2>&1
done
unset GC_HIP_LIB
paste -d'|' $O/ubench_attn_prev.txt $O/ubench_attn_new.txt | awk -F'|' '{split($1,a,":"); split($2,b,":"); print a[1] ":" substr(a[2],1,24) " |" substr(b[2],1,24)}'
for L in prev new prev new; do
  if [ $L = prev ]; then export GC_HIP_LIB=$R/gaussctrl_amd/libgaussctrl_hip_prev.so; else unset GC_HIP_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_$L.json 2> $O/bench_$L.err
  python -c "
import json; d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:(v['launches'], v['ms']) for k,v in d['roofline']['other'].items() if 'k_attn' in k})"
done
unset GC_HIP_LIB
timeout 1500 python -m pytest tests/test_fullgeom_gpu.py tests/test_denoise_model_gpu.py -m gpu -q -x -k "batch_invariant or edit_f7_h64_all or edit_chunk" 2>&1 | grep -v "^  x0" | tail -4 > $O/tests_m.log; tail -3 $O/tests_m.log
