# Marginal value of each op class in the concurrent schedule: bench.py with the class's launches skipped (KernelOptions.ablate; results
# wrong by construction).  views/s(ablated) vs views/s(full) = what a zero-cost version of that class would buy.  -> gpurun_out/$1/ablate.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-abl}; mkdir -p $O
export TMPDIR=/tmp
: > $O/ablate.txt
for a in "" gn ln gn,ln attn40 attn80 attn160 linear conv tail head conv_hw64 conv_hw32 conv_hw16,conv_hw8; do
  GC_ABLATE=$a timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/abl_tmp.json 2> $O/abl_tmp.err
  python - "$a" $O/abl_tmp.json >> $O/ablate.txt <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    print(f"ablate={sys.argv[1] or '-':24s} views/s {d['value']:.3f}  ms_per_step {d['ms_per_step']:.1f}")
except Exception as e:
    print(f"ablate={sys.argv[1]:24s} FAILED {e}")
P
  tail -1 $O/ablate.txt
done
