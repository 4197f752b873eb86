# Round-5 sweep of the MSM tail's knobs at the small sizes (one gpurun call): ms per proof and the MSM kernels' sums.
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # $1 = label, rest = env assignments
  label=$1; shift
  for cfg in "--log-n 16 --table range" "--log-n 20 --table range"; do
    env "$@" python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-inflight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks={k['name']:round(k['ms'],3) for k in d['kernels']}
print('%-28s %-12s %.3f ms  reduce %s levels %s acc0 %s' % ('$label', '$cfg'.split()[1], d['value'], ks.get('msm_bucket_reduce'), ks.get('msm_accumulate_levels'), ks.get('msm_accumulate0')))"
  done
}
run default LH_NOP=1
for s in 4 8 16; do run "SEG=$s" LH_MSM_SEG=$s; done
for k in 2 8 16; do run "K=$k" LH_MSM_K=$k; done
for t in 65536 1048576 4194304; do run "TREE_MAX=$t" LH_MSM_TREE_MAX=$t; done
# (the tile-size line of the committed sweep, LH_MSM_TREE_T=256, went with the knob: 64-slot tiles won)
for k2 in 2 8; do run "K2=$k2" LH_MSM_K2=$k2; done
for c in 3 5 6; do run "C_OFF=$c" LH_MSM_C_OFF=$c; done
