# HISTORICAL: the knob this script sweeps (see profiles/README.md for its result) was removed from the library in round 5;
# kept as the record of how the committed numbers were made, it no longer changes anything.
# development: the continuation levels as trees (LH_MSM_TREE_MAX = longest list that goes by trees, 0: never; LH_MSM_TREE_T
# = slots per tile)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "0 64" "65536 64" "65536 256" "262144 64" "16384 64" "0 64" "65536 64"; do
  set -- $cfg
  export LH_MSM_TREE_MAX=$1 LH_MSM_TREE_T=$2
  for w in "--log-n 16 --table range" "--log-n 20 --table range" ""; do
    python3 bench.py $w --steps 12 --warmup 3 --no-cpu-baseline --no-inflight 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['name']:(x['launches'],x['ms']) for x in d['kernels']}
print('tree_max $1 T $2 | %-28s | %.3f ms | levels %s | reduce %s' % ('$w', d['value'], k.get('msm_accumulate_levels'), k.get('msm_bucket_reduce')))"
  done
done
