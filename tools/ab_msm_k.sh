# development: entries per accumulate thread (LH_MSM_K; 0 = by batch size) now that the continuation levels are trees
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for k in 0 8 16 32 0 8; do
  for w in "--log-n 16 --table range" "--log-n 20 --table range"; do
    LH_MSM_K=$k python3 bench.py $w --steps 12 --warmup 3 --no-cpu-baseline --no-inflight 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['name']:(x['launches'],x['ms']) for x in d['kernels']}
print('K $k | %-26s | %.3f ms | acc0 %s | levels %s' % ('$w', d['value'], k.get('msm_accumulate0'), k.get('msm_accumulate_levels')))"
  done
done
for k in 0 32 64 128; do
  LH_MSM_K=$k python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-inflight 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['name']:(x['launches'],x['ms']) for x in d['kernels']}
print('K $k | and 2^24 | %.3f ms | acc0 %s | levels %s' % (d['value'], k.get('msm_accumulate0'), k.get('msm_accumulate_levels')))"
done
