cd "${GRAFT_REPO_ROOT:-.}"
B="--steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-extra"
for rep in a b; do
for r in 0 10 14 17 20; do
  export LH_GKR_HOOK_LAYER=$r
  echo -n "layer=$r and24 "; python bench.py $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print(d['value'], 'gkr', p['gkr'], 'evals', p['evals'], 'open', p['open_n'])"
  echo -n "layer=$r range20 "; python bench.py $B --log-n 20 --table range | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print(d['value'], 'gkr', p['gkr'], 'open', p['open_n'])"
done
done
