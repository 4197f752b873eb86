cd "${GRAFT_REPO_ROOT:-.}"
for g in 0 1024 1536 2048 3072 4608 6144; do
  for rep in 1 2; do
  LH_ACC_GRID=$g python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inflight 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid=$g', d['value'], [k['ms'] for k in d['kernels'] if k['name']=='msm_accumulate0'], {k:d['phases_ms'][k] for k in ('commit','gkr','open_n')})"
  done
done
