# A/B: the helper ctx's accumulate0 grid capped (LH_HELPER_ACC_GRID workgroups of 128 threads; 0 = the whole chip)
cd "${GRAFT_REPO_ROOT:-.}"
for g in 0 256 512 768 1024 1536; do
  for rep in 1 2; do
  LH_HELPER_ACC_GRID=$g python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid=$g', d['value'], {k:d['phases_ms'][k] for k in ('gkr','open_n')})"
  done
done
for g in 0 512; do
  LH_HELPER_ACC_GRID=$g python bench.py --log-n 24 --table range --steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('range24 grid=$g', d['value'], {k:d['phases_ms'][k] for k in ('gkr','open_n')})"
  LH_HELPER_ACC_GRID=$g python bench.py --log-n 20 --table range --steps 20 --warmup 5 --no-cpu-baseline --no-inflight --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('range20 grid=$g', d['value'], {k:d['phases_ms'][k] for k in ('gkr','open_n')})"
done
