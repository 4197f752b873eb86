# bench lines at the small sizes (2^16, 2^20 range; 2^24 AND) - value, phases, top kernels
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "--log-n 16 --table range" "--log-n 20 --table range" "${1:---log-n 24 --table and --steps 10 --warmup 3}"; do
  python bench.py $cfg --no-cpu-baseline --no-inflight 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:30], d['value'], d['phases_ms'], [(k['name'],k['launches'],round(k['ms'],2)) for k in d['kernels']])"
done
