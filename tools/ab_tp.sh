# development: up to how many (pair, term) items a factored round runs one thread per (pair, term) instead of the
# entry-per-lane kernels (LH_SC_TP_MAX_ITEMS, default 2^17)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for t in 131072 262144 524288 1048576 131072 524288; do
  for w in "" "--log-n 20 --table range"; do
    LH_SC_TP_MAX_ITEMS=$t python3 bench.py $w --steps 10 --warmup 3 --no-cpu-baseline --no-inflight 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['name']:(x['launches'],x['ms']) for x in d['kernels']}
print('tp_max $t | %-26s | %.3f ms | gkr %.2f | %s' % ('$w', d['value'], d['phases_ms']['gkr'], {n:v for n,v in k.items() if 'sc_round' in n}))"
  done
done
