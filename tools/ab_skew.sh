# A/B: arena skew in front of large allocations (channel staggering of the tables a round kernel streams together)
for sk in 0 4352 33024 266240 0; do
  LH_ARENA_SKEW=$sk python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-inflight 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); ks={k['name']:k for k in d['kernels']}
print('skew $sk', d['value'], ' '.join('%s %.2f (%.0f GB/s)' % (n, ks[n]['ms'], ks[n]['GBps_largest']) for n in ('sc_round_pp<bind>','sc_round_open<bind>','sc_round_rw<bind>','sc_round_pp<first>','lincomb','lasso_rw_leaves','msm_sort') if n in ks))"
done
