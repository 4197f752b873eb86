# 8-rank loopback wall (rank 0) against the exchange policy and the round-combination variant: ms per proof, collectives
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in and24 range26; do
for x in 17 18 19 20; do for cr in 0 1; do
  LH_SHARD_EXCHANGE_LOG=$x LH_COMM_ROUND=$cr python tools/sharded_rank_profile.py --configs $cfg --worlds 8 --steps 5 --out /tmp/x.json 2>/dev/null > /dev/null
  python - <<PY
import json
d=json.load(open('/tmp/x.json'))['configs']['$cfg']['worlds']['8']
r=d['ranks'][0]
print('$cfg xlog=$x comm_round=$cr wall %.2f busy %.2f collectives %s sharded_rounds %d' % (d['max_rank_wall_ms'], d['max_rank_busy_ms'], r['collectives_per_proof'], r['route']['sharded_rounds']))
PY
done; done; done
