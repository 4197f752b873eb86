# HISTORICAL: the knob this script sweeps (see profiles/README.md for its result) was removed from the library in round 5;
# kept as the record of how the committed numbers were made, it no longer changes anything.
for cfg in "pc0:LH_OPEN_PRECOMMIT=0" "pc1:LH_OPEN_PRECOMMIT=1" "hook15:LH_GKR_HOOK_AT=15" "hook17:LH_GKR_HOOK_AT=17" "hook19:LH_GKR_HOOK_AT=19"; do
  tag=${cfg%%:*}; kv=${cfg#*:}
  for rep in 1 2; do
    env $kv python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-profile 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('$tag', d['value'], 'gkr', p['gkr'], 'open', p['open_n'], 'commit', p['commit'])"
  done
done
