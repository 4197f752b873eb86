#!/bin/bash
# A/B of one environment switch on the three bench configurations (2^24 AND, 2^20 range, the Keccak circuit), three
# repetitions each.  usage (GPU box): tools/ab_env.sh NAME [values...]   e.g. tools/ab_env.sh LH_OPEN_PRECOMMIT 0 1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
NAME=$1; shift
VALS="${@:-0 1}"
B="--steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-extra"
for rep in a b c; do
for v in $VALS; do
  env $NAME=$v python bench.py $B > gpurun_out/ab_${NAME}_and24_${v}_$rep.json 2>> gpurun_out/ab_$NAME.err
  env $NAME=$v python bench.py $B --log-n 20 --table range > gpurun_out/ab_${NAME}_range20_${v}_$rep.json 2>> gpurun_out/ab_$NAME.err
  env $NAME=$v python bench.py $B --workload hyperplonk --lookup lasso --circuit keccak --steps 5 --warmup 2 > gpurun_out/ab_${NAME}_keccak_${v}_$rep.json 2>> gpurun_out/ab_$NAME.err
done
done
python - "$NAME" <<'PY'
import json, glob, sys, collections
name = sys.argv[1]
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/ab_%s_*_[abc].json" % name)):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line", e); continue
    cfg, val = f.split("ab_%s_" % name)[1].rsplit("_", 2)[0:2]
    acc[(cfg, val)].append(d["value"])
for k in sorted(acc):
    print(k, " ".join("%.3f" % v for v in acc[k]), " mean %.3f" % (sum(acc[k]) / len(acc[k])))
PY
