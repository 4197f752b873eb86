# A/B of the MSM half-batch pipeline (option msm_half_batches) and its knobs on 2^24 AND; usage (GPU box): bash tools/ab_half.sh
cd "$(dirname "$0")/.."
B="--steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-extra"
for rep in a b; do
  for cfg in "LH_MSM_HALF_BATCHES=0" "LH_MSM_HALF_BATCHES=1" "LH_MSM_HALF_BATCHES=1 LH_MSM_HALF_COVER=24" "LH_MSM_HALF_BATCHES=1 LH_MSM_HALF_COVER=40"; do
    v=$(env $cfg python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms']['commit'], d['phases_ms']['open_n'])")
    echo "$cfg : $v"
  done
done
