#!/bin/bash
# A/B of Options::msm_window_tables (window tables of the SRS levels: one bucket set per full-width MSM job) on the GPU box.
# usage: tools/ab_window_tables.sh [levels...]   (default 0 20 22; writes gpurun_out/r03_wt_*.json)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-extra"
LEVELS="${@:-0 20 22}"
for rep in a b; do
for wt in $LEVELS; do
  LH_MSM_WINDOW_TABLES=$wt python bench.py $B > gpurun_out/r03_wt_and24_${wt}_$rep.json 2>> gpurun_out/r03_wt.err
  LH_MSM_WINDOW_TABLES=$wt python bench.py $B --log-n 20 --table range > gpurun_out/r03_wt_range20_${wt}_$rep.json 2>> gpurun_out/r03_wt.err
  LH_MSM_WINDOW_TABLES=$wt python bench.py $B --workload hyperplonk --lookup lasso --circuit keccak --steps 5 --warmup 2 > gpurun_out/r03_wt_keccak_${wt}_$rep.json 2>> gpurun_out/r03_wt.err
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_wt_*_[ab].json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line", e); continue
    k = {x["name"]: x["ms"] for x in d.get("kernels", [])}
    print(f, d["value"], {n: k.get(n) for n in ("msm_accumulate0", "msm_bucket_reduce", "msm_sort", "msm_accumulate_levels")})
PY
