cd $GRAFT_REPO_ROOT
cp halo2-lasso_amd/liblasso_hip.so /tmp/base.so
for v in base w444 w645 w856 w534 base; do
  if [ $v = base ]; then cp /tmp/base.so halo2-lasso_amd/liblasso_hip.so; else cp tools/ubench/liblasso_$v.bin halo2-lasso_amd/liblasso_hip.so; fi
  python3 bench.py --no-cpu-baseline --no-inflight --steps 10 --warmup 3 > gpurun_out/r03_w_$v.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r03_w_$v.json"))
ks={k["name"]:k["ms"] for k in d["kernels"]}
print("$v", d["value"], {n:ks.get(n) for n in ["sc_round<2,bind>","sc_round<2,first>","sc_round_open<bind>","sc_round_rw<bind>"]})
PY
done
cp /tmp/base.so halo2-lasso_amd/liblasso_hip.so
