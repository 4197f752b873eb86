# development: terms per shared reduction in sc_round_pp_kernel (4: 160 registers, 3 waves; 2: 127 registers, 4 waves)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics"
for g in 2 4 2 4; do
  ( cd halo2-lasso_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DLH_PP_GROUP=$g -c kernels_sumcheck.hip -o kernels_sumcheck.o 2>/dev/null && make 2>/dev/null | tail -1 > /dev/null )
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inflight > $O/ab_ppg_$g.json 2> $O/ab_ppg.err
  python3 -c "
import json; d=json.load(open('$O/ab_ppg_$g.json'))
k={x['name']:x['ms'] for x in d['kernels']}
print('pp group $g: total %.2f ms, sc_round_pp<bind> %.3f, <first> %.3f, gkr phase %.2f' % (d['value'], k.get('sc_round_pp<bind>',0), k.get('sc_round_pp<first>',0), d['phases_ms']['gkr']))"
done
# (the loop ends on the default build: the tree's library must not stay a variant)
