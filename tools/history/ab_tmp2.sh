cd "${GRAFT_REPO_ROOT:-.}"
timeout 500 python -m pytest tests/test_gpu_parity_large.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r03_pc3_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03_pc3_tests.log | tail -2
B="--steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-extra"
for n in 17 18 19 20; do
for v in 0 1 0 1; do
  echo -n "range n=$n precommit=$v "; LH_OPEN_PRECOMMIT=$v python bench.py $B --log-n $n --table range | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
done
done
