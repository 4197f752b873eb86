cd "${GRAFT_REPO_ROOT:-.}"
B="--steps 10 --warmup 3 --no-cpu-baseline --no-inflight --no-extra"
for n in 17 18 19; do
for v in 0 1 0 1; do
  echo -n "range n=$n precommit=$v "; LH_OPEN_PRECOMMIT=$v python bench.py $B --log-n $n --table range | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
done
done
bash tools/ab_tmp.sh
