for f in "--steps 12" "--log-n 24 --table range --steps 12"; do
for rep in a b c; do
  for m in 0 1; do
    v=$(LH_OPEN_FOLD_COLS=$m timeout 300 python bench.py $f --warmup 3 --no-cpu-baseline --no-inflight --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms']['open_n'])")
    echo "$f fold_cols=$m : $v"
  done
done
done
