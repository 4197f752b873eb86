# development: VALU / memory-unit utilisation of the streaming kernels (rocprofv3 derived counters, one pass each)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
for ctr in VALUBusy MemUnitBusy MemUnitStalled SQ_WAVES_sum; do
  rm -rf $O/pmcb_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmcb_$ctr -- python3 tools/big_run.py and 24 > $O/pmcb_$ctr.log 2>&1
  F=$(find $O/pmcb_$ctr -name "*counter_collection.csv" | head -1)
  python3 - "$F" $ctr <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
try:
    rows = list(csv.DictReader(open(f)))
except Exception as e:
    print(ctr, "no data", e); sys.exit(0)
for r in rows:
    name = r.get("Kernel_Name", "")
    short = name.split("(")[0].replace("void ", "").replace("lh::", "")
    v = float(r.get("Counter_Value", 0) or 0)
    a = agg[short]; a[0] += 1; a[1] += v; a[2] = max(a[2], v)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:400]:
    if any(s in k for s in ("sc_round", "msm_accumulate0", "lincomb", "fix_var", "tree_up", "rs_scatter", "lasso_rw")):
        print("%-14s %-48s launches %4d  mean %10.2f  max %10.2f" % (ctr, k[:48], a[0], a[1] / a[0], a[2]))
PY
  rm -rf $O/pmcb_$ctr
done
