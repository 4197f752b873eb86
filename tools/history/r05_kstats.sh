# rocprofv3 kernel stats of `tools/big_run.py <kind> <n>` (setup + 3 proofs) under the env assignments given: the rows matching $PAT
# usage: bash tools/r05_kstats.sh "<env assignments>" <kind> <n> <grep pattern> <tag>
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
ENVS=$1; KIND=${2:-and}; N=${3:-24}; PAT=${4:-msm}; TAG=${5:-x}
rm -rf $O/ks_$TAG
export $ENVS
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$TAG -- python3 tools/big_run.py $KIND $N > $O/ks_$TAG.log 2>&1
F=$(find $O/ks_$TAG -name "*kernel_stats.csv" | head -1)
echo "== $ENVS ($KIND $N)"; tail -3 $O/ks_$TAG.log | head -1
python3 - "$F" "$PAT" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
pat=re.compile(sys.argv[2])
for r in rows:
    if pat.search(r['Name']):
        print('  %-70s calls %5s total %9.3f ms avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
rm -rf $O/ks_$TAG
