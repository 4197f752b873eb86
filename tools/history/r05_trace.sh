# Round-5 development traces (one gpurun call): kernel timelines (rocprofv3 --kernel-trace, rocpd database ->
# tools/trace_gaps.py) of ONE proof of rank 0 of an 8-rank world over the loopback communicator, 2^24 AND and 2^26 range.
# usage: bash tools/r05_trace.sh [tag] [configs] [kernel regex for the per-launch listing]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-base}
CFGS=${2:-"and24 range26"}
PAT=${3:-"sum_publish|gather_interleave"}
mkdir -p $O
for cfg in $CFGS; do
  rm -rf $O/tr_$cfg
  rocprofv3 --kernel-trace --output-format rocpd -d $O/tr_$cfg -- python3 tools/sharded_trace.py --config $cfg --world 8 > $O/r05_shtrace_${TAG}_$cfg.log 2>&1
  DB=$(find $O/tr_$cfg -name "*.db" | head -1)
  LH_TRACE_SPLIT_IDLE_MS=50 python3 tools/trace_gaps.py "$DB" "$PAT" > $O/r05_gaps_${TAG}_w8_$cfg.txt 2>&1
  rm -rf $O/tr_$cfg
  tail -2 $O/r05_shtrace_${TAG}_$cfg.log
  head -40 $O/r05_gaps_${TAG}_w8_$cfg.txt
done
