# Round-4 development traces (one gpurun call): kernel timelines of one proof at 2^16 / 2^20 range and 2^24 AND
# (rocprofv3 --kernel-trace, rocpd database -> tools/trace_gaps.py), and the resident tail's per-round device stamps
# with and without the opening's precommit beside it.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for cfg in "r16:2:--log-n 16 --table range" "r20:2:--log-n 20 --table range" "a24:4:--log-n 24 --table and"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; cols=${rest%%:*}; flags=${rest#*:}
  rm -rf $O/tr_$tag
  rocprofv3 --kernel-trace --output-format rocpd -d $O/tr_$tag -- python3 bench.py $flags --steps 3 --warmup 2 --no-cpu-baseline --no-inflight --no-profile > $O/tr_$tag.log 2>&1
  DB=$(find $O/tr_$tag -name "*.db" | head -1)
  LH_TRACE_COLUMNS=$cols python3 tools/trace_gaps.py "$DB" "sc_tail|sc_round_lds" > $O/r04_gaps_$tag.txt 2>&1
  rm -rf $O/tr_$tag
done
for pc in; do
  LH_OPEN_PRECOMMIT=$pc LH_SC_TAIL_TRACE=1 LH_SC_DEBUG=1 python3 bench.py --log-n 24 --steps 1 --warmup 1 --no-cpu-baseline --no-inflight --no-profile > /dev/null 2> $O/r04_tailtrace_pc$pc.txt
done
