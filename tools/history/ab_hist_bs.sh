# HISTORICAL: the knob this script sweeps (see profiles/README.md for its result) was removed from the library in round 5;
# kept as the record of how the committed numbers were made, it no longer changes anything.
export TMPDIR=/tmp
for bs in 128 256 512; do
  rm -rf gpurun_out/hp
  LH_RS_HIST_BS=$bs rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hp -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inflight --no-extra > gpurun_out/hp.log 2>&1
  F=$(find gpurun_out/hp -name "*kernel_stats.csv" | head -1)
  echo "bs=$bs $(grep -E "rs_hist" $F | cut -c90-)"
done
rm -rf gpurun_out/hp
