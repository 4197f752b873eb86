# Evidence run of a round (one gpurun call): bench lines, rocprofv3 kernel stats, the two PMC passes, the loopback rank profile.
# usage (on the GPU box, from the repo root): [R=r06] bash tools/evidence.sh
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
R=${R:-r06}
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_and_2p24.json 2> $O/${R}_bench_and_2p24.err
python3 bench.py --log-n 20 --table range --steps 20 --warmup 5 > $O/${R}_bench_2p20.json 2>> $O/${R}_bench.err
python3 bench.py --log-n 16 --table range --steps 20 --warmup 5 > $O/${R}_bench_2p16.json 2>> $O/${R}_bench.err
python3 bench.py --log-n 24 --table range --steps 10 --warmup 3 --no-cpu-baseline > $O/${R}_bench_range_2p24.json 2>> $O/${R}_bench.err
python3 bench.py --log-n 24 --table xor --steps 10 --warmup 3 --no-cpu-baseline --no-inflight > $O/${R}_bench_xor_2p24.json 2>> $O/${R}_bench.err
python3 bench.py --workload hyperplonk --log-n 20 > $O/${R}_bench_hyperplonk_2p20.json 2>> $O/${R}_bench.err
python3 bench.py --workload hyperplonk --lookup lasso --log-n 20 > $O/${R}_bench_hyperplonk_lasso_2p20.json 2>> $O/${R}_bench.err
python3 bench.py --workload hyperplonk --lookup lasso --circuit keccak --log-n 20 > $O/${R}_bench_hyperplonk_keccak_2p20.json 2>> $O/${R}_bench.err
for n in 17 18 19 22 26; do python3 bench.py --log-n $n --table range --steps 5 --warmup 2 --no-cpu-baseline --no-inflight > $O/${R}_bench_range_2p$n.json 2>> $O/${R}_bench.err; done
python3 tools/sharded_rank_profile.py --configs and24,range26,keccak20 --worlds 1,2,4,8 --in-flight 2 --out $O/${R}_sharded_rank_ms.json > $O/${R}_rank.log 2>&1
# rocprofv3 kernel trace + stats of the default command (the program itself after --: no env / shell hop)
rm -rf $O/${R}_prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inflight > $O/${R}_prof_stats.log 2>&1
find $O/${R}_prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_rocprof_kernel_stats.csv
# PMC: separate passes, counters only with --kernel-trace
rm -rf $O/${R}_pmcf $O/${R}_pmcw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${R}_pmcf -- python3 tools/big_run.py and 24 > $O/${R}_pmcf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${R}_pmcw -- python3 tools/big_run.py and 24 > $O/${R}_pmcw.log 2>&1
F=$(find $O/${R}_pmcf -name "*counter_collection.csv" | head -1)
W=$(find $O/${R}_pmcw -name "*counter_collection.csv" | head -1)
python3 tools/pmc_extract.py "$F" "$W" "tools/big_run.py and 24 (setup + 3 Lasso proofs of 2^24 AND lookups)" 3 $O/${R}_pmc_and_2p24.json > $O/${R}_pmc_extract.log 2>&1
rm -rf $O/${R}_pmcf $O/${R}_pmcw $O/${R}_prof_stats   # (raw traces are large; the summaries stay)
for f in ${R}_bench_and_2p24 ${R}_bench_2p20 ${R}_bench_2p16 ${R}_bench_range_2p18 ${R}_bench_range_2p22 ${R}_bench_range_2p24 ${R}_bench_range_2p26 ${R}_bench_xor_2p24 ${R}_bench_hyperplonk_2p20 ${R}_bench_hyperplonk_lasso_2p20 ${R}_bench_hyperplonk_keccak_2p20; do
  python3 -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['value'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('proof_bytes_equal_gpu'))"
done
head -5 $O/${R}_rocprof_kernel_stats.csv
tail -3 $O/${R}_pmc_extract.log
tail -n 9 $O/${R}_rank.log
# kernel stats of configs[1] (2^20 range) and configs[4] (Keccak-f circuit) as well
for cfg in "2p20:--log-n 20 --table range" "keccak:--workload hyperplonk --lookup lasso --circuit keccak --log-n 20"; do
  tag=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $O/${R}_prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_$tag -- python3 bench.py $flags --steps 5 --warmup 2 --no-cpu-baseline --no-inflight > $O/${R}_prof_$tag.log 2>&1
  find $O/${R}_prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_rocprof_kernel_stats_$tag.csv
  rm -rf $O/${R}_prof_$tag
done

# the reference-defined micro-workloads (tests/micro_bench.py: zero-check 2^20-2^23, mKZG commit / open, GKR x3, streams)
python3 tests/micro_bench.py 20 > $O/${R}_microbench.json 2> $O/${R}_microbench.err
# the driver's N = 8 command as a plumbing run on this one GPU
TAG=$R bash tools/dryrun_gpus8.sh 20 5 > $O/${R}_dryrun.log 2>&1
# instruction audit of the streaming rounds and the bucket accumulation (static, no GPU needed)
python3 tools/isa_audit.py kernels_sumcheck.hip 'sc_round_pp_kernel|sc_round_open_kernel<4|sc_round_rw_kernel<4' > $O/${R}_isa_audit.txt 2>&1
python3 tools/isa_audit.py msm.hip 'msm_accumulate0_kernel' >> $O/${R}_isa_audit.txt 2>&1
if [ "${FULLSIZE:-0}" = "1" ]; then
  # configs[3] at FULL size against the C++ oracle's bytes (2^26 range-check lookups, ~4 min of host time)
  python3 bench.py --log-n 26 --table range --steps 3 --warmup 1 --cpu-sample-log-n 26 --no-inflight --no-profile > $O/${R}_bench_range_2p26_fullsize_cpu.json 2>> $O/${R}_bench.err
  python3 -c "import json; d=json.load(open('$O/${R}_bench_range_2p26_fullsize_cpu.json')); print('2^26 full size:', d['value'], d['cpu_baseline'])"
fi
