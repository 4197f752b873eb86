# Round-3 evidence run (one gpurun call): bench lines, rocprofv3 kernel stats, the two PMC passes, exchange-policy sweep.
# usage (on the GPU box, from the repo root): bash tools/r03_evidence.sh
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/r03_bench_and_2p24.json 2> $O/r03_bench_and_2p24.err
python3 bench.py --log-n 20 --table range --steps 20 --warmup 5 > $O/r03_bench_2p20.json 2>> $O/r03_bench.err
python3 bench.py --log-n 16 --table range --steps 20 --warmup 5 > $O/r03_bench_2p16.json 2>> $O/r03_bench.err
python3 bench.py --log-n 24 --table range --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_range_2p24.json 2>> $O/r03_bench.err
python3 bench.py --log-n 24 --table xor --steps 10 --warmup 3 --no-cpu-baseline --no-inflight > $O/r03_bench_xor_2p24.json 2>> $O/r03_bench.err
python3 bench.py --workload hyperplonk --log-n 20 > $O/r03_bench_hyperplonk_2p20.json 2>> $O/r03_bench.err
python3 bench.py --workload hyperplonk --lookup lasso --log-n 20 > $O/r03_bench_hyperplonk_lasso_2p20.json 2>> $O/r03_bench.err
python3 bench.py --workload hyperplonk --lookup lasso --circuit keccak --log-n 20 > $O/r03_bench_hyperplonk_keccak_2p20.json 2>> $O/r03_bench.err
for n in 18 22 26; do python3 bench.py --log-n $n --table range --steps 5 --warmup 2 --no-cpu-baseline --no-inflight > $O/r03_bench_range_2p$n.json 2>> $O/r03_bench.err; done
python3 tools/sharded_rank_profile.py --configs and24,range26 --worlds 1,2,4,8 --out $O/r03_sharded_rank_ms.json > $O/r03_rank.log 2>&1
# rocprofv3 kernel trace + stats of the default command (the program itself after --: no env / shell hop)
rm -rf $O/r03_prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inflight > $O/r03_prof_stats.log 2>&1
find $O/r03_prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_rocprof_kernel_stats.csv
# PMC: separate passes, counters only with --kernel-trace
rm -rf $O/r03_pmcf $O/r03_pmcw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r03_pmcf -- python3 tools/big_run.py and 24 > $O/r03_pmcf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r03_pmcw -- python3 tools/big_run.py and 24 > $O/r03_pmcw.log 2>&1
F=$(find $O/r03_pmcf -name "*counter_collection.csv" | head -1)
W=$(find $O/r03_pmcw -name "*counter_collection.csv" | head -1)
python3 tools/pmc_extract.py "$F" "$W" "tools/big_run.py and 24 (setup + 3 Lasso proofs of 2^24 AND lookups)" 3 $O/r03_pmc_and_2p24.json > $O/r03_pmc_extract.log 2>&1
rm -rf $O/r03_pmcf $O/r03_pmcw $O/r03_prof_stats   # (raw traces are large; the summaries stay)
for f in r03_bench_and_2p24 r03_bench_2p20 r03_bench_2p16 r03_bench_range_2p18 r03_bench_range_2p22 r03_bench_range_2p24 r03_bench_range_2p26 r03_bench_xor_2p24 r03_bench_hyperplonk_2p20 r03_bench_hyperplonk_lasso_2p20 r03_bench_hyperplonk_keccak_2p20; do
  python3 -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['value'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('proof_bytes_equal_gpu'))"
done
head -5 $O/r03_rocprof_kernel_stats.csv
tail -3 $O/r03_pmc_extract.log
tail -n 9 $O/r03_rank.log
# kernel stats of configs[1] (2^20 range) and configs[4] (Keccak-f circuit) as well
for cfg in "2p20:--log-n 20 --table range" "keccak:--workload hyperplonk --lookup lasso --circuit keccak --log-n 20"; do
  tag=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $O/r03_prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof_$tag -- python3 bench.py $flags --steps 5 --warmup 2 --no-cpu-baseline --no-inflight > $O/r03_prof_$tag.log 2>&1
  find $O/r03_prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_rocprof_kernel_stats_$tag.csv
  rm -rf $O/r03_prof_$tag
done
