# The driver's exact N = 8 command, dry-run on ONE GPU (eight ranks share it over gloo: a PLUMBING run,
# not a timing - the ranks take turns on the one GPU and every collective goes through the host);
# watchdogs scaled for the shared GPU (arguments: steps warmup; default 1 1, the driver runs 20 5).
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O
export LH_DEVICE=0 LH_DIST_BACKEND=gloo LH_BENCH_SHARDED_TIMEOUT=${LH_BENCH_SHARDED_TIMEOUT:-1800} LH_BENCH_EXTRAS_TIMEOUT=${LH_BENCH_EXTRAS_TIMEOUT:-2400}
T0=$(date +%s); python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29655 \
  bench.py --gpus 8 --steps ${1:-1} --warmup ${2:-1} > $O/${TAG:-r06}_bench_gpus8_dryrun.json 2> $O/${TAG:-r06}_bench_gpus8_dryrun.err
echo "rc=$? wall=$(( $(date +%s) - T0 )) s"
tail -c 3000 $O/${TAG:-r06}_bench_gpus8_dryrun.json
tail -5 $O/${TAG:-r06}_bench_gpus8_dryrun.err
