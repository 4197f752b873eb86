"""CPU, world_size 2 over gloo: the multi-GPU plumbing of bench.py (halo2-lasso_amd/dist.py).
Every rank owns an independent batch (no data-path collective); the job's time is the max over ranks."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    from halo2_lasso_amd import dist as hdist
    rank, local_rank, world = hdist.env_rank()
    d = hdist.init("gloo")
    assert d is not None and d.get_world_size() == 2 and d.get_rank() == rank
    # each rank's batch: deterministic, different across ranks
    seed = hdist.batch_seed(10, rank)
    dims = np.random.Generator(np.random.PCG64(seed)).integers(0, 1 << 16, size=1 << 10, dtype=np.uint32)
    hdist.barrier(d)
    elapsed = hdist.max_over_ranks(d, 1.0 + rank)      # rank 1 is the slow one
    m = hdist.job_metrics(elapsed, steps=4, world=world, lookups_per_proof=1 << 10)
    print(json.dumps({"rank": rank, "seed": seed, "sum": int(dims.sum()), "elapsed": elapsed, **m}), flush=True)
    hdist.barrier(d)
    d.destroy_process_group()
""") % ROOT


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    a, b = sorted(outs, key=lambda x: x["rank"])
    assert a["seed"] != b["seed"] and a["sum"] != b["sum"]          # disjoint batches
    assert a["elapsed"] == b["elapsed"] == 2.0                        # max over ranks
    assert a["ms_per_step"] == 500.0 and a["value_ms_per_proof"] == 250.0
    assert a["lookups_per_s"] == (1 << 10) * 2 / 0.5


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from halo2_lasso_amd import dist as hdist
    os.environ.pop("WORLD_SIZE", None)
    assert hdist.init() is None
    assert hdist.max_over_ranks(None, 1.5) == 1.5
    assert hdist.job_metrics(2.0, 4, 1, 1 << 20)["value_ms_per_proof"] == 500.0


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (no WORLD_SIZE): bench.py itself starts the two ranks before any GPU
    call, rank 0's line comes back on stdout, the exit code is the ranks'.  (--rendezvous-only: process group over gloo
    and one all-reduce, no GPU work - the launch path on this CPU-only machine.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LH_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "max_rank": 1, "rendezvous": "ok"}
    # a rank that fails takes the job's exit code with it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=dict(env, LH_DIST_BACKEND="no-such-backend"), capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
