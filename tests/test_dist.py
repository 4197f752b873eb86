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


def test_sharded_fallback_chain_under_torchrun():
    """a sharded run that raises on every rank is re-run in fresh processes - first sharded again with the
    personalised exchange staged through all-gathers, then as replicas.  Launched by torch.distributed.run (as the driver
    does) every worker inherits TORCHELASTIC_USE_AGENT_STORE=True, under which even rank 0 is only a CLIENT of the
    agent's store: the successors must drop it or nobody listens on their new port.  (--rendezvous-only over gloo: the
    whole chain without a GPU; LH_BENCH_TEST_RAISE names the attempts that fail.)"""
    import json
    import socket

    def run(raise_at):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
        env.update(LH_DIST_BACKEND="gloo", LH_BENCH_TEST_RAISE=raise_at)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return r, [json.loads(ln) for ln in lines]

    r, lines = run("first")
    assert r.returncode == 0, r.stderr[-2000:]
    assert lines == [{"n_gpus": 2, "max_rank": 1, "rendezvous": "ok", "stage": "a2a", "mode": "sharded", "a2a": "allgather"}]
    r, lines = run("first,a2a")
    assert r.returncode == 0, r.stderr[-2000:]
    assert lines == [{"n_gpus": 2, "max_rank": 1, "rendezvous": "ok", "stage": "replicas", "mode": "replicas", "a2a": "allgather"}]
    r, lines = run("first,a2a,replicas")  # nothing left to fall back to: the job fails, no line
    assert r.returncode != 0 and lines == []


COUNTERS_WORKER = textwrap.dedent("""
    # A numpy restatement of the sharded access counters (csrc/lasso.cpp lasso_counters_sharded: partition by address
    # owner, personalised exchange staged through an all-gather as the callback transports do, rank inside the address
    # run on the owner, ranks back along the same segments, final counts all-gathered) over gloo, 2 ranks, CPU only.
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    from halo2_lasso_amd import dist as hdist
    rank, _, world = hdist.env_rank()
    d = hdist.init("gloo")
    n, l, j = 12, 6, 3
    rho = world.bit_length() - 1
    N, M, NL = 1 << n, 1 << l, (1 << n) >> rho
    rng = np.random.default_rng(12)
    col = rng.integers(0, M, size=N, dtype=np.int64)
    col[rng.random(N) < 0.3] = 5                                   # a hot address
    # reference: the sequential definition (oracle/pyref/lasso.py witness)
    cnt = np.zeros(M, dtype=np.int64); rts = np.zeros(N, dtype=np.int64)
    for k, a in enumerate(col):
        rts[k] = cnt[a]; cnt[a] += 1
    def shard(v, s):                                               # local index (hi || lo) <-> global (hi, s, lo)
        return v.reshape(N >> (j + rho), world, 1 << j)[:, s, :].reshape(-1)
    mine = shard(col, rank)
    def gather(arr):                                               # rank-major all-gather of equal-size int64 arrays
        out = [torch.empty(len(arr), dtype=torch.int64) for _ in range(world)]
        d.all_gather(out, torch.from_numpy(np.ascontiguousarray(arr)))
        return [o.numpy() for o in out]
    # k_cs_partition: stable sort by owner = address mod R; send key = (address >> rho) << n | global index
    li = np.arange(NL, dtype=np.int64)
    gidx = ((li >> j) << (j + rho)) | (rank << j) | (li & ((1 << j) - 1))
    owner = mine & (world - 1)
    sidx = np.argsort(owner, kind="stable")
    send = ((mine[sidx] >> rho) << n) | gidx[sidx]
    start = np.searchsorted(owner[sidx], np.arange(world + 1))    # start[o] .. start[o + 1]: the segment for owner o
    starts = gather(start.astype(np.int64))
    seg = lambda s, o: int(starts[s][o + 1] - starts[s][o])
    # forward exchange staged through an all-gather of the whole send buffers
    all_send = gather(send)
    recv = np.concatenate([all_send[p][starts[p][rank]:starts[p][rank + 1]] for p in range(world)])
    # k_cs_rank on the owner: sort by (address, global index); rank inside the run; per-address totals
    order = np.argsort(recv, kind="stable")
    skey = recv[order]
    addr = skey >> n
    first = np.ones(len(skey), dtype=bool); first[1:] = addr[1:] != addr[:-1]
    run_start = np.maximum.accumulate(np.where(first, np.arange(len(skey)), 0))
    ret = np.empty(len(skey), dtype=np.int64)
    ret[order] = np.arange(len(skey)) - run_start
    m_loc = max(M >> rho, 1)
    counts = np.bincount(addr, minlength=m_loc).astype(np.int64)
    # the way back: owner p's return buffer holds the lookups of ranks 0..me-1 first (recv_max = the common span)
    recv_max = max(sum(seg(s, o) for s in range(world)) for o in range(world))
    padded = np.zeros(recv_max, dtype=np.int64); padded[:len(ret)] = ret
    all_ret = gather(padded)
    back = np.concatenate([all_ret[p][sum(seg(s, p) for s in range(rank)):][:seg(rank, p)] for p in range(world)])
    my_rts = np.empty(NL, dtype=np.int64)
    my_rts[sidx] = back
    all_counts = np.concatenate(gather(counts))
    a = np.arange(M)
    fcs = all_counts[(a & (world - 1)) * m_loc + (a >> rho)]
    ok = bool((my_rts == shard(rts, rank)).all() and (fcs == cnt).all())
    print(json.dumps({"rank": rank, "ok": ok}), flush=True)
    hdist.barrier(d)
    d.destroy_process_group()
""") % ROOT


def test_sharded_access_counters_algorithm_two_ranks_gloo(tmp_path):
    """the N > 1 data path's one non-local step - read_ts in the GLOBAL lookup order without any rank holding a whole
    column - restated in numpy over gloo (CPU, world 2) with the formulas of the device code (global index of a local
    entry, owner = address mod R, segment offsets of both exchange directions) against the sequential definition"""
    import json
    script = tmp_path / "counters_worker.py"
    script.write_text(COUNTERS_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-3000:]
        assert json.loads(o.strip().splitlines()[-1])["ok"] is True


FLIGHT_WORKER = textwrap.dedent("""
    import os, sys, json, threading
    sys.path.insert(0, %r)
    from halo2_lasso_amd import dist as hdist
    rank, _, world = hdist.env_rank()
    d = hdist.init("gloo")
    # the channels of two sharded proofs in flight on every rank (bench.py sharded_two_in_flight): the default group for the
    # first, a flight group of its own for the second; one host thread per proof, collectives in each thread's own order
    ag = [hdist.host_all_gather(d, None), hdist.host_all_gather(d, hdist.flight_group(d))]
    bad = []
    def flight(k):
        for it in range(200):
            mine = bytes([k, rank, it %% 251]) * (1 + (it %% 7) * (k + 1))    # the two flights send different sizes
            got = ag[k](mine)
            want = b"".join(bytes([k, r, it %% 251]) * (1 + (it %% 7) * (k + 1)) for r in range(world))
            if got != want:
                bad.append((k, it))
    th = [threading.Thread(target=flight, args=(k,)) for k in (0, 1)]
    for t in th: t.start()
    for t in th: t.join()
    print(json.dumps({"rank": rank, "bad": bad}), flush=True)
    hdist.barrier(d)
    d.destroy_process_group()
""") % ROOT


def test_two_flights_on_two_groups_do_not_mix(tmp_path):
    """Two sharded proofs in flight per rank talk over two process groups from two host threads (dist.flight_group): 200
    all-gathers per flight with payloads of different sizes, issued by each thread at its own pace, every result the
    concatenation of that flight's contributions in rank order."""
    script = tmp_path / "flight_worker.py"
    script.write_text(FLIGHT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-3000:]
        assert __import__("json").loads(o.strip().splitlines()[-1])["bad"] == []
