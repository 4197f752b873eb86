"""GPU: ONE Lasso proof sharded over 2 / 4 ranks must produce the single-GPU proof bytes.
The ranks are separate processes that share the one GPU of the test box and talk over gloo (host all-gather
callback: RCCL refuses two ranks on one device); the RCCL transport itself is exercised by a world-1 job on the one
GPU (same code path: ncclAllGather on the prover's stream, sum-and-publish kernel) and by a two-GPU job that runs
when the box has two devices."""
import array
import json
import os
import random
import subprocess
import sys
import textwrap
import zlib

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json, array, random, faulthandler
    faulthandler.dump_traceback_later(900, exit=True)   # a hung rank reports where it sits instead of timing the test out
    sys.path.insert(0, %r)
    import numpy as np
    import halo2_lasso_amd as hl
    from halo2_lasso_amd import dist as hdist
    cfg = json.loads(sys.argv[1])
    rank, _, world = hdist.env_rank()
    d = hdist.init("gloo")
    ctx = hl.Context(0)                      # every rank on the one GPU of the test box
    rng = random.Random(cfg["seed"])
    ss = [rng.randrange(hl.R_MOD) for _ in range(cfg["n"])]
    dims = [[rng.randrange(1 << cfg["l"]) for _ in range(1 << cfg["n"])] for _ in range(cfg["c"])]
    pp = hl.MultilinearKzg.setup(ctx, ss)
    if cfg["kind"] == "nonlinear":           # g with product terms (tests/test_gpu_parity.py _nonlinear_tables)
        sys.path.insert(0, os.path.join(%r, "tests"))
        from test_gpu_parity import _nonlinear_tables
        table = _nonlinear_tables(hl, cfg["c"], cfg["l"])[1]
    else:
        table = hl.LassoTable.range(cfg["c"], cfg["l"]) if cfg["kind"] == "range" else hl.LassoTable.bitwise(
            hl.SUBTABLE_AND if cfg["kind"] == "and" else hl.SUBTABLE_XOR, cfg["c"], cfg["l"])
    # every rank holds only its shard of the lookup columns
    d_dims = [ctx.upload(hl.shard_of(np.array(col, dtype=np.uint32), rank, world, cfg["shard_bit"]).tobytes())
              for col in dims]
    hl.attach_comm(ctx, rank, world, hdist.host_all_gather(d), cfg["shard_bit"])
    if "xlog" in cfg:                        # when the residual tables of a sharded sum-check travel (0: at the last moment)
        hl.set_option(ctx, "shard_exchange_log", cfg["xlog"])
    if "comm_round" in cfg:                  # 1 / 2: the rounds' partial sums by ONE all-reduce of u64 lanes (lasso_hip.h)
        hl.set_option(ctx, "comm_round", cfg["comm_round"])
    t = hl.Keccak256Transcript()
    second = None
    if cfg.get("in_flight") == 2:
        # TWO sharded proofs in flight on every rank (bench.py sharded_two_in_flight): a second ctx on the same device with its
        # own communicator over its own gloo group, a host thread per proof; the second proof is of a DIFFERENT batch
        import threading
        ctx2 = hl.Context(0)
        hl.attach_comm(ctx2, rank, world, hdist.host_all_gather(d, hdist.flight_group(d)), cfg["shard_bit"])
        for name in ("shard_exchange_log", "comm_round"):
            hl.set_option(ctx2, name, hl.get_option(ctx, name))
        dims2 = [[rng.randrange(1 << cfg["l"]) for _ in range(1 << cfg["n"])] for _ in range(cfg["c"])]
        d_dims2 = [ctx2.upload(hl.shard_of(np.array(col, dtype=np.uint32), rank, world, cfg["shard_bit"]).tobytes())
                   for col in dims2]
        t2 = hl.Keccak256Transcript()
        errs = []
        def run(p, cols, tr):
            try:
                hl.lasso_prove_sharded(p, table, cfg["n"], cols, tr)
            except Exception as e:
                errs.append(repr(e))
        th = [threading.Thread(target=run, args=(pp, d_dims, t)), threading.Thread(target=run, args=(pp.view(ctx2), d_dims2, t2))]
        for x in th: x.start()
        for x in th: x.join()
        assert not errs, errs
        second = t2.into_proof().hex()
        hl.detach_comm(ctx2)
    else:
        hl.lasso_prove_sharded(pp, table, cfg["n"], d_dims, t)
    # (to a file: a proof of this size does not fit a pipe's buffer, and the parent reads the pipes only at the end)
    with open(sys.argv[2] + ".%%d" %% rank, "w") as f:
        json.dump({"rank": rank, "proof": t.into_proof().hex(), "second": second, "stats": hl.comm_stats(ctx),
                   "route": hl.lasso_last_route(ctx)}, f)
    hdist.barrier(d)
    d.destroy_process_group()
""") % (ROOT, ROOT)


def _wait_all(procs, timeout, out_prefix):
    """Wait for every rank; as soon as one fails the others are stopped (a dead rank leaves its peers in a collective
    for gloo's 30-minute timeout) and ITS output is reported."""
    import time
    deadline = time.time() + timeout
    while True:
        codes = [p.poll() for p in procs]
        failed = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if failed or all(c == 0 for c in codes) or time.time() > deadline:
            break
        time.sleep(0.2)
    outs = []
    for i, p in enumerate(procs):
        if p.poll() is None:
            p.kill()
        o, err = p.communicate()
        outs.append((p.returncode, o, err))
    for i in failed:
        raise AssertionError("rank %d failed (rc %s):\n%s" % (i, outs[i][0], outs[i][2][-8000:]))
    assert all(rc == 0 for rc, _, _ in outs), "timeout: " + " | ".join(e[-600:] for _, _, e in outs)
    return [json.load(open("%s.%d" % (out_prefix, i))) for i in range(len(procs))]


def run_ranks(tmp_path, world, cfg, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(r),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), json.dumps(cfg), str(tmp_path / "out")], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    return _wait_all(procs, 1000, str(tmp_path / "out"))


CASES = [
    # world, kind, c, l, n, shard_bit, xlog (None: default - these tiny tables travel before round 1; 0: every sum-check
    # keeps its rounds sharded until the shard bits reach bit 0)
    (2, "range", 2, 3, 6, 2, 0),
    (2, "and", 2, 4, 7, 3, None),
    (4, "range", 2, 4, 8, 2, 0),
    (4, "xor", 3, 4, 7, 3, 0),
    (2, "range", 2, 3, 9, 5, 0),
    (8, "and", 2, 4, 9, 2, 0),
    (2, "xor", 2, 4, 9, 4, 5),
    (4, "nonlinear", 2, 4, 8, 2, 0),
    (2, "nonlinear", 3, 4, 9, 3, None),
]


@pytest.mark.parametrize("world,kind,c,l,n,shard_bit,xlog", CASES)
def test_sharded_proof_equals_single_gpu_and_oracle(tmp_path, world, kind, c, l, n, shard_bit, xlog):
    from oracle.pyref import lasso as o_lasso, kzg as o_kzg
    from oracle.pyref.field import R_MOD
    from oracle.pyref.transcript import Keccak256Transcript as OT
    seed = zlib.crc32(repr((world, kind, c, l, n)).encode())
    cfg = dict(seed=seed, kind=kind, c=c, l=l, n=n, shard_bit=shard_bit)
    if xlog is not None:
        cfg["xlog"] = xlog
    outs = run_ranks(tmp_path, world, cfg, 29600 + (seed % 300))
    proofs = {o["proof"] for o in outs}
    assert len(proofs) == 1, "ranks disagree on the proof"
    rng = random.Random(seed)
    ss = [rng.randrange(R_MOD) for _ in range(n)]
    dims = [[rng.randrange(1 << l) for _ in range(1 << n)] for _ in range(c)]
    if kind == "nonlinear":
        import halo2_lasso_amd as hl
        from test_gpu_parity import _nonlinear_tables
        spec = _nonlinear_tables(hl, c, l)[0]
    else:
        spec = o_lasso.range_table(c, l) if kind == "range" else o_lasso.bitwise_table(
            o_lasso.SUBTABLE_AND if kind == "and" else o_lasso.SUBTABLE_XOR, c, l)
    ot = OT()
    opp = o_kzg.setup(ss)
    o_lasso.prove(opp, spec, dims, ot)
    assert proofs.pop() == ot.into_proof().hex()
    o_lasso.verify(opp, spec, n, OT(ot.into_proof()))


@pytest.mark.parametrize("world,kind,c,l,n,shard_bit,xlog", [(2, "and", 2, 4, 9, 3, 0), (4, "range", 2, 4, 9, 2, 0), (2, "xor", 2, 4, 9, 4, None)])
def test_two_sharded_proofs_in_flight_per_rank(tmp_path, world, kind, c, l, n, shard_bit, xlog):
    """every rank drives TWO sharded proofs at once - two ctxs on its device, each with its own
    communicator (here: its own gloo group), a host thread each - so that one proof's latency-bound stretches and waits for
    peers run under the other's kernels.  Both proofs (of two different batches) must be the specification's, on every rank."""
    from oracle.pyref import lasso as o_lasso, kzg as o_kzg
    from oracle.pyref.field import R_MOD
    from oracle.pyref.transcript import Keccak256Transcript as OT
    seed = zlib.crc32(repr((world, kind, c, l, n, "two in flight")).encode())
    cfg = dict(seed=seed, kind=kind, c=c, l=l, n=n, shard_bit=shard_bit, in_flight=2)
    if xlog is not None:
        cfg["xlog"] = xlog
    outs = run_ranks(tmp_path, world, cfg, 29150 + (seed % 140))
    assert len({o["proof"] for o in outs}) == 1 and len({o["second"] for o in outs}) == 1, "ranks disagree on a proof"
    rng = random.Random(seed)
    ss = [rng.randrange(R_MOD) for _ in range(n)]
    dims = [[rng.randrange(1 << l) for _ in range(1 << n)] for _ in range(c)]
    dims2 = [[rng.randrange(1 << l) for _ in range(1 << n)] for _ in range(c)]
    spec = o_lasso.range_table(c, l) if kind == "range" else o_lasso.bitwise_table(
        o_lasso.SUBTABLE_AND if kind == "and" else o_lasso.SUBTABLE_XOR, c, l)
    opp = o_kzg.setup(ss)
    for cols, got in ((dims, outs[0]["proof"]), (dims2, outs[0]["second"])):
        ot = OT()
        o_lasso.prove(opp, spec, cols, ot)
        assert got == ot.into_proof().hex()


@pytest.mark.parametrize("world,kind,c,l,n,shard_bit,comm_round", [(2, "range", 2, 3, 9, 5, 1), (4, "xor", 3, 4, 7, 3, 1),
                                                                   (8, "and", 2, 4, 9, 2, 1), (2, "nonlinear", 3, 4, 9, 3, 2)])
def test_sharded_rounds_by_all_reduce_equal_the_specification(tmp_path, world, kind, c, l, n, shard_bit, comm_round):
    """Option comm_round: every sharded sum-check round combines its partial sums by ONE all-reduce of
    u64 lanes (32-bit limb | tag << 40; over gloo the lanes are all-gathered through the host and added there - the same
    lanes, the same self-validating wait, the same lazy reduction mod r) instead of all-gather + sum-and-publish kernel:
    same proof bytes as the Python specification, on 2 / 4 / 8 ranks, with every sum-check kept sharded to the last moment."""
    from oracle.pyref import lasso as o_lasso, kzg as o_kzg
    from oracle.pyref.field import R_MOD
    from oracle.pyref.transcript import Keccak256Transcript as OT
    seed = zlib.crc32(repr((world, kind, c, l, n, "lanes")).encode())
    cfg = dict(seed=seed, kind=kind, c=c, l=l, n=n, shard_bit=shard_bit, xlog=0, comm_round=comm_round)
    outs = run_ranks(tmp_path, world, cfg, 29300 + (seed % 150))
    proofs = {o["proof"] for o in outs}
    assert len(proofs) == 1, "ranks disagree on the proof"
    assert all(o["route"]["sharded_rounds"] > 0 for o in outs)
    rng = random.Random(seed)
    ss = [rng.randrange(R_MOD) for _ in range(n)]
    dims = [[rng.randrange(1 << l) for _ in range(1 << n)] for _ in range(c)]
    if kind == "nonlinear":
        import halo2_lasso_amd as hl
        from test_gpu_parity import _nonlinear_tables
        spec = _nonlinear_tables(hl, c, l)[0]
    else:
        spec = o_lasso.range_table(c, l) if kind == "range" else o_lasso.bitwise_table(
            o_lasso.SUBTABLE_AND if kind == "and" else o_lasso.SUBTABLE_XOR, c, l)
    ot = OT()
    o_lasso.prove(o_kzg.setup(ss), spec, dims, ot)
    assert proofs.pop() == ot.into_proof().hex()


LARGE = [
    # world, kind, c, l, n, shard_bit, xlog: the streaming kernels on shards - eq-factored rounds, the read/write leaf
    # kernel, packed and derived commitments, the column-wise top quotient - checked against the C++ oracle
    pytest.param(2, "range", 2, 16, 18, 15, None, marks=pytest.mark.heavy(est=8)),
    pytest.param(4, "xor", 4, 16, 18, 14, 0, marks=pytest.mark.heavy(est=25)),
    # (the sharded prover against the ORACLE, not only against the single-GPU prover, at 2^20 lookups on 4 ranks)
    pytest.param(4, "and", 4, 16, 20, 14, None, marks=pytest.mark.heavy(est=45)),
]


@pytest.mark.parametrize("world,kind,c,l,n,shard_bit,xlog", LARGE)
def test_sharded_proof_large_vs_cpp_oracle(tmp_path, world, kind, c, l, n, shard_bit, xlog):
    import numpy as np
    import halo2_lasso_amd as hl
    from oracle import cpu_oracle as co
    seed = zlib.crc32(repr((world, kind, c, l, n)).encode())
    cfg = dict(seed=seed, kind=kind, c=c, l=l, n=n, shard_bit=shard_bit)
    if xlog is not None:
        cfg["xlog"] = xlog
    if world == 2:
        cfg["comm_round"] = 1  # (the streaming kernels' epilogues write the all-reduce variant's lanes at this size)
    outs = run_ranks(tmp_path, world, cfg, 29900 + (seed % 90))
    proofs = {o["proof"] for o in outs}
    assert len(proofs) == 1, "ranks disagree on the proof"
    assert all(o["stats"]["host"] > 0 for o in outs)
    # the shards ran the single-GPU prover's routes: factored rounds (with collectives), the leaf-layer kernel, derived
    # commitments for the bitwise tables
    for o in outs:
        r = o["route"]
        assert r["sharded_rounds"] > 0 and r["eq_factored_rounds"] > 0 and r["rw_leaf_rounds"] > 0, r
        assert r["shard_exchanges"] > 0, r
        if kind != "range":
            assert r["derived_commitments"] == c, r
    rng = random.Random(seed)
    ss = [rng.randrange(hl.R_MOD) for _ in range(n)]
    dims = [[rng.randrange(1 << l) for _ in range(1 << n)] for _ in range(c)]
    table = hl.LassoTable.range(c, l) if kind == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, c, l)
    ot = co.Transcript()
    co.lasso_prove(ot, co.setup(ss), n, table.to_c(), n, [np.array(d, dtype=np.uint32).tobytes() for d in dims])
    assert proofs.pop() == ot.into_proof().hex()


BIG_WORKER = textwrap.dedent("""
    import time
    T0 = time.time()
    import os, sys, json, hashlib, faulthandler
    faulthandler.dump_traceback_later(1700, exit=True)
    sys.path.insert(0, %r)
    import numpy as np
    import halo2_lasso_amd as hl
    from halo2_lasso_amd import dist as hdist
    import bench
    cfg = json.loads(sys.argv[1])
    rank, _, world = hdist.env_rank()
    d = hdist.init("gloo")
    ctx = hl.Context(0)                      # every rank on the one GPU of the test box
    table, _ = bench.make_table(hl, cfg["kind"])
    n = cfg["n"]
    pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(n))
    d_dims = [ctx.upload(hl.shard_of(col, rank, world, cfg["shard_bit"]).tobytes()) for col in bench.gen_dims(table, n, 0)]
    import time
    t_ready = time.time()
    hl.attach_comm(ctx, rank, world, hdist.host_all_gather(d), cfg["shard_bit"])
    if "xlog" in cfg:
        hl.set_option(ctx, "shard_exchange_log", cfg["xlog"])
    t = hl.Keccak256Transcript()
    hl.lasso_prove_sharded(pp, table, n, d_dims, t)
    proof = t.into_proof()
    with open(sys.argv[2] + ".%%d" %% rank, "w") as f:
        json.dump({"rank": rank, "sha256": hashlib.sha256(proof).hexdigest(), "bytes": len(proof), "stats": hl.comm_stats(ctx),
                   "route": hl.lasso_last_route(ctx), "phases": hl.lasso_last_timing(ctx),
                   "seconds": {"start_to_ready": t_ready - T0, "prove": time.time() - t_ready}}, f)
    hdist.barrier(d)
    d.destroy_process_group()
""") % ROOT


def run_big(tmp_path, world, cfg, port):
    script = tmp_path / "big_worker.py"
    script.write_text(BIG_WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(r),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), json.dumps(cfg), str(tmp_path / "out")], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    return _wait_all(procs, 1800, str(tmp_path / "out"))


@pytest.mark.parametrize("world,kind,n", [pytest.param(8, "range", 26, marks=pytest.mark.heavy(est=95)),
                                           pytest.param(4, "and", 24, marks=pytest.mark.heavy(est=30))])
def test_full_size_configs_sharded_on_one_gpu(tmp_path, hl, ctx, world, kind, n):
    """BASELINE.json configs[3] at FULL size - 2^26 range-check lookups, one proof sharded over 8 ranks - and configs[2]
    (2^24 AND) over 4 ranks, the ranks sharing the one GPU of the test box over gloo: every rank's proof is byte for byte
    the single-GPU prover's proof of the same lookups (which test_gpu_verify / test_gpu_parity_large tie to the verifier
    and the C++ oracle).  Exercises the sharded code at the real sizes: the repartitioned access counters on 2^23-entry
    shards, eq-factored sharded rounds, the column-wise top quotients on shards, SRS shards of 2^23 points."""
    import hashlib
    import bench
    table, _ = bench.make_table(hl, kind)
    rho = world.bit_length() - 1
    shard_bit = max(table.l - rho, 10)
    pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(n))
    full = [ctx.upload(c.tobytes()) for c in bench.gen_dims(table, n, 0)]
    t = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, full, t)
    single = t.into_proof()
    del full, pp
    # the full-size bytes are pinned to something independent of the prover: the host verifier accepts them (and rejects
    # them with one byte flipped in the middle of the memory-checking argument)
    vp = hl.MultilinearKzgVerifierParams.setup(bench.trapdoor(n))
    hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(single))
    bad = bytearray(single)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(hl.Error):
        hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(bytes(bad)))
    # (ranks that share one GPU pay ~0.5 s of GPU process switching per collective: the residual tables travel a little
    # earlier than by default - fewer sharded rounds - which changes no byte)
    outs = run_big(tmp_path, world, dict(kind=kind, n=n, shard_bit=shard_bit, xlog=20), 29480 + world + n)
    want = hashlib.sha256(single).hexdigest()
    print("ranks: seconds", outs[0]["seconds"], "phases ms", {k: round(v) for k, v in outs[0]["phases"].items()})
    for o in outs:
        assert o["bytes"] == len(single) and o["sha256"] == want, "rank %d: the sharded proof differs" % o["rank"]
        r = o["route"]
        assert r["sharded_rounds"] > 0 and r["eq_factored_rounds"] > 0 and r["rw_leaf_rounds"] > 0 and r["open_small_depth"] >= 1, r
        assert r["open_precommit"] == 1, r  # the shard's column-wise commitments ran on the helper ctx beside the sum-checks


@pytest.mark.parametrize("comm_round", [0, 1, 2])
def test_sharded_world1_over_rccl(hl, ctx, comm_round):
    """The RCCL transport on the one GPU of the test box: a world of ONE rank runs the whole sharded prover - every
    exchange an ncclAllGather on the prover's stream followed by the sum-and-publish kernel (comm_round 0), or every
    round's sums by ONE ncclAllReduce of u64 lanes into the pinned memory the host polls (1) / into device memory and a
    copy (2) - and must give the bytes of lasso_prove.  Asserts that the device-side path was taken and nothing went
    through a host callback."""
    import numpy as np
    n, shard_bit = 18, 16  # (a world of one: rho = 0, the replicated subtables need shard_bit >= 16, the tables two more variables)
    table = hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 4, 16)
    rng = np.random.default_rng(171)
    prng = random.Random(171)
    pp = hl.MultilinearKzg.setup(ctx, [prng.randrange(1, hl.R_MOD) for _ in range(n)])
    dims = [ctx.upload(rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32).tobytes()) for _ in range(4)]
    single = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, dims, single)
    hl.attach_comm_rccl(ctx, 0, 1, hl.rccl_unique_id(), shard_bit)
    hl.set_option(ctx, "comm_round", comm_round)
    try:
        t = hl.Keccak256Transcript()
        hl.lasso_prove_sharded(pp, table, n, dims, t)
        stats = hl.comm_stats(ctx)
    finally:
        hl.set_option(ctx, "comm_round", 0)
        hl.detach_comm(ctx)
    assert t.into_proof() == single.into_proof()
    assert stats["device"] >= 15 and stats["host"] == 0, stats


P2P_SELF_WORKER = textwrap.dedent("""
    import sys, random
    sys.path.insert(0, %r)
    import numpy as np
    import halo2_lasso_amd as hl
    ctx = hl.Context(0)
    n, shard_bit = 18, 16
    table = hl.LassoTable.bitwise(hl.SUBTABLE_AND, 4, 16)
    rng = np.random.default_rng(181)
    prng = random.Random(181)
    pp = hl.MultilinearKzg.setup(ctx, [prng.randrange(1, hl.R_MOD) for _ in range(n)])
    cols = [rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32) for _ in range(4)]
    cols[0][rng.random(1 << n) < 0.4] = 9                  # a hot address: long runs for the owner's ranking
    dims = [ctx.upload(c.tobytes()) for c in cols]
    single = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, dims, single)
    hl.attach_comm_rccl(ctx, 0, 1, hl.rccl_unique_id(), shard_bit)
    t = hl.Keccak256Transcript()
    hl.lasso_prove_sharded(pp, table, n, dims, t)
    stats = hl.comm_stats(ctx)
    by_phase = hl.comm_phase_stats(ctx)
    hl.detach_comm(ctx)
    assert t.into_proof() == single.into_proof(), "bytes differ"
    assert stats["host"] == 0 and by_phase["witness"]["collectives"] >= 4, (stats, by_phase)   # boundaries, keys out, ranks back, counts: all columns together
    print("P2P-OK", stats, by_phase["witness"])
""") % ROOT


def test_rccl_grouped_send_recv_executes_on_one_gpu():
    """the personalised exchange of the sharded access counters (grouped ncclSend / ncclRecv, csrc/comm.cpp)
    had never executed - a world of one skipped the repartitioned counters and the exchange skipped the self segment.
    With LH_SHARDED_COUNTERS_MIN_R=1 and LH_COMM_A2A_SELF=1 a world of ONE rank over real RCCL takes the repartitioned
    counters and sends its own segment to itself through ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the
    prover's stream: symbols, group semantics and stream ordering run before the first multi-GPU job does; the proof
    must be lasso_prove's."""
    env = dict(os.environ, LH_SHARDED_COUNTERS_MIN_R="1", LH_COMM_A2A_SELF="1")
    r = subprocess.run([sys.executable, "-c", P2P_SELF_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "P2P-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


PROBE_FAIL_WORKER = textwrap.dedent("""
    import sys, random
    sys.path.insert(0, %r)
    import numpy as np
    import halo2_lasso_amd as hl
    ctx = hl.Context(0)
    n, shard_bit = 18, 16
    table = hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 4, 16)
    rng = np.random.default_rng(191)
    prng = random.Random(191)
    pp = hl.MultilinearKzg.setup(ctx, [prng.randrange(1, hl.R_MOD) for _ in range(n)])
    dims = [ctx.upload(rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32).tobytes()) for _ in range(4)]
    single = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, dims, single)
    hl.attach_comm_rccl(ctx, 0, 1, hl.rccl_unique_id(), shard_bit)   # (the probe runs here - and is told to fail)
    hl.set_option(ctx, "comm_round", 1)
    t = hl.Keccak256Transcript()
    hl.lasso_prove_sharded(pp, table, n, dims, t)
    stats = hl.comm_stats(ctx)
    hl.detach_comm(ctx)
    assert t.into_proof() == single.into_proof(), "bytes differ"
    print("PROBE-FALLBACK-OK", stats)
""") % ROOT


def test_all_reduce_into_host_memory_is_probed_at_attach():
    """comm_round = 1 lets ncclAllReduce write straight into pinned host memory; whether a fabric accepts that across ranks
    is unknown until a multi-GPU job runs.  lh_ctx_set_comm_rccl therefore runs the collective once at attach
    (comm.cpp comm_probe_host_recv: tagged lanes, bounded wait, the ranks agree on the verdict through a device-side
    all-reduce); a failed probe makes comm_round = 1 behave as 2 with a note on stderr.  Here the probe is told to fail
    (LH_COMM_PROBE_FAIL=1) in a world of one over real RCCL: the note appears, the proof is lasso_prove's.  (The passing
    probe runs in every other RCCL test of this file.)"""
    env = dict(os.environ, LH_COMM_PROBE_FAIL="1")
    r = subprocess.run([sys.executable, "-c", PROBE_FAIL_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PROBE-FALLBACK-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "comm_round = 1 runs as 2" in r.stderr, r.stderr[-2000:]


RCCL_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    import halo2_lasso_amd as hl
    from halo2_lasso_amd import dist as hdist
    rank, local_rank, world = hdist.env_rank()
    d = hdist.init("nccl")
    ctx = hl.Context(local_rank)             # one rank per GPU
    n, shard_bit = 18, 15
    table = hl.LassoTable.range(2, 16)
    rng = np.random.default_rng(5)
    import random
    prng = random.Random(5)
    pp = hl.MultilinearKzg.setup(ctx, [prng.randrange(1, hl.R_MOD) for _ in range(n)])
    cols = [rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32) for _ in range(2)]
    single = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, [ctx.upload(c.tobytes()) for c in cols], single)
    assert hdist.attach_sharded(ctx, d, shard_bit) == "rccl"
    t = hl.Keccak256Transcript()
    hl.lasso_prove_sharded(pp, table, n, [ctx.upload(hl.shard_of(c, rank, world, shard_bit).tobytes()) for c in cols], t)
    with open(sys.argv[2] + ".%%d" %% rank, "w") as f:
        json.dump({"rank": rank, "same": t.into_proof() == single.into_proof(), "stats": hl.comm_stats(ctx)}, f)
    hl.detach_comm(ctx)
    hdist.barrier(d)
    d.destroy_process_group()
""") % ROOT


def test_sharded_two_gpus_over_rccl(tmp_path):
    """two ranks, two GPUs, RCCL over xGMI (skipped on a one-GPU box): bytes of the single-GPU proof, no host callback"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29871", WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), "-", str(tmp_path / "out")], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    for out in _wait_all(procs, 600, str(tmp_path / "out")):
        assert out["same"] and out["stats"]["device"] > 20 and out["stats"]["host"] == 0, out


@pytest.mark.heavy(est=10)
def test_bench_gpus_2_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts the two ranks itself (both on GPU 0
    here, rendezvous over gloo) and relays rank 0's line: n_gpus == 2, one sharded proof per step."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LH_DEVICE="0", LH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "17", "--table",
                        "range", "--no-cpu-baseline", "--no-inflight", "--no-extra", "--no-profile"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["proofs_per_step"] == 1
    assert d["sharded_proof_equals_single_gpu"] is True


@pytest.mark.heavy(est=10)
def test_bench_two_ranks_launched_like_the_driver():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` exactly as the driver launches it,
    except that both ranks share GPU 0 (LH_DEVICE) and rendezvous over gloo: one JSON line, from rank 0; by default
    the two ranks prove ONE sharded proof (strong scaling), `--mode replicas` proves one batch per rank."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, LH_DEVICE="0", LH_DIST_BACKEND="gloo")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--log-n", "17", "--table", "range", "--no-cpu-baseline", "--no-inflight"] + \
            ([] if "--extra" in extra else ["--no-extra"]) + [e for e in extra if e != "--extra"]
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-1500:]
        return json.loads(lines[0])

    d = run([])
    assert "replicas" not in d
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["metric"] == "lasso_prove_time_ms" and d["higher_is_better"] is False
    assert abs(d["value"] - d["ms_per_step"]) <= 1e-3          # one proof per step, whatever the number of ranks
    assert d["config"]["lookups_per_proof"] == 1 << 17 and d["config"]["proofs_per_step"] == 1
    assert d["config"]["transport"] == "host" and d["sharded_proof_equals_single_gpu"] is True
    d = run(["--mode", "replicas"])
    assert d["scaling"] == "weak" and d["config"]["proofs_per_step"] == 2
    assert abs(d["value"] - d["ms_per_step"] / 2) <= 1e-3       # two proofs per step
    # the default N > 1 line with its extra objects: configs[3] (here 2^18 instead of 2^26) sharded, and replicas
    os.environ["LH_BENCH_CONFIG3_LOG_N"] = "18"
    try:
        d = run(["--extra"])
    finally:
        del os.environ["LH_BENCH_CONFIG3_LOG_N"]
    assert d["scaling"] == "strong" and d["sharded_proof_equals_single_gpu"] is True
    assert d["config3_2p26_range_sharded"]["ms_per_proof"] > 0 and d["replicas"]["proofs_per_step"] == 2
    # ... the A/B of the two link-dependent switches (four runs, a best one; byte equality is asserted inside bench.py when the
    # headline is re-timed with it) and two sharded proofs in flight per rank
    assert len(d["ab"]["runs"]) == 4 and all(r["ms_per_proof"] > 0 for r in d["ab"]["runs"]) and "error" not in d["ab"], d["ab"]
    assert {(r["comm_round"], r["shard_exchange_log"]) for r in d["ab"]["runs"]} == {(0, 19), (0, 20), (1, 19), (1, 20)}
    assert d["ab"]["best"]["ms_per_proof"] <= min(r["ms_per_proof"] for r in d["ab"]["runs"]) + 1e-6
    assert d["sharded_two_in_flight"]["ms_per_proof"] > 0, d["sharded_two_in_flight"]
