"""GPU: ONE HyperPlonk(+Lasso) proof sharded over 2 / 4 ranks (lh_hyperplonk_prove_sharded; VERDICT r03 row e2, BASELINE.json
configs[4] "8xMI355X") must be, byte for byte, the single-GPU proof - which tests/test_gpu_keccak.py and
tests/test_gpu_hyperplonk.py tie to the Python specification and the C++ oracle (re-checked here for the small circuit).
The ranks are separate processes sharing the test box's one GPU over gloo, each holding only its shard of every
preprocess / permutation / witness poly.  Circuits: a small Keccak-f (Lasso XOR / AND lookups, copy constraints: the
rotation gather and the permutation products' gather), the full Keccak-f[1600] circuit of 2^17 rows (2 ranks; 4 ranks were run once by hand: same bytes), vanilla gates with a
32-bit AND Lasso lookup; a circuit with a LogUp lookup is refused."""
import json
import os
import random
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = textwrap.dedent("""
    import os, sys, json, random
    sys.path.insert(0, %r)
    import numpy as np


    def build(hl, ctx, cfg):
        \"\"\"(pcs params, prover param, instances, witness polys) of the test circuit - the same on every rank and in the parent\"\"\"
        from halo2_lasso_amd import hyperplonk as g_hp, keccak_circuit as kc, synthetic
        k = cfg["k"]
        prng = random.Random(cfg["seed"])
        ss = [prng.randrange(1, hl.R_MOD) for _ in range(k)]
        pcs = hl.MultilinearKzg.setup(ctx, ss)
        if cfg["circuit"] == "keccak_small":
            w, ub, rounds = cfg["w"], cfg["ub"], cfg["rounds"]
            prog = kc.keccak_program(w, ub, rounds)
            rng = np.random.default_rng(k)
            col = kc.build_columns(prog, k, rng.integers(0, 1 << w, size=((1 << k) // prog.num_rows, 25), dtype=np.uint64))
            pre, wit = kc.field_columns(col)
            info = g_hp.keccak_circuit_info(k, pre, kc.copy_cycles(col), hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 1, 2 * ub),
                                            hl.LassoTable.bitwise(hl.SUBTABLE_AND, 1, 2 * ub))
            pp = g_hp.HyperPlonk.preprocess(pcs, info)
            return pcs, pp, [[]], [hl.MultilinearPolynomial.new(ctx, x) for x in wit]
        if cfg["circuit"] == "keccak":
            circ = synthetic.keccak_f(ctx, k, seed=cfg["seed"])
        elif cfg["circuit"] == "vanilla_lasso":
            circ = synthetic.vanilla_plonk_with_lasso(ctx, k, kind="and", seed=cfg["seed"])
        else:
            circ = synthetic.vanilla_plonk_with_lookup(ctx, k, seed=cfg["seed"])
        return pcs, synthetic.prover_param(pcs, circ), circ.instances, circ.d_witness
""") % ROOT

WORKER = COMMON + textwrap.dedent("""
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)
    import halo2_lasso_amd as hl
    from halo2_lasso_amd import dist as hdist, hyperplonk as g_hp
    cfg = json.loads(sys.argv[1])
    rank, _, world = hdist.env_rank()
    d = hdist.init("gloo")
    ctx = hl.Context(0)                      # every rank on the one GPU of the test box
    pcs, pp, instances, witness = build(hl, ctx, cfg)
    sb = cfg["shard_bit"]
    pp_local = g_hp.HyperPlonk.shard_param(pp, rank, world, sb)
    wit_local = [hl.shard_poly(p, rank, world, sb) for p in witness]
    del pp, witness                          # from here on this rank holds shards only
    hl.attach_comm(ctx, rank, world, hdist.host_all_gather(d), sb)
    if "xlog" in cfg:
        hl.set_option(ctx, "shard_exchange_log", cfg["xlog"])
    if "comm_round" in cfg:                  # the rounds' partial sums by one all-reduce of u64 lanes (lasso_hip.h)
        hl.set_option(ctx, "comm_round", cfg["comm_round"])
    t = hl.Keccak256Transcript()
    err = None
    try:
        g_hp.HyperPlonk.prove_sharded(pp_local, instances, wit_local, t)
    except hl.Error as e:
        if not cfg.get("expect_error"):
            raise
        err = "%s: %s" % (type(e).__name__, e)
    with open(sys.argv[2] + ".%d" % rank, "w") as f:
        json.dump({"rank": rank, "proof": None if err else t.into_proof().hex(), "error": err, "stats": hl.comm_stats(ctx),
                   "route": hl.lasso_last_route(ctx)}, f)
    hdist.barrier(d)
    d.destroy_process_group()
""")


def run_ranks(tmp_path, world, cfg, port):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_sharded import _wait_all
    script = tmp_path / "hp_worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(r),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), json.dumps(cfg), str(tmp_path / "out")], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    return _wait_all(procs, 1500, str(tmp_path / "out"))


def single_gpu_proof(hl, ctx, cfg):
    from halo2_lasso_amd import hyperplonk as g_hp
    scope = {}
    exec(COMMON, scope)
    pcs, pp, instances, witness = scope["build"](hl, ctx, cfg)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, instances, witness, t)
    return t.into_proof()


CASES = [
    # world, circuit config, shard_bit, xlog (0: the zero-check keeps its rounds sharded until the shard bits reach bit 0)
    pytest.param(2, dict(circuit="keccak_small", w=8, ub=4, rounds=2, k=10, seed=910), 7, 0, id="keccak_small-2"),
    pytest.param(4, dict(circuit="keccak_small", w=8, ub=4, rounds=2, k=10, seed=910), 6, None, id="keccak_small-4"),
    pytest.param(2, dict(circuit="keccak_small", w=4, ub=4, rounds=1, k=9, seed=909), 7, 3, id="keccak_tiny-2"),
    # (the zero-check's sharded rounds - interpreted program, LDS-staged and streaming kernels - through the all-reduce variant)
    pytest.param(4, dict(circuit="keccak_small", w=8, ub=4, rounds=2, k=10, seed=911, comm_round=1), 6, 0, id="keccak_small-4-allreduce"),
    pytest.param(2, dict(circuit="vanilla_lasso", k=17, seed=171, comm_round=1), 15, None, marks=pytest.mark.heavy(est=25), id="vanilla_lasso_2p17-2"),
    pytest.param(2, dict(circuit="keccak", k=17, seed=16), 15, None, marks=pytest.mark.heavy(est=50), id="keccak_f1600_2p17-2"),
]


@pytest.mark.parametrize("world,cfg,shard_bit,xlog", CASES)
def test_sharded_hyperplonk_proof_equals_single_gpu(tmp_path, hl, ctx, world, cfg, shard_bit, xlog):
    cfg = dict(cfg, shard_bit=shard_bit)
    if xlog is not None:
        cfg["xlog"] = xlog
    want = single_gpu_proof(hl, ctx, cfg)
    outs = run_ranks(tmp_path, world, cfg, 29200 + (cfg["k"] * 7 + world * 13 + shard_bit) % 300)
    proofs = {o["proof"] for o in outs}
    assert len(proofs) == 1, "ranks disagree on the proof"
    assert proofs.pop() == want.hex(), "the sharded proof differs from the single-GPU proof"
    for o in outs:
        assert o["stats"]["host"] > 0 and o["route"]["sharded_rounds"] > 0 and o["route"]["shard_exchanges"] > 0, o
    if cfg["circuit"] == "keccak":
        # ... and, directly, from the C++ oracle's proof of the same Keccak-f[1600] circuit (not only through
        # the single-GPU prover)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_keccak import cpp_oracle_keccak_proof
        from halo2_lasso_amd import synthetic
        prng = random.Random(cfg["seed"])
        pcs = hl.MultilinearKzg.setup(ctx, [prng.randrange(1, hl.R_MOD) for _ in range(cfg["k"])])
        circ = synthetic.keccak_f(ctx, cfg["k"], seed=cfg["seed"])
        assert cpp_oracle_keccak_proof(hl, ctx, pcs, circ, cfg["k"]) == want, "the C++ oracle's proof differs"


def test_small_sharded_keccak_matches_the_specification(tmp_path, hl, ctx):
    """the bytes of the 2-rank proof of the small Keccak-f circuit against oracle/pyref/hyperplonk.py (not only against the
    single-GPU prover)"""
    from halo2_lasso_amd import keccak_circuit as kc
    from oracle.pyref import hyperplonk as o_hp, kzg as o_kzg, lasso as o_lasso
    from oracle.pyref.field import R_MOD as P
    from oracle.pyref.transcript import Keccak256Transcript as OT
    cfg = dict(circuit="keccak_small", w=8, ub=4, rounds=2, k=10, seed=910, shard_bit=7)
    outs = run_ranks(tmp_path, 2, cfg, 29555)
    k, w, ub = 10, 8, 4
    prog = kc.keccak_program(w, ub, 2)
    rng = np.random.default_rng(k)
    col = kc.build_columns(prog, k, rng.integers(0, 1 << w, size=((1 << k) // prog.num_rows, 25), dtype=np.uint64))
    pre, wit = kc.field_columns(col)
    o_info = o_hp.keccak_circuit_info(k, pre, kc.copy_cycles(col), o_lasso.bitwise_table(o_lasso.SUBTABLE_XOR, 1, 2 * ub),
                                      o_lasso.bitwise_table(o_lasso.SUBTABLE_AND, 1, 2 * ub))
    prng = random.Random(910)
    ss = [prng.randrange(1, P) for _ in range(k)]
    o_pp = o_hp.preprocess(o_kzg.setup(ss), o_info)
    ot = OT()
    o_hp.prove(o_pp, [[]], lambda r, ch: wit, ot)
    assert {o["proof"] for o in outs} == {ot.into_proof().hex()}
    o_hp.verify(o_pp, [[]], OT(ot.into_proof()))


def test_logup_circuits_are_refused(tmp_path):
    """a circuit with a LogUp lookup (global sort-merge join, prover.rs:139-200) does not shard: LH_ERR_ARG on every rank"""
    cfg = dict(circuit="vanilla_lookup", k=12, seed=12, shard_bit=8, expect_error=True)
    outs = run_ranks(tmp_path, 2, cfg, 29577)
    for o in outs:
        assert o["proof"] is None and "LogUp" in o["error"], o


@pytest.mark.heavy(est=45)
def test_bench_hyperplonk_keccak_two_ranks():
    """`bench.py --workload hyperplonk --lookup lasso --circuit keccak --gpus 2` (both ranks on GPU 0, gloo): ONE proof of
    the Keccak-f[1600] circuit sharded over the two ranks - strong scaling, bytes equal to the single-GPU proof."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LH_DEVICE="0", LH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "hyperplonk", "--lookup", "lasso", "--circuit",
                        "keccak", "--log-n", "17", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-profile"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["metric"] == "hyperplonk_prove_time_ms" and d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["proofs_per_step"] == 1 and d["sharded_proof_equals_single_gpu"] is True
    assert abs(d["value"] - d["ms_per_step"]) <= 1e-3
    assert d["sharded_two_in_flight"]["ms_per_proof"] > 0, d["sharded_two_in_flight"]  # two sharded proofs in flight per rank
