#!/usr/bin/env python3
"""The compute half of the scaling curve, measured on ONE GPU (profiles/r03_sharded_rank_ms.json).

For world = 1, 2, 4, 8 and every rank of that world, the rank proves alone on the GPU over the loopback communicator
(lh_ctx_set_comm_loopback: every peer is a copy of this rank, device gathers are device copies): exactly the kernels,
sizes and exchange volumes the rank has in the real sharded job, no other process on the GPU, no link.  Recorded per rank:
  wall_ms   wall-clock of lh_lasso_prove_sharded (median of `--steps`): a rank's time with every collective free
  busy_ms   sum of the HIP-event durations of the instrumented launches of one separately profiled prove
  collectives, the route counters (lh_lasso_last_route), the largest kernels
and for world = 1 the plain lh_lasso_prove next to it.  What is NOT in these numbers: link time (RCCL over xGMI) and
waiting for slower peers.  The transcripts of loopback runs are not valid proofs (byte equality of the real sharded
prover is tests/test_gpu_sharded.py's business).

usage: python tools/sharded_rank_profile.py [--configs and24,range26] [--worlds 1,2,4,8] [--out FILE]
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def in_flight(ctxs, fns, steps):
    """ms per proof with one proof per ctx in flight at a time (one host thread per ctx, `steps` proofs each): what a rank
    gains when the latency-bound stretches of one sharded proof - resident rounds, MSM tails, collectives - run under the
    streaming kernels of another (bench.py two_proofs_in_flight, for shards)"""
    import threading

    def worker(fn, k):
        for _ in range(k):
            fn()
    dt = 0.0
    for k in (1, steps):
        th = [threading.Thread(target=worker, args=(fn, k)) for fn in fns]
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        for c in ctxs:
            c.sync()
        dt = time.perf_counter() - t0
    return round(dt * 1e3 / (len(fns) * steps), 3)


def hyperplonk_keccak(hl, bench, ctx, k, args):
    """the same measurement for lh_hyperplonk_prove_sharded over the loopback communicator"""
    from halo2_lasso_amd import hyperplonk as hp, synthetic
    pcs = hl.MultilinearKzg.setup(ctx, bench.trapdoor(k))
    circ = synthetic.keccak_f(ctx, k, seed=k)
    pp = synthetic.prover_param(pcs, circ)

    def timed(fn):
        for _ in range(2):
            fn()
        ts = []
        for _ in range(args.steps):
            ctx.sync()
            t0 = time.perf_counter()
            fn()
            ctx.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        hl.profile_enable(ctx, True)
        fn()
        ctx.sync()
        aggs = bench.aggregate(hl.profile_read(ctx))
        hl.profile_enable(ctx, False)
        return round(statistics.median(ts), 3), aggs

    wall, aggs = timed(lambda: hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, hl.Keccak256Transcript()))
    single = {"wall_ms": wall, "busy_ms": round(sum(a["ms"] for a in aggs), 3)}
    entry = {"workload": "HyperPlonk + Lasso prove of the Keccak-f[1600] circuit, 2^%d rows (%d permutations)" % (k, circ.num_permutations),
             "single_gpu_hyperplonk_prove": single, "worlds": {}}
    for world in [int(w) for w in args.worlds.split(",")]:
        rho = world.bit_length() - 1
        shard_bit = max(16 - rho, min(10, k - rho - 1), 1)
        if k <= shard_bit + rho:
            continue
        ranks = list(range(world)) if args.all_ranks else sorted({0, world - 1})
        per_rank = []
        for rank in ranks:
            pp_local = hp.HyperPlonk.shard_param(pp, rank, world, shard_bit)
            wit_local = [hl.shard_poly(p, rank, world, shard_bit) for p in circ.d_witness]
            hl.attach_comm_loopback(ctx, rank, world, shard_bit)
            two = None
            try:
                stats0 = hl.comm_stats(ctx)
                wall, aggs = timed(lambda: hp.HyperPlonk.prove_sharded(pp_local, circ.instances, wit_local, hl.Keccak256Transcript()))
                stats1 = hl.comm_stats(ctx)
                if args.in_flight > 1:
                    for c2 in args.more:
                        hl.attach_comm_loopback(c2, rank, world, shard_bit)
                    try:
                        views = [pp_local] + [hp.HyperPlonk.rebind_param(pp_local, c2) for c2 in args.more]
                        two = in_flight([ctx] + args.more, [
                            (lambda p=p: hp.HyperPlonk.prove_sharded(p, circ.instances, wit_local, hl.Keccak256Transcript())) for p in views],
                            args.steps)
                    finally:
                        for c2 in args.more:
                            hl.detach_comm(c2)
            finally:
                hl.detach_comm(ctx)
            del pp_local, wit_local
            per_rank.append({"rank": rank, "wall_ms": wall, "two_in_flight_ms_per_proof": two, "busy_ms": round(sum(a["ms"] for a in aggs), 3),
                             "collectives_per_proof": {kk: (stats1[kk] - stats0[kk]) // (args.steps + 3) for kk in stats0},
                             "top_kernels": [{"name": a["name"], "launches": a["launches"], "ms": round(a["ms"], 3)} for a in aggs[:8]]})
        worst_wall = max(r["wall_ms"] for r in per_rank)
        worst_busy = max(r["busy_ms"] for r in per_rank)
        entry["worlds"][str(world)] = {
            "shard_bit": shard_bit, "ranks": per_rank, "max_rank_wall_ms": worst_wall, "max_rank_busy_ms": worst_busy,
            "ideal_ms": round(single["wall_ms"] / world, 3),
            "compute_speedup_vs_single_gpu": round(single["wall_ms"] / worst_wall, 3)}
        two_txt = ""
        if args.in_flight > 1:
            worst_two = max(r["two_in_flight_ms_per_proof"] for r in per_rank)
            entry["worlds"][str(world)].update(max_rank_two_in_flight_ms_per_proof=worst_two,
                                               two_in_flight_speedup_vs_single_gpu=round(single["wall_ms"] / worst_two, 3))
            two_txt = ", %d in flight %.2f ms per proof" % (args.in_flight, worst_two)
        print("keccak%d world %d: max rank wall %.2f ms, busy %.2f ms (single GPU %.2f / %.2f)%s" % (
            k, world, worst_wall, worst_busy, single["wall_ms"], single["busy_ms"], two_txt), file=sys.stderr, flush=True)
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="and24,range26")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--all-ranks", action="store_true", help="every rank of each world (default: ranks 0 and R-1)")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="k > 1: also ms per proof with k sharded proofs in flight on the rank (k ctxs, a host thread each)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_sharded_rank_ms.json"))
    args = ap.parse_args()
    import halo2_lasso_amd as hl
    import bench
    ctx = hl.Context(0)
    args.more = [hl.Context(0) for _ in range(args.in_flight - 1)]
    result = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "configs": {}}
    for cfg in args.configs.split(","):
        kind, n = cfg.rstrip("0123456789"), int(cfg[len(cfg.rstrip("0123456789")):])
        if kind == "keccak":  # HyperPlonk + Lasso prove of the Keccak-f[1600] circuit of 2^n rows (BASELINE configs[4])
            result["configs"][cfg] = hyperplonk_keccak(hl, bench, ctx, n, args)
            continue
        table, desc = bench.make_table(hl, kind)
        pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(n))
        cols = bench.gen_dims(table, n, 0)
        entry = {"workload": desc % n, "worlds": {}}
        # the single-GPU prover
        full = [ctx.upload(c.tobytes()) for c in cols]
        for _ in range(2):
            hl.lasso_prove(pp, table, n, full, hl.Keccak256Transcript())
        ts = []
        for _ in range(args.steps):
            ctx.sync()
            t0 = time.perf_counter()
            hl.lasso_prove(pp, table, n, full, hl.Keccak256Transcript())
            ctx.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        hl.profile_enable(ctx, True)
        hl.lasso_prove(pp, table, n, full, hl.Keccak256Transcript())
        ctx.sync()
        aggs = bench.aggregate(hl.profile_read(ctx))
        hl.profile_enable(ctx, False)
        single = {"wall_ms": round(statistics.median(ts), 3), "busy_ms": round(sum(a["ms"] for a in aggs), 3),
                  "route": hl.lasso_last_route(ctx)}
        entry["single_gpu_lasso_prove"] = single
        del full
        for world in [int(w) for w in args.worlds.split(",")]:
            rho = world.bit_length() - 1
            shard_bit = max(table.l - rho, min(10, n - rho - 1), 1)
            if world == 1:
                shard_bit = max(shard_bit, table.l)
            ranks = list(range(world)) if args.all_ranks else sorted({0, world - 1})
            per_rank = []
            for rank in ranks:
                d_dims = [ctx.upload(hl.shard_of(c, rank, world, shard_bit).tobytes()) for c in cols]
                hl.attach_comm_loopback(ctx, rank, world, shard_bit)
                try:
                    for _ in range(2):
                        hl.lasso_prove_sharded(pp, table, n, d_dims, hl.Keccak256Transcript())
                    ts = []
                    for _ in range(args.steps):
                        ctx.sync()
                        t0 = time.perf_counter()
                        hl.lasso_prove_sharded(pp, table, n, d_dims, hl.Keccak256Transcript())
                        ctx.sync()
                        ts.append((time.perf_counter() - t0) * 1e3)
                    phases = hl.lasso_last_timing(ctx)
                    stats0 = hl.comm_stats(ctx)
                    hl.comm_phase_stats(ctx, reset=True)
                    hl.profile_enable(ctx, True)
                    hl.lasso_prove_sharded(pp, table, n, d_dims, hl.Keccak256Transcript())
                    ctx.sync()
                    aggs = bench.aggregate(hl.profile_read(ctx))
                    hl.profile_enable(ctx, False)
                    stats1 = hl.comm_stats(ctx)
                    by_phase = hl.comm_phase_stats(ctx, reset=True)
                    route = hl.lasso_last_route(ctx)
                    two = None
                    if args.in_flight > 1:
                        views = [pp] + [pp.view(c2) for c2 in args.more]  # (the SRS is device memory: shared, owned by `pp`)
                        for c2 in args.more:
                            hl.attach_comm_loopback(c2, rank, world, shard_bit)
                        try:
                            two = in_flight([ctx] + args.more, [
                                (lambda p=p: hl.lasso_prove_sharded(p, table, n, d_dims, hl.Keccak256Transcript())) for p in views], args.steps)
                        finally:
                            for c2 in args.more:
                                hl.detach_comm(c2)
                finally:
                    hl.detach_comm(ctx)
                del d_dims
                per_rank.append({
                    "rank": rank, "wall_ms": round(statistics.median(ts), 3), "two_in_flight_ms_per_proof": two,
                    "busy_ms": round(sum(a["ms"] for a in aggs), 3),
                    "collectives_per_proof": {k: stats1[k] - stats0[k] for k in stats0},
                    "collectives_by_phase": {k: v["collectives"] for k, v in by_phase.items() if v["collectives"]},
                    "phases_ms": {k: round(v, 3) for k, v in phases.items()},
                    "route": route,
                    "top_kernels": [{"name": a["name"], "launches": a["launches"], "ms": round(a["ms"], 3)} for a in aggs[:8]]})
            worst_wall = max(r["wall_ms"] for r in per_rank)
            worst_busy = max(r["busy_ms"] for r in per_rank)
            entry["worlds"][str(world)] = {
                "shard_bit": shard_bit, "ranks": per_rank, "max_rank_wall_ms": worst_wall, "max_rank_busy_ms": worst_busy,
                "ideal_ms": round(single["wall_ms"] / world, 3),
                "wall_over_ideal": round(worst_wall / (single["wall_ms"] / world), 3),
                "busy_over_ideal": round(worst_busy / (single["busy_ms"] / world), 3),
                "compute_speedup_vs_single_gpu": round(single["wall_ms"] / worst_wall, 3)}
            two_txt = ""
            if args.in_flight > 1:
                worst_two = max(r["two_in_flight_ms_per_proof"] for r in per_rank)
                entry["worlds"][str(world)].update(max_rank_two_in_flight_ms_per_proof=worst_two,
                                                   two_in_flight_speedup_vs_single_gpu=round(single["wall_ms"] / worst_two, 3))
                two_txt = ", %d in flight %.2f ms per proof" % (args.in_flight, worst_two)
            print("%s world %d: max rank wall %.2f ms, busy %.2f ms (single GPU %.2f / %.2f)%s" % (
                cfg, world, worst_wall, worst_busy, single["wall_ms"], single["busy_ms"], two_txt), file=sys.stderr, flush=True)
        result["configs"][cfg] = entry
        del pp
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps({k: {w: (v["worlds"][w]["max_rank_wall_ms"], v["worlds"][w]["max_rank_busy_ms"],
                              v["worlds"][w].get("max_rank_two_in_flight_ms_per_proof")) for w in v["worlds"]}
                      for k, v in result["configs"].items()}))


if __name__ == "__main__":
    main()
