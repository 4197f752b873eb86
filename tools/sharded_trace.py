#!/usr/bin/env python3
"""Development aid: rank `--rank` of a `--world`-rank sharded proof, alone on the GPU over the loopback communicator
(tools/sharded_rank_profile.py's set-up), as a program to put under `rocprofv3 --kernel-trace`: a few warm-up proofs, a
100 ms pause, then `--proofs` proofs back to back - tools/trace_gaps.py (LH_TRACE_SPLIT_IDLE_MS=50) takes the kernels after
the pause.  usage: rocprofv3 --kernel-trace --output-format rocpd -d DIR -- python3 tools/sharded_trace.py --config and24 --world 8"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="and24")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--proofs", type=int, default=1)
    args = ap.parse_args()
    import halo2_lasso_amd as hl
    import bench
    ctx = hl.Context(0)
    cfg = args.config
    kind, n = cfg.rstrip("0123456789"), int(cfg[len(cfg.rstrip("0123456789")):])
    if kind == "keccak":  # HyperPlonk + Lasso prove of the Keccak-f[1600] circuit of 2^n rows
        from halo2_lasso_amd import hyperplonk as hp, synthetic
        world, rank = args.world, args.rank
        rho = world.bit_length() - 1
        shard_bit = max(16 - rho, min(10, n - rho - 1), 1)
        pcs = hl.MultilinearKzg.setup(ctx, bench.trapdoor(n))
        circ = synthetic.keccak_f(ctx, n, seed=n)
        pp = synthetic.prover_param(pcs, circ)
        pp_local = hp.HyperPlonk.shard_param(pp, rank, world, shard_bit)
        wit_local = [hl.shard_poly(p, rank, world, shard_bit) for p in circ.d_witness]
        hl.attach_comm_loopback(ctx, rank, world, shard_bit)
        try:
            for _ in range(3):
                hp.HyperPlonk.prove_sharded(pp_local, circ.instances, wit_local, hl.Keccak256Transcript())
            ctx.sync()
            time.sleep(0.1)
            t0 = time.perf_counter()
            for _ in range(args.proofs):
                hp.HyperPlonk.prove_sharded(pp_local, circ.instances, wit_local, hl.Keccak256Transcript())
            ctx.sync()
            print("%s world %d rank %d: %.3f ms per proof" % (cfg, world, rank, (time.perf_counter() - t0) * 1e3 / args.proofs), flush=True)
        finally:
            hl.detach_comm(ctx)
        return
    table, _ = bench.make_table(hl, kind)
    pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(n))
    cols = bench.gen_dims(table, n, 0)
    world, rank = args.world, args.rank
    rho = world.bit_length() - 1
    shard_bit = max(table.l - rho, min(10, n - rho - 1), 1)
    if world == 1:
        shard_bit = max(shard_bit, table.l)
    d_dims = [ctx.upload(hl.shard_of(c, rank, world, shard_bit).tobytes()) for c in cols]
    hl.attach_comm_loopback(ctx, rank, world, shard_bit)
    try:
        for _ in range(3):
            hl.lasso_prove_sharded(pp, table, n, d_dims, hl.Keccak256Transcript())
        ctx.sync()
        time.sleep(0.1)
        t0 = time.perf_counter()
        for _ in range(args.proofs):
            hl.lasso_prove_sharded(pp, table, n, d_dims, hl.Keccak256Transcript())
        ctx.sync()
        print("%s world %d rank %d: %.3f ms per proof, phases %s" % (cfg, world, rank, (time.perf_counter() - t0) * 1e3 / args.proofs,
              {k: round(v, 2) for k, v in hl.lasso_last_timing(ctx).items()}), flush=True)
    finally:
        hl.detach_comm(ctx)


if __name__ == "__main__":
    main()
