#!/usr/bin/env python3
"""Soak run (GPU box): the same proof over and over on one ctx - bytes identical every time, per-proof wall time and the
arena's high-water mark steady, route counters identical.  Catches what a parity test of one proof cannot: a resident
kernel that sometimes falls back, a sequence number that wraps, a leak in the arena or the pinned blocks.
usage: python tools/soak.py [--log-n 16 --table range --proofs 500]"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--table", default="range")
    ap.add_argument("--proofs", type=int, default=500)
    args = ap.parse_args()
    import halo2_lasso_amd as hl
    import bench
    ctx = hl.Context(0)
    table, desc = bench.make_table(hl, args.table)
    n = args.log_n
    pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(max(n, table.l)))
    bufs = [ctx.upload(c.tobytes()) for c in bench.gen_dims(table, n, 0)]
    first = route0 = None
    times, marks = [], []
    for i in range(args.proofs):
        tr = hl.Keccak256Transcript()
        t0 = time.perf_counter()
        hl.lasso_prove(pp, table, n, bufs, tr)
        times.append((time.perf_counter() - t0) * 1e3)
        proof, route = tr.into_proof(), hl.lasso_last_route(ctx)
        if first is None:
            first, route0 = proof, route
        assert proof == first, "proof %d differs from the first" % i
        if i >= 2:  # (the first proofs build caches: SRS levels, the helper ctx)
            assert route == route1, "proof %d took another route: %r against %r" % (i, route, route1)
        if i == 1:
            route1 = route
        marks.append(hl.memory_stats(ctx)["arena_high_water_bytes"])
    assert marks[-1] == marks[len(marks) // 2], "the arena's high-water mark still grows: %r" % (marks[len(marks) // 2:][::50],)
    t = sorted(times[2:])
    print("%s: %d proofs, bytes identical, route steady, arena high water %.1f MiB; ms per proof median %.3f, p90 %.3f, p99 %.3f, "
          "max %.3f" % (desc % n, args.proofs, marks[-1] / 2**20, statistics.median(t), t[int(len(t) * 0.9)],
                        t[int(len(t) * 0.99)], t[-1]))


if __name__ == "__main__":
    main()
