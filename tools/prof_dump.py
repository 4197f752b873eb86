"""Development aid: every instrumented launch of ONE profiled Lasso prove, in order (name, ms, GB/s algorithmic).
usage: python tools/prof_dump.py [log_n] [table]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import halo2_lasso_amd as hl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
kind = sys.argv[2] if len(sys.argv) > 2 else "and"
ctx = hl.Context(0)
table, _ = bench.make_table(hl, kind)
pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(max(n, table.l)))
dims = [ctx.upload(c.tobytes()) for c in bench.gen_dims(table, n, 0)]
for _ in range(2):
    hl.lasso_prove(pp, table, n, dims, hl.Keccak256Transcript())
hl.profile_enable(ctx, True)
hl.lasso_prove(pp, table, n, dims, hl.Keccak256Transcript())
recs = hl.profile_read(ctx)
hl.profile_enable(ctx, False)
t = 0.0
for r in recs:
    t += r["ms"]
    print("%8.3f  %-28s %8.3f ms  %7.1f GB/s  items %.3g" % (t, r["name"], r["ms"], r["bytes"] / max(r["ms"], 1e-9) / 1e6, r["items"]))
