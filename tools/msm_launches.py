"""Development aid: every instrumented MSM launch of ONE profiled Lasso prove (name, ms, items) - what a batch's halves cost
next to the undivided batch.  usage: python tools/msm_launches.py [log_n] [table]   (env: LH_MSM_HALF_BATCHES=0/1, LH_MSM_DEBUG=1)"""
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
tot = {}
for r in recs:
    if r["name"].startswith("msm") or r["name"].startswith("lasso_counters"):
        print("%-26s %8.3f ms  items %.3g" % (r["name"], r["ms"], r["items"]))
        tot[r["name"]] = tot.get(r["name"], 0.0) + r["ms"]
print({k: round(v, 2) for k, v in tot.items()}, "route", hl.lasso_last_route(ctx).get("msm_half_batches"))
