"""Development aid: per-proof wall times of a Lasso workload (min / median / max and the slow ones) - exposes intermittent
stalls that an average hides.  usage: python tools/step_times.py [log_n] [table] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import halo2_lasso_amd as hl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "range"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ctx = hl.Context(0)
table, _ = bench.make_table(hl, kind)
pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(max(n, table.l)))
dims = [ctx.upload(c.tobytes()) for c in bench.gen_dims(table, n, 0)]
for _ in range(3):
    hl.lasso_prove(pp, table, n, dims, hl.Keccak256Transcript())
ts, phases = [], []
for _ in range(steps):
    ctx.sync()
    t0 = time.perf_counter()
    hl.lasso_prove(pp, table, n, dims, hl.Keccak256Transcript())
    ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
    phases.append(hl.lasso_last_timing(ctx))
s = sorted(ts)
print("2^%d %s: min %.2f median %.2f mean %.2f max %.2f ms over %d proofs" % (n, kind, s[0], s[len(s) // 2], sum(ts) / len(ts), s[-1], steps))
for i, t in enumerate(ts):
    if t > 1.25 * s[len(s) // 2]:
        print("  slow proof %d: %.2f ms  %s" % (i, t, {k: round(v, 2) for k, v in phases[i].items()}))
