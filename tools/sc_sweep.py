"""Per-launch timing of the sum-check round kernels by size (tuning aid)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lasso_amd as hl
ctx = hl.Context(0)
n = 20
table = hl.LassoTable.range(2, 16)
rng = np.random.default_rng(1)
pp = hl.MultilinearKzg.setup(ctx, [int(v) for v in rng.integers(1, 1 << 62, size=n)])
dims = [ctx.upload(rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32).tobytes()) for _ in range(2)]
for _ in range(2):
    hl.lasso_prove(pp, table, n, dims, hl.Keccak256Transcript())
hl.profile_enable(ctx, True)
hl.lasso_prove(pp, table, n, dims, hl.Keccak256Transcript())
recs = hl.profile_read(ctx)
by = collections.defaultdict(list)
for r in recs:
    if r["name"].startswith("sc_round<3"):
        by[(r["name"], int(r["items"]), int(r["bytes"] / max(r["items"], 1)))].append(r["ms"])
for k in sorted(by, key=lambda k: (k[1], k[0])):
    v = by[k]
    print("%-26s size=%7d B/pair=%5d  n=%2d  avg=%.1f us" % (k[0], k[1], k[2], len(v), 1e3 * sum(v) / len(v)))
print("total", sum(r["ms"] for r in recs))
