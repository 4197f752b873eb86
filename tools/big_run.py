"""Large configurations (BASELINE configs[2]: 2^24 AND lookups) - timing + memory, no oracle comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lasso_amd as hl
kind, n = sys.argv[1], int(sys.argv[2])
ctx = hl.Context(0)
chunks = 8 if kind.endswith("64") else 4   # and64 / xor64: 64-bit operands as 8 chunks of 8+8 bits
table = hl.LassoTable.range(2, 16) if kind == "range" else hl.LassoTable.bitwise(
    hl.SUBTABLE_AND if kind.startswith("and") else hl.SUBTABLE_XOR, chunks, 16)
rng = np.random.default_rng(1)
t = time.perf_counter()
ss = [int(v) for v in rng.integers(1, 1 << 62, size=n)]
zm = os.environ.get("LH_PCS") == "zeromorph"
nv_max = max(n, table.l)
pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, ss[0], 1 << nv_max), 1 << nv_max) if zm else hl.MultilinearKzg.setup(ctx, ss)
print("setup %.2fs" % (time.perf_counter() - t), flush=True)
dims = [ctx.upload(rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32).tobytes()) for _ in range(table.c)]
for rep in range(3):
    tr = hl.Keccak256Transcript()
    t = time.perf_counter()
    hl.lasso_prove(pp, table, n, dims, tr)
    print("%s 2^%d: %.1f ms proof %d B %s" % (kind, n, (time.perf_counter() - t) * 1e3, len(tr.into_proof()),
          {k: round(v, 1) for k, v in hl.lasso_last_timing(ctx).items()}), flush=True)
if os.environ.get("LH_VERIFY"):
    t = time.perf_counter()
    vp = hl.ZeromorphVerifierParam.setup(ss[0], 1 << nv_max, 1 << nv_max) if zm else hl.MultilinearKzgVerifierParams.setup(ss)
    hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(tr.into_proof()))
    print("proof accepted by the host verifier (%.0f ms)" % ((time.perf_counter() - t) * 1e3), flush=True)
