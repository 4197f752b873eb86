import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lasso_amd as hl
import resource
ctx = hl.Context(0)
rng = np.random.default_rng(3)
n = 14
ss = [int(v) for v in rng.integers(1, 1 << 62, size=16)]
pp = hl.MultilinearKzg.setup(ctx, ss)
vp = hl.MultilinearKzgVerifierParams.setup(ss)
tables = [hl.LassoTable.range(2, 16), hl.LassoTable.bitwise(hl.SUBTABLE_AND, 4, 16), hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 8, 16)]
t0 = time.time()
for it in range(150):
    tb = tables[it % 3]
    nn = 10 + it % 5
    dims = [ctx.upload(rng.integers(0, 1 << 16, size=1 << nn, dtype=np.uint32).tobytes()) for _ in range(tb.c)]
    tr = hl.Keccak256Transcript()
    hl.lasso_prove(pp, tb, nn, dims, tr)
    if it % 10 == 0:
        hl.lasso_verify(vp, tb, nn, hl.Keccak256Transcript.from_proof(tr.into_proof()))
    if it % 50 == 0:
        print(it, "rss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, flush=True)
print("150 proofs ok in %.1f s, rss MB %d" % (time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024))
