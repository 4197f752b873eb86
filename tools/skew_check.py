"""Development aid: 2^22-lookup proofs over heavily skewed lookup columns (a hot cell with half of the accesses, Zipf,
all lookups into one cell) for the AND / range / 64-bit XOR tables, every proof checked by the host verifier: the paths
that depend on the data (packed read_ts pairs only when the counts are small, offsets and widths of the column-wise top
quotient, continuation levels of the bucket accumulation).  usage: python tools/skew_check.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import halo2_lasso_amd as hl
ctx = hl.Context(0)
rng = np.random.default_rng(5)
n = 22
ss = [int(v) for v in rng.integers(1, 1 << 62, size=n)]
pp = hl.MultilinearKzg.setup(ctx, ss)
vp = hl.MultilinearKzgVerifierParams.setup(ss)
for kind, table in (("and", hl.LassoTable.bitwise(hl.SUBTABLE_AND, 4, 16)), ("range", hl.LassoTable.range(2, 16)),
                    ("xor64", hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 8, 16))):
    for skew in ("hot", "zipf", "const"):
        cols = []
        for j in range(table.c):
            if skew == "hot":
                v = rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32)
                v[rng.random(1 << n) < 0.5] = 7
            elif skew == "zipf":
                v = (rng.zipf(1.3, size=1 << n) % (1 << 16)).astype(np.uint32)
            else:
                v = np.full(1 << n, 513 + j, dtype=np.uint32)
            cols.append(ctx.upload(v.tobytes()))
        tr = hl.Keccak256Transcript()
        t = time.perf_counter()
        hl.lasso_prove(pp, table, n, cols, tr)
        ms = (time.perf_counter() - t) * 1e3
        hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(tr.into_proof()))
        print(kind, skew, "2^%d: %.1f ms, verified" % (n, ms), flush=True)
