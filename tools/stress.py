"""Development aid: many proofs of varying shapes from two host threads with their own contexts (two resident sum-check
tails polling their hosts at the same time, two MSM pipelines, one GPU), every proof checked by the host verifier.
usage: python tools/stress.py [proofs_per_thread]"""
import os
import resource
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import halo2_lasso_amd as hl  # noqa: E402

COUNT = int(sys.argv[1]) if len(sys.argv) > 1 else 150
errors = []


def worker(tid):
    try:
        ctx = hl.Context(0)
        rng = np.random.default_rng(3 + tid)
        ss = [int(v) for v in rng.integers(1, 1 << 62, size=16)]
        pp = hl.MultilinearKzg.setup(ctx, ss)
        vp = hl.MultilinearKzgVerifierParams.setup(ss)
        tables = [hl.LassoTable.range(2, 16), hl.LassoTable.bitwise(hl.SUBTABLE_AND, 4, 16),
                  hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 8, 16)]
        for it in range(COUNT):
            tb = tables[(it + tid) % 3]
            nn = 11 + (it * 7 + tid) % 5  # fewer lookups than ~2^10 into a 2^16 table are often all distinct: a zero column,
            # whose identity commitment the transcript rejects (as the reference does)
            dims = [ctx.upload(rng.integers(0, 1 << 16, size=1 << nn, dtype=np.uint32).tobytes()) for _ in range(tb.c)]
            tr = hl.Keccak256Transcript()
            hl.lasso_prove(pp, tb, nn, dims, tr)
            hl.lasso_verify(vp, tb, nn, hl.Keccak256Transcript.from_proof(tr.into_proof()))
    except Exception as e:  # noqa: BLE001
        errors.append((tid, repr(e)))


t0 = time.time()
threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
for t in threads:
    t.start()
for t in threads:
    t.join()
assert not errors, errors
print("2 x %d proofs proved and verified in %.1f s, rss MB %d" % (COUNT, time.time() - t0,
                                                                  resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024))
