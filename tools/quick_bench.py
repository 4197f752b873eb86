"""Ad-hoc timing of the Lasso prove and micro-kernels on one GPU (development aid, not bench.py)."""
import array
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lasso_amd as hl  # noqa: E402
import numpy as np  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:]] or [16, 20]
    ctx = hl.Context(0)
    lib = ctx.lib
    rng = np.random.default_rng(1)
    # ---- micro: Fr mul chain, bind
    n = 1 << 22
    raw = rng.integers(0, 1 << 62, size=n * 4, dtype=np.uint64).tobytes()  # < r as 4x62-bit limbs... top limb small
    a, b, out = ctx.upload(raw), ctx.upload(raw[::-1]), ctx.alloc(32 * n)
    for iters in (64,):
        lib.lh_fr_mul_chain(ctx.h, a.ptr, b.ptr, n, iters, out.ptr)
        ctx.sync()
        t = time.perf_counter()
        lib.lh_fr_mul_chain(ctx.h, a.ptr, b.ptr, n, iters, out.ptr)
        ctx.sync()
        dt = time.perf_counter() - t
        print("fr_mul_chain: %.1f G mul/s" % (n * iters / dt / 1e9))
    x = hl._fr_array([12345])
    for m in (22,):
        nn = 1 << m
        lib.lh_fix_var(ctx.h, a.ptr, nn, x, out.ptr)
        ctx.sync()
        t = time.perf_counter()
        for _ in range(10):
            lib.lh_fix_var(ctx.h, a.ptr, nn, x, out.ptr)
        ctx.sync()
        dt = (time.perf_counter() - t) / 10
        print("fix_var 2^%d: %.1f us, %.0f GB/s (96 B per output)" % (m, dt * 1e6, 96 * (nn / 2) / dt / 1e9))
    del a, b, out

    for n in ns:
        l, c = 16, 2
        t = time.perf_counter()
        ss = [int(v) for v in rng.integers(1, 1 << 62, size=max(n, l))]
        pp = hl.MultilinearKzg.setup(ctx, ss)
        print("setup 2^%d: %.2f s" % (n, time.perf_counter() - t))
        dims = [ctx.upload(rng.integers(0, 1 << l, size=1 << n, dtype=np.uint32).tobytes()) for _ in range(c)]
        table = hl.LassoTable.range(c, l)
        for rep in range(3):
            tr = hl.Keccak256Transcript()
            t = time.perf_counter()
            hl.lasso_prove(pp, table, n, dims, tr)
            dt = time.perf_counter() - t
            ph = hl.lasso_last_timing(ctx)
            print("lasso range 2^%d: %.1f ms  proof %d B  " % (n, dt * 1e3, len(tr.into_proof())) +
                  " ".join("%s=%.1f" % kv for kv in ph.items()))
        pp.free()


if __name__ == "__main__":
    main()
