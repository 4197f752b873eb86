"""Micro-workloads mirroring the reference's criterion benches:
  benches/pcs.rs:26,63-124     MultilinearKzg commit / open at 2^16..2^20
  fractional_sum_check.rs:329  GKR fractional sum-check, 3 batched fractions
plus the two stream ceilings every roofline in DESIGN.md is priced against (bind GB/s beyond the 256 MiB
Infinity Cache, Fr multiplications/s).  GPU = HIP path through the C-ABI, CPU = oracle/cpu on all cores.
Writes one JSON document (stdout).

Not a pytest module (a measurement script: `python tests/micro_bench.py 20`); it lives under tests/ because it
times the oracle next to the GPU path, and only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import halo2_lasso_amd as hl  # noqa: E402
from oracle import cpu_oracle as co  # noqa: E402


def rand_fr_bytes(rng, n):
    # 4 x 62-bit limbs with a clear top: a valid Montgomery representative (< r) without big-int work
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= (1 << 59) - 1
    return a.tobytes()


def gpu_time(ctx, fn, reps=3):
    fn()
    ctx.sync()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ctx.sync()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


def main():
    max_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = hl.Context(0)
    lib = ctx.lib
    rng = np.random.default_rng(7)
    out = {"host_cores": co.num_threads()}

    # ---- stream ceilings
    n = 1 << 25  # 1 GiB table: well beyond the Infinity Cache
    a = ctx.upload(rand_fr_bytes(rng, n))
    o = ctx.alloc(16 * n)
    x = hl._fr_array([0x1234567])
    ms = gpu_time(ctx, lambda: lib.lh_fix_var(ctx.h, a.ptr, n, x, o.ptr))
    gb = 96.0 * (n / 2) / ms / 1e6
    out["fix_var_2p25"] = {"ms": ms, "GBps_algorithmic": gb, "frac_of_8TBps": gb / 8000.0,
                           "bytes": "96 B per bound entry (SURVEY.md 8d); a 1 GiB table: beyond the 256 MiB MALL"}
    n22 = 1 << 22
    ms = gpu_time(ctx, lambda: lib.lh_fix_var(ctx.h, a.ptr, n22, x, o.ptr), reps=10)
    gb = 96.0 * (n22 / 2) / ms / 1e6
    out["fix_var_2p22"] = {"ms": ms, "GBps_algorithmic": gb, "frac_of_8TBps": gb / 8000.0, "bytes": "a 128 MiB table: MALL-resident"}
    m = 1 << 22
    b = ctx.alloc(32 * m)
    ms = gpu_time(ctx, lambda: lib.lh_fr_mul_chain(ctx.h, a.ptr, a.ptr + 32 * m, m, 64, b.ptr))
    peak_mul = m * 64 / ms / 1e6  # G Fr-mul/s: the reference every `alu` fraction below is taken against
    out["fr_mul_chain"] = {"ms": ms, "G_mul_per_s": peak_mul}
    del a, o, b

    # ---- mKZG commit / open (benches/pcs.rs)
    ss = [int(v) for v in rng.integers(1, 1 << 62, size=max_n)]
    pp = hl.MultilinearKzg.setup(ctx, ss)
    srs = C.create_string_buffer(64 * ((2 << max_n) - 1))
    hl._check(lib.lh_srs_download(ctx.h, pp.h, srs))
    out["mkzg"] = []
    for nv in range(16, max_n + 1, 2):
        raw = rand_fr_bytes(rng, 1 << nv)
        poly = hl.MultilinearPolynomial(ctx, ctx.upload(raw), nv)
        point = [int(v) for v in rng.integers(1, 1 << 62, size=nv)]
        g_commit = gpu_time(ctx, lambda: hl.MultilinearKzg.commit(pp, poly))
        g_open = gpu_time(ctx, lambda: hl.MultilinearKzg.open(pp, poly, point, hl.Keccak256Transcript()))
        cpu = {}
        if nv <= 20:
            outp = C.create_string_buffer(64)
            t = time.perf_counter()
            co._chk(co.lib().orc_commit(srs, C.c_size_t(max_n), raw, C.c_size_t(nv), outp))
            cpu["commit_ms"] = (time.perf_counter() - t) * 1e3
            tr = co.Transcript()
            ev = C.create_string_buffer(32)
            t = time.perf_counter()
            co._chk(co.lib().orc_open(tr.h, srs, C.c_size_t(max_n), raw, C.c_size_t(nv), co.fr_bytes(point), ev))
            cpu["open_ms"] = (time.perf_counter() - t) * 1e3
            gt = hl.Keccak256Transcript()
            hl.MultilinearKzg.open(pp, poly, point, gt)
            cpu["open_bytes_equal"] = gt.into_proof() == tr.into_proof()
            cpu["commit_equal"] = hl.MultilinearKzg.commit(pp, poly) == co.g1_point(outp.raw)
        # algorithmic bytes: commit = MSM of 2^nv points, 96 B each; open = the quotient pass (128 B per pair
        # at every level: 128 (2^nv - 1)) + the MSMs of the nv quotients (96 B per point, 2^nv - 1 points); the MSM is
        # integer-ALU work: 10 Fq products per mixed addition x (255 / window) windows per point, against the measured chain
        npts = 1 << nv
        cbits = max(4, min(17, nv - 4))
        wins = -(-255 // cbits)
        out["mkzg"].append({"num_vars": nv, "gpu_commit_ms": g_commit, "gpu_open_ms": g_open,
                            "gpu_commit_Mpts_per_s": npts / g_commit / 1e3,
                            "commit_GBps_algorithmic": 96.0 * npts / g_commit / 1e6,
                            "commit_frac_of_mul_chain": 10.0 * wins * npts / g_commit / 1e6 / peak_mul,
                            "open_GBps_algorithmic": (128.0 + 96.0) * (npts - 1) / g_open / 1e6, **cpu})

    # ---- GKR fractional sum-check, 3 fractions (fractional_sum_check.rs:327-370)
    out["frac_gkr"] = []
    for nv in (16, 20):
        if nv > max_n:
            continue
        raws = [rand_fr_bytes(rng, 1 << nv) for _ in range(6)]
        polys = [hl.MultilinearPolynomial(ctx, ctx.upload(r), nv) for r in raws]
        g = gpu_time(ctx, lambda: hl.prove_fractional_sum_check(ctx, [None] * 3, [None] * 3, polys[:3], polys[3:],
                                                                hl.Keccak256Transcript()))
        # algorithmic bytes: layer-up 192 B per output and fraction at every level (192 x 3 x 2^nv in all), the layer
        # sum-checks 96 B per bound entry over 4 x 3 + 1 tables, levels of 2^(nv-1) .. 2 entries (96 x 13 x 2^nv in all)
        rec = {"num_vars": nv, "batch": 3, "gpu_ms": g, "GBps_algorithmic": (192.0 * 3 + 96.0 * 13) * (1 << nv) / g / 1e6}
        if nv <= 16:
            arr, keep = co._ptrs(raws[:3])
            arr2, keep2 = co._ptrs(raws[3:])
            px, qx, xx = (C.create_string_buffer(96), C.create_string_buffer(96), C.create_string_buffer(32 * nv))
            tr = co.Transcript()
            t = time.perf_counter()
            co._chk(co.lib().orc_frac_gkr_prove(tr.h, C.c_size_t(3), C.c_size_t(nv), arr, arr2, px, qx, xx))
            rec["cpu_ms"] = (time.perf_counter() - t) * 1e3
            gt = hl.Keccak256Transcript()
            hl.prove_fractional_sum_check(ctx, [None] * 3, [None] * 3, polys[:3], polys[3:], gt)
            rec["bytes_equal"] = gt.into_proof() == tr.into_proof()
        out["frac_gkr"].append(rec)
    # ---- zero-check: ClassicSumCheck<EvaluationsProver> over vanilla_plonk_expression (benches/zero_check.rs:24-42),
    # random tables (the prover's work does not depend on the claim being true), sum claimed 0 as in the bench
    from halo2_lasso_amd import hyperplonk as g_hp
    out["zero_check"] = []
    for nv in range(20, 24):
        info = g_hp.vanilla_plonk_circuit_info(nv, 0, [[]] * 5, [[(6, 1)], [(7, 1)], [(8, 1)]])
        nz, expr = g_hp.compose(info)
        polys = [hl.MultilinearPolynomial(ctx, ctx.upload(rand_fr_bytes(rng, 1 << nv)), nv) for _ in range(13)]
        challenges = [int(v) for v in rng.integers(1, 1 << 62, size=3)]
        ys = [[int(v) for v in rng.integers(1, 1 << 62, size=nv)]]
        g = gpu_time(ctx, lambda: hl.sum_check_prove_expression(ctx, nv, expr, polys, challenges, ys, 0,
                                                                hl.Keccak256Transcript()))
        # 96 B per bound entry over the 13 polys (the eq table is factored out of the streaming rounds): 96 x 13 x 2^nv; the
        # compiled round is integer-ALU work; CPU: the oracle's expression sum-check is only exposed
        # inside hyperplonk_prove (bench.py --workload hyperplonk times it)
        out["zero_check"].append({"num_vars": nv, "gpu_ms": g, "polys": 13, "degree": expr.degree(),
                                  "GBps_algorithmic": 96.0 * 13 * (1 << nv) / g / 1e6,
                                  "frac_of_8TBps": 96.0 * 13 * (1 << nv) / g / 1e6 / 8000.0})
        del polys
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
