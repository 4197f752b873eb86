"""HyperPlonk + LogUp prove time on one GPU for a synthetic circuit of the reference's
`vanilla_plonk_with_lookup` shape (backend/hyperplonk/util.rs:63-86,216-316; SURVEY.md §8d C5 substitute):
13 polys (pi | q_l q_r q_m q_o q_c q_lookup t_l t_r t_o | w_l w_r w_o), one 3-column lookup, the permutation
argument over the three witness columns, degree-5 zero-check.  Development / measurement aid (not bench.py):
the circuit is generated with numpy + device field ops so that 2^20..2^24 rows build in seconds, the proof
is checked by the product's host verifier, and per-kernel times come from lh_profile.

usage: python tools/hp_bench.py <log_rows> [reps]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import halo2_lasso_amd as hl  # noqa: E402
from halo2_lasso_amd import hyperplonk as hp  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    size = 1 << k
    ctx = hl.Context(0)
    lib = ctx.lib
    rng = np.random.default_rng(k)

    def rand_fr(n):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < r: a uniform-looking field element in Montgomery form
        return a

    def const_fr(v, n):
        return np.tile(np.frombuffer(hl.fr_to_bytes(v), dtype=np.uint64), (n, 1))

    def up(a):
        return hl.MultilinearPolynomial(ctx, ctx.upload(np.ascontiguousarray(a).tobytes()), k)

    def down(p):
        return np.frombuffer(p.buf.read(), dtype=np.uint64).reshape(size, 4).copy()

    def binop(fn, a, b):
        out = hl.MultilinearPolynomial(ctx, ctx.alloc(32 * size), k)
        hl._check(fn(ctx.h, a.ptr, b.ptr, size, out.ptr))
        return out

    t0 = time.perf_counter()
    rows = np.arange(size)
    live = rows < size - 1                     # the reference leaves the last row empty
    is_add, is_mul = live & (rows % 4 < 2), live & (rows % 4 == 2)
    is_lookup = live & (rows % 4 == 3)
    gate = is_add | is_mul
    zero, one, minus1 = const_fr(0, size), const_fr(1, size), const_fr(hl.R_MOD - 1, size)
    sel = lambda m, v: np.where(m[:, None], v, zero)
    q_l = q_r = sel(is_add, one)
    q_m, q_o, q_c = sel(is_mul, one), sel(gate, minus1), sel(gate, rand_fr(size))
    q_lookup = sel(is_lookup, one)
    t_l, t_r, t_o = (sel(rows >= 2, rand_fr(size)) for _ in range(3))
    t_idx = rng.integers(1, size, size=size)
    w_l, w_r = sel(gate, rand_fr(size)), sel(gate, rand_fr(size))
    w_l[is_lookup], w_r[is_lookup] = t_l[t_idx[is_lookup]], t_r[t_idx[is_lookup]]
    d_ql, d_qm, d_qc = up(q_l), up(q_m), up(q_c)

    def out_column(wl, wr):
        """w_o = q_l w_l + q_r w_r + q_m w_l w_r + q_c on gate rows (q_o = -1), t_o[t] on lookup rows"""
        a, b = up(wl), up(wr)
        lin = binop(lib.lh_fr_mul, d_ql, binop(lib.lh_fr_add, a, b))
        quad = binop(lib.lh_fr_mul, d_qm, binop(lib.lh_fr_mul, a, b))
        wo = down(binop(lib.lh_fr_add, binop(lib.lh_fr_add, lin, quad), d_qc))
        wo[is_lookup] = t_o[t_idx[is_lookup]]
        return wo

    # copies: gate rows of the second half take w_l from w_o and w_r from w_r of the row half a table earlier
    w_o = out_column(w_l, w_r)
    half = size // 2
    dst = rows[gate & (rows > half)]           # source row = dst - half >= 1
    src = dst - half
    w_l[dst], w_r[dst] = w_o[src], w_r[src]
    w_o = out_column(w_l, w_r)
    # permutation polys over (w_l, w_r, w_o) = polys 10, 11, 12 (preprocessor.rs:172-203): 2-cycles swap ids
    ident = lambda p: (np.uint64(p) << np.uint64(k)) + rows.astype(np.uint64)
    perm = [ident(0), ident(1), ident(2)]
    perm[0][dst], perm[2][src] = ident(2)[src], ident(0)[dst]      # (w_o, src) <-> (w_l, dst)
    perm[1][dst], perm[1][src] = ident(1)[src], ident(1)[dst]      # (w_r, src) <-> (w_r, dst)
    d_perm = []
    for p in perm:
        out = hl.MultilinearPolynomial(ctx, ctx.alloc(32 * size), k)
        staged = ctx.upload(p.tobytes())
        hl._check(lib.lh_fr_from_u64(ctx.h, staged.ptr, size, out.ptr))
        ctx.sync()
        d_perm.append(out)
    print("circuit 2^%d rows built in %.1f s (%d copies, %d lookups)" % (k, time.perf_counter() - t0, len(dst),
                                                                        int(is_lookup.sum())), flush=True)

    t0 = time.perf_counter()
    ss = [int(v) for v in rng.integers(1, 1 << 62, size=k)]
    pcs_pp, pcs_vp = hl.MultilinearKzg.setup(ctx, ss), hl.MultilinearKzgVerifierParams.setup(ss)
    # nine (device-resident) preprocess polys: compose() only needs their count
    info = hp.vanilla_plonk_with_lookup_circuit_info(k, 0, [[]] * 9, [[(10, 1)], [(11, 1)], [(12, 1)]])
    pp = hp.HyperPlonkProverParam()
    pp.pcs, pp.num_vars, pp.info = pcs_pp, k, info
    pp.preprocess_polys = [up(a) for a in (q_l, q_r, q_m, q_o, q_c, q_lookup, t_l, t_r, t_o)]
    pp.permutation_polys = d_perm
    pp.num_permutation_z_polys, pp.expression = hp.compose(info)
    vp = hp.HyperPlonkVerifierParam()
    vp.pcs, vp.num_vars, vp.info = pcs_vp, k, info
    vp.num_permutation_z_polys, vp.expression = pp.num_permutation_z_polys, pp.expression
    vp.preprocess_comms = hl.MultilinearKzg.batch_commit(pcs_pp, pp.preprocess_polys)
    vp.permutation_comms = hl.MultilinearKzg.batch_commit(pcs_pp, pp.permutation_polys)
    witness = [up(w_l), up(w_r), up(w_o)]
    print("setup + preprocess %.1f s" % (time.perf_counter() - t0), flush=True)

    best = None
    for rep in range(reps):
        tr = hl.Keccak256Transcript()
        ctx.sync()
        t0 = time.perf_counter()
        hp.HyperPlonk.prove(pp, [[]], witness, tr)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
        proof = tr.into_proof()
        print("hyperplonk vanilla+lookup 2^%d: %.1f ms, proof %d B" % (k, dt, len(proof)), flush=True)
    t0 = time.perf_counter()
    hp.HyperPlonk.verify(vp, [[]], hl.Keccak256Transcript.from_proof(proof))
    print("verified by the host verifier in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)

    hl.profile_enable(ctx, True)
    hp.HyperPlonk.prove(pp, [[]], witness, hl.Keccak256Transcript())
    recs = hl.profile_read(ctx)
    hl.profile_enable(ctx, False)
    agg = {}
    for r in recs:
        a = agg.setdefault(r["name"], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += r["ms"]
        a[2] += r["bytes"]
    total = sum(a[1] for a in agg.values())
    print("profiled kernels: %.1f ms total" % total)
    for name, (cnt, ms, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-28s %4d launches %8.3f ms  %7.1f GB/s" % (name, cnt, ms, by / ms / 1e6 if ms else 0))
    import json
    print(json.dumps({"workload": "hyperplonk vanilla_plonk_with_lookup", "log_rows": k, "prove_ms": round(best, 2),
                      "proof_bytes": len(proof)}))


if __name__ == "__main__":
    main()
