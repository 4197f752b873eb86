"""Generates tests/golden/vectors.json from the Python oracle (oracle/pyref).

The reference (Rust) cannot be built or run in this environment and holds no golden vectors of its
own, so these vectors pin the ORACLE (regression) and give the HIP path fixed
bytes to reproduce; they are not outputs of the reference.  Run:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyref import gkr, kzg, lasso, sum_check as sc, expression as ex, hyperplonk as hp, zeromorph as zm  # noqa: E402
from oracle.pyref.field import R_MOD as P  # noqa: E402
from oracle.pyref.poly import evaluate, eq_xy  # noqa: E402
from oracle.pyref.transcript import Keccak256Transcript as T  # noqa: E402

H = lambda xs: [hex(x) for x in xs]


def main():
    rng = random.Random(20261002)
    out = {}
    ss = [rng.randrange(P) for _ in range(5)]
    pp = kzg.setup(ss)
    out["srs"] = {"ss": H(ss), "eqs_level3": [[hex(c) for c in pt] for pt in pp.eqs[3]]}

    # sum-check, EvaluationsProver: eq * (c0 p0 p1 + c1 p2), degree 3
    nv = 4
    tabs = [[rng.randrange(P) for _ in range(1 << nv)] for _ in range(3)]
    y = [rng.randrange(P) for _ in range(nv)]
    c = [rng.randrange(P) for _ in range(2)]
    expr = ex.EqXY(0) * (ex.Poly(0) * ex.Poly(1) * c[0] + ex.Poly(2) * c[1])
    eq = eq_xy(y)
    claim = sum(eq[b] * (c[0] * tabs[0][b] * tabs[1][b] + c[1] * tabs[2][b]) for b in range(1 << nv)) % P
    t = T()
    x, ev = sc.prove(sc.EvaluationsProver, nv, sc.VirtualPolynomial(expr, tabs, [], [y]), claim, t)
    out["sum_check_eval"] = {"num_vars": nv, "tables": [H(a) for a in tabs], "y": H(y), "coeffs": H(c),
                             "claim": hex(claim), "x": H(x), "evals": H(ev), "proof": t.into_proof().hex()}

    # sum-check, CoefficientsProver: sum_j s_j eq_j p_j
    ys = [[rng.randrange(P) for _ in range(nv)] for _ in range(2)]
    s = [rng.randrange(P) for _ in range(2)]
    expr = ex.sum_exprs(ex.EqXY(j) * ex.Poly(j) * s[j] for j in range(2))
    claim = sum(s[j] * evaluate(tabs[j], ys[j]) for j in range(2)) % P
    t = T()
    x, ev = sc.prove(sc.CoefficientsProver, nv, sc.VirtualPolynomial(expr, tabs[:2], [], ys), claim, t)
    out["sum_check_coeff"] = {"num_vars": nv, "tables": [H(a) for a in tabs[:2]], "ys": [H(a) for a in ys],
                              "scalars": H(s), "claim": hex(claim), "x": H(x), "evals": H(ev),
                              "proof": t.into_proof().hex()}

    # fractional sum-check GKR, 2 fractions
    nv = 3
    tabs = [[rng.randrange(P) for _ in range(1 << nv)] for _ in range(4)]
    t = T()
    p_xs, q_xs, x = gkr.prove_fractional_sum_check([None] * 2, [None] * 2, tabs[:2], tabs[2:], t)
    out["frac_gkr"] = {"num_vars": nv, "ps": [H(a) for a in tabs[:2]], "qs": [H(a) for a in tabs[2:]],
                       "p_xs": H(p_xs), "q_xs": H(q_xs), "x": H(x), "proof": t.into_proof().hex()}

    # grand product, mixed depth
    vs = [[rng.randrange(1, P) for _ in range(n)] for n in (8, 4, 8)]
    t = T()
    roots, claims = gkr.prove_grand_product(vs, t)
    out["grand_product"] = {"leaves": [H(v) for v in vs], "roots": H(roots),
                            "claims": [[hex(cl), H(pt)] for cl, pt in claims], "proof": t.into_proof().hex()}

    # mKZG batch commit + batch open
    nv = 3
    polys = [[rng.randrange(P) for _ in range(1 << nv)] for _ in range(3)]
    t = T()
    kzg.batch_commit_and_write(pp, polys, t)
    pts = [t.squeeze_challenges(nv) for _ in range(2)]
    pairs = [(0, 0), (1, 0), (2, 1), (0, 1)]
    vals = [evaluate(polys[p], pts[q]) for p, q in pairs]
    t.write_field_elements(vals)
    kzg.batch_open(pp.trim(nv), nv, polys, pts, [kzg.Evaluation(p, q, v) for (p, q), v in zip(pairs, vals)], t)
    out["kzg_batch"] = {"num_vars": nv, "polys": [H(a) for a in polys], "pairs": pairs, "proof": t.into_proof().hex()}

    # Lasso proofs
    out["lasso"] = []
    for kind, c_, l, n in (("range", 2, 3, 4), ("and", 2, 4, 5), ("xor", 2, 4, 4)):
        spec = lasso.range_table(c_, l) if kind == "range" else lasso.bitwise_table(
            lasso.SUBTABLE_AND if kind == "and" else lasso.SUBTABLE_XOR, c_, l)
        dims = [[rng.randrange(1 << l) for _ in range(1 << n)] for _ in range(c_)]
        t = T()
        lasso.prove(pp, spec, dims, t)
        proof = t.into_proof()
        lasso.verify(pp, spec, n, T(proof))
        out["lasso"].append({"kind": kind, "c": c_, "l": l, "n": n, "dims": dims, "proof": proof.hex()})
    # identically zero columns commit to the identity (lasso.py write_commitments): an all-zero high limb (dim_1 = E_1
    # = 0), pairwise distinct indices (read_ts_0 = 0); own rng: the vectors above stay unchanged
    lrng = random.Random(4242)
    perm = list(range(16))
    lrng.shuffle(perm)
    for c_, l, n, dims in ((2, 3, 4, [[lrng.randrange(8) for _ in range(16)], [0] * 16]),
                           (2, 4, 4, [perm, [lrng.randrange(16) for _ in range(16)]])):
        spec = lasso.range_table(c_, l)
        t = T()
        lasso.prove(pp, spec, dims, t)
        proof = t.into_proof()
        lasso.verify(pp, spec, n, T(proof))
        assert int.from_bytes(proof[:32], "big") != 0  # the identity mask is not empty
        out["lasso"].append({"kind": "range", "c": c_, "l": l, "n": n, "dims": dims, "proof": proof.hex()})

    # HyperPlonk proofs of the reference's sample circuits (own rng: earlier vectors stay unchanged)
    out["hyperplonk"] = []
    for nv, with_lookup in ((3, False), (3, True), (4, True)):
        crng = random.Random(1000 + 2 * nv + with_lookup)
        gen = hp.rand_vanilla_plonk_with_lookup_circuit if with_lookup else hp.rand_vanilla_plonk_circuit
        info, instances, witness = gen(nv, crng)
        hpp = hp.preprocess(pp, info)
        t = T()
        hp.prove(hpp, instances, lambda rnd, ch: witness, t)
        proof = t.into_proof()
        hp.verify(hpp, instances, T(proof))
        assert not with_lookup or len(proof) == 1024 + 352 * nv
        out["hyperplonk"].append({"num_vars": nv, "with_lookup": with_lookup,
                                  "preprocess_polys": [H(a) for a in info.preprocess_polys],
                                  "permutations": [[list(c) for c in cyc] for cyc in info.permutations],
                                  "instances": [H(a) for a in instances], "witness": [H(a) for a in witness],
                                  "proof": proof.hex()})

    # Lasso as HyperPlonk's lookup argument (hyperplonk.py LassoLookup): vanilla gates + one AND lookup proven by Lasso
    out["hyperplonk_lasso"] = []
    for nv, kind, c_, l in ((4, "range", 2, 2), (5, "and", 2, 4)):
        spec = lasso.range_table(c_, l) if kind == "range" else lasso.bitwise_table(lasso.SUBTABLE_AND, c_, l)
        seed = 2000 + nv
        while True:  # a tiny circuit can have an identically zero witness column: the reference's transcript rejects
            crng = random.Random(seed)  # the identity commitment (transcript.rs:172-179); take the next seed then
            info, instances, witness = hp.rand_vanilla_plonk_with_lasso_circuit(nv, crng, spec)
            hpp = hp.preprocess(pp, info)
            t = T()
            try:
                hp.prove(hpp, instances, lambda rnd, ch: witness, t)
                break
            except Exception as e:
                assert "Invalid elliptic curve point" in str(e)
                seed += 100
        proof = t.into_proof()
        hp.verify(hpp, instances, T(proof))
        out["hyperplonk_lasso"].append({"num_vars": nv, "kind": kind, "c": c_, "l": l,
                                        "preprocess_polys": [H(a) for a in info.preprocess_polys],
                                        "permutations": [[list(c) for c in cyc] for cyc in info.permutations],
                                        "instances": [H(a) for a in instances], "witness": [H(a) for a in witness],
                                        "proof": proof.hex()})

    # Zeromorph over univariate KZG: one opening with a trim offset, one batch opening (own rng)
    zrng = random.Random(777)
    s_ = zrng.randrange(1, P)
    nv = 3
    zpp, zvp = zm.trim(zm.setup(s_, (1 << nv) + 2), 1 << nv)
    tabs = [[zrng.randrange(P) for _ in range(1 << nv)] for _ in range(3)]
    point = [zrng.randrange(P) for _ in range(nv)]
    t = T()
    zm.open_(zpp, tabs[0], point, evaluate(tabs[0], point), t)
    zm.verify(zvp, zm.commit(zpp, tabs[0]), point, evaluate(tabs[0], point), T(t.into_proof()))
    tb = T()
    zm.batch_commit_and_write(zpp, tabs, tb)
    pts = [tb.squeeze_challenges(nv) for _ in range(2)]
    pairs = [[0, 0], [1, 0], [2, 1], [1, 1]]
    vals = [evaluate(tabs[p], pts[q]) for p, q in pairs]
    tb.write_field_elements(vals)
    zm.batch_open(zpp, nv, tabs, pts, [kzg.Evaluation(p, q, v) for (p, q), v in zip(pairs, vals)], tb)
    out["zeromorph"] = {"s": hex(s_), "param_size": (1 << nv) + 2, "num_vars": nv, "polys": [H(a) for a in tabs],
                        "point": H(point), "commitment": [hex(c) for c in zm.commit(zpp, tabs[0])],
                        "open_proof": t.into_proof().hex(), "pairs": pairs, "batch_proof": tb.into_proof().hex()}

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote vectors.json")


if __name__ == "__main__":
    main()
