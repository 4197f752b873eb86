"""GPU: Zeromorph over univariate KZG (pcs/multilinear/zeromorph.rs, pcs/univariate/kzg.rs) against the oracle
(oracle/pyref/zeromorph.py) byte for byte, in the shape of the reference's PCS tests
(pcs/multilinear.rs:293-406 run_commit_open_verify / run_batch_commit_open_verify), then through both verifiers."""
import random

import pytest

from oracle.pyref import zeromorph as o_zm, kzg as o_kzg
from oracle.pyref.field import R_MOD as P
from oracle.pyref.poly import evaluate
from oracle.pyref.transcript import Keccak256Transcript as OT

pytestmark = pytest.mark.gpu


def test_univariate_setup_matches_oracle(hl, ctx):
    s = random.Random(1).randrange(1, P)
    params = hl.Zeromorph.setup(ctx, s, 37)
    assert params.size == 37 and params.powers() == o_zm.setup(s, 37).powers_g1
    again = hl.Zeromorph.upload(ctx, params.powers())
    assert again.powers() == params.powers()
    with pytest.raises(hl.InvalidPcsParam):
        hl.Zeromorph.trim(params, 64)


@pytest.mark.parametrize("num_vars,extra", [(1, 0), (2, 0), (3, 5), (6, 0), (7, 0), (10, 3)])
def test_zeromorph_commit_open_verify(hl, ctx, num_vars, extra):
    rng = random.Random(10 * num_vars + extra)
    s = rng.randrange(1, P)
    size = (1 << num_vars) + extra
    o_param = o_zm.setup(s, size)
    o_pp, o_vp = o_zm.trim(o_param, 1 << num_vars)
    pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, s, size), 1 << num_vars)
    vp = hl.ZeromorphVerifierParam.setup(s, size, 1 << num_vars)
    evals = [rng.randrange(P) for _ in range(1 << num_vars)]
    poly = hl.MultilinearPolynomial.new(ctx, evals)
    comm = hl.Zeromorph.commit(pp, poly)
    assert comm == o_zm.commit(o_pp, evals)
    point = [rng.randrange(P) for _ in range(num_vars)]
    ev = evaluate(evals, point)
    ot = OT()
    o_zm.open_(o_pp, evals, point, ev, ot)
    t = hl.Keccak256Transcript()
    hl.Zeromorph.open(pp, poly, point, t)
    proof = t.into_proof()
    assert proof == ot.into_proof() and len(proof) == 64 * (num_vars + 2)
    o_zm.verify(o_vp, comm, point, ev, OT(proof))
    hl.Zeromorph.verify(vp, comm, point, ev, hl.Keccak256Transcript.from_proof(proof))
    with pytest.raises(hl.InvalidPcsOpen, match="Invalid Zeromorph KZG open"):
        hl.Zeromorph.verify(vp, comm, point, (ev + 1) % P, hl.Keccak256Transcript.from_proof(proof))
    bad = bytearray(proof)
    bad[-7] ^= 1
    with pytest.raises(hl.Error):
        hl.Zeromorph.verify(vp, comm, point, ev, hl.Keccak256Transcript.from_proof(bytes(bad)))


@pytest.mark.parametrize("num_vars,batch", [(2, 2), (4, 3), (8, 4)])
def test_zeromorph_batch_commit_open_verify(hl, ctx, num_vars, batch):
    rng = random.Random(50 + num_vars)
    s = rng.randrange(1, P)
    o_pp, o_vp = o_zm.trim(o_zm.setup(s, 1 << num_vars), 1 << num_vars)
    pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, s, 1 << num_vars), 1 << num_vars)
    vp = hl.ZeromorphVerifierParam.setup(s, 1 << num_vars, 1 << num_vars)
    tables = [[rng.randrange(P) for _ in range(1 << num_vars)] for _ in range(batch)]
    polys = [hl.MultilinearPolynomial.new(ctx, t) for t in tables]
    pairs = [(p, q) for p in range(batch) for q in range(2) if (p + q) % 3 != 2]
    ot, t = OT(), hl.Keccak256Transcript()
    o_comms = o_zm.batch_commit_and_write(o_pp, tables, ot)
    comms = hl.Zeromorph.batch_commit_and_write(pp, polys, t)
    assert comms == o_comms
    o_pts = [ot.squeeze_challenges(num_vars) for _ in range(2)]
    pts = [t.squeeze_challenges(num_vars) for _ in range(2)]
    assert pts == o_pts
    vals = [evaluate(tables[p], pts[q]) for p, q in pairs]
    ot.write_field_elements(vals), t.write_field_elements(vals)
    o_zm.batch_open(o_pp, num_vars, tables, pts, [o_kzg.Evaluation(p, q, v) for (p, q), v in zip(pairs, vals)], ot)
    hl.Zeromorph.batch_open(pp, num_vars, polys, pts, [hl.Evaluation(p, q, v) for (p, q), v in zip(pairs, vals)], t)
    proof = t.into_proof()
    assert proof == ot.into_proof()
    for verifier in ("oracle", "product"):
        r = OT(proof) if verifier == "oracle" else hl.Keccak256Transcript.from_proof(proof)
        c2 = r.read_commitments(batch)
        p2 = [r.squeeze_challenges(num_vars) for _ in range(2)]
        v2 = r.read_field_elements(len(pairs))
        if verifier == "oracle":
            o_zm.batch_verify(o_vp, num_vars, c2, p2, [o_kzg.Evaluation(p, q, v) for (p, q), v in zip(pairs, v2)], r)
        else:
            hl.Zeromorph.batch_verify(vp, num_vars, c2, p2, [hl.Evaluation(p, q, v) for (p, q), v in zip(pairs, v2)], r)
            assert r.remaining() == 0


def test_zeromorph_large_open_verifies(hl, ctx):
    """2^16 coefficients: several levels of the suffix-Horner recursion; checked by the pairing verifier"""
    import numpy as np
    num_vars = 16
    rng = random.Random(77)
    s = rng.randrange(1, P)
    pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, s, 1 << num_vars), 1 << num_vars)
    vp = hl.ZeromorphVerifierParam.setup(s, 1 << num_vars, 1 << num_vars)
    raw = np.random.default_rng(5).integers(0, 1 << 60, size=(1 << num_vars, 4), dtype=np.uint64)  # < 2^252 < r
    poly = hl.MultilinearPolynomial(ctx, ctx.upload(raw.tobytes()), num_vars)
    comm = hl.Zeromorph.commit(pp, poly)
    point = [rng.randrange(P) for _ in range(num_vars)]
    ev = poly.evaluate(point)
    t = hl.Keccak256Transcript()
    hl.Zeromorph.open(pp, poly, point, t)
    hl.Zeromorph.verify(vp, comm, point, ev, hl.Keccak256Transcript.from_proof(t.into_proof()))
    with pytest.raises(hl.InvalidPcsOpen):
        hl.Zeromorph.verify(vp, comm, point[::-1], ev, hl.Keccak256Transcript.from_proof(t.into_proof()))


@pytest.mark.parametrize("kind,c,l,n", [("range", 2, 3, 4), ("and", 2, 4, 3), ("xor", 2, 4, 6)])
def test_lasso_over_zeromorph_matches_oracle(hl, ctx, kind, c, l, n):
    """the Lasso argument with the other PCS: same protocol bytes up to the commitments / opening proof"""
    import array
    from oracle.pyref import lasso as o_lasso
    rng = random.Random(700 + n)
    s = rng.randrange(1, P)
    nv = max(n, l)
    spec = o_lasso.range_table(c, l) if kind == "range" else o_lasso.bitwise_table(
        o_lasso.SUBTABLE_AND if kind == "and" else o_lasso.SUBTABLE_XOR, c, l)
    table = hl.LassoTable.range(c, l) if kind == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, c, l)
    dims = [[rng.randrange(1 << l) for _ in range(1 << n)] for _ in range(c)]
    o_pp, o_vp = o_zm.trim(o_zm.setup(s, 1 << nv), 1 << nv)
    ot = OT()
    o_lasso.prove(o_pp, spec, dims, ot, pcs=o_zm)
    pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, s, 1 << nv), 1 << nv)
    vp = hl.ZeromorphVerifierParam.setup(s, 1 << nv, 1 << nv)
    t = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, [ctx.upload(array.array("I", d).tobytes()) for d in dims], t)
    proof = t.into_proof()
    assert proof == ot.into_proof()
    o_lasso.verify(o_vp, spec, n, OT(proof), pcs=o_zm)
    hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(proof))
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 8
    with pytest.raises(hl.Error):
        hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(bytes(bad)))
