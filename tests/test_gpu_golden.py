"""GPU: the HIP path must reproduce the committed golden vectors (tests/golden/vectors.json) byte for byte."""
import array
import json
import os

import pytest

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
I = lambda xs: [int(x, 16) for x in xs]


@pytest.fixture(scope="module")
def pp5(hl, ctx):
    pp = hl.MultilinearKzg.setup(ctx, I(GOLDEN["srs"]["ss"]))
    assert [[hex(c) for c in pt] for pt in pp.eqs()[3]] == GOLDEN["srs"]["eqs_level3"]
    return pp


def test_golden_sum_check_eval(hl, ctx):
    g = GOLDEN["sum_check_eval"]
    c = I(g["coeffs"])
    polys = [hl.MultilinearPolynomial.new(ctx, I(a)) for a in g["tables"]]
    t = hl.Keccak256Transcript()
    x, ev = hl.ClassicSumCheck.prove(ctx, hl.EvaluationsProver, g["num_vars"],
                                     hl.SumOfProducts([(c[0], [0, 1]), (c[1], [2])], global_eq=0), polys, [I(g["y"])],
                                     int(g["claim"], 16), t)
    assert (x, ev, t.into_proof().hex()) == (I(g["x"]), I(g["evals"]), g["proof"])


def test_golden_sum_check_coeff(hl, ctx):
    g = GOLDEN["sum_check_coeff"]
    s = I(g["scalars"])
    polys = [hl.MultilinearPolynomial.new(ctx, I(a)) for a in g["tables"]]
    t = hl.Keccak256Transcript()
    x, ev = hl.ClassicSumCheck.prove(ctx, hl.CoefficientsProver, g["num_vars"],
                                     hl.SumOfProducts([(s[j], [2 + j, j]) for j in range(2)]), polys,
                                     [I(a) for a in g["ys"]], int(g["claim"], 16), t)
    assert (x, ev, t.into_proof().hex()) == (I(g["x"]), I(g["evals"]), g["proof"])


def test_golden_gkr(hl, ctx):
    g = GOLDEN["frac_gkr"]
    ps = [hl.MultilinearPolynomial.new(ctx, I(a)) for a in g["ps"]]
    qs = [hl.MultilinearPolynomial.new(ctx, I(a)) for a in g["qs"]]
    t = hl.Keccak256Transcript()
    out = hl.prove_fractional_sum_check(ctx, [None] * 2, [None] * 2, ps, qs, t)
    assert out == (I(g["p_xs"]), I(g["q_xs"]), I(g["x"])) and t.into_proof().hex() == g["proof"]
    g = GOLDEN["grand_product"]
    t = hl.Keccak256Transcript()
    roots, claims = hl.prove_grand_product(ctx, [hl.MultilinearPolynomial.new(ctx, I(v)) for v in g["leaves"]], t)
    assert roots == I(g["roots"]) and t.into_proof().hex() == g["proof"]
    assert [[hex(c), [hex(v) for v in p]] for c, p in claims] == g["claims"]


def test_golden_kzg_batch(hl, ctx, pp5):
    g = GOLDEN["kzg_batch"]
    nv = g["num_vars"]
    polys = [hl.MultilinearPolynomial.new(ctx, I(a)) for a in g["polys"]]
    t = hl.Keccak256Transcript()
    hl.MultilinearKzg.batch_commit_and_write(pp5, polys, t)
    pts = [t.squeeze_challenges(nv) for _ in range(2)]
    vals = [polys[p].evaluate(pts[q]) for p, q in g["pairs"]]
    t.write_field_elements(vals)
    hl.MultilinearKzg.batch_open(pp5, nv, polys, pts, [hl.Evaluation(p, q, v) for (p, q), v in zip(g["pairs"], vals)], t)
    assert t.into_proof().hex() == g["proof"]


@pytest.mark.parametrize("idx", range(5))  # 3, 4: identically zero columns (identity commitments)
def test_golden_lasso(hl, ctx, pp5, idx):
    g = GOLDEN["lasso"][idx]
    table = hl.LassoTable.range(g["c"], g["l"]) if g["kind"] == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if g["kind"] == "and" else hl.SUBTABLE_XOR, g["c"], g["l"])
    d_dims = [ctx.upload(array.array("I", d).tobytes()) for d in g["dims"]]
    t = hl.Keccak256Transcript()
    hl.lasso_prove(pp5, table, g["n"], d_dims, t)
    assert t.into_proof().hex() == g["proof"]


@pytest.mark.parametrize("idx", range(3))
def test_golden_hyperplonk(hl, ctx, pp5, idx):
    from halo2_lasso_amd import hyperplonk as g_hp
    g = GOLDEN["hyperplonk"][idx]
    nv = g["num_vars"]
    mk = g_hp.vanilla_plonk_with_lookup_circuit_info if g["with_lookup"] else g_hp.vanilla_plonk_circuit_info
    perms = [[tuple(c) for c in cyc] for cyc in g["permutations"]]
    info = mk(nv, len(g["instances"][0]), [I(a) for a in g["preprocess_polys"]], perms)
    pp = g_hp.HyperPlonk.preprocess(pp5, info)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, [I(a) for a in g["instances"]],
                          [hl.MultilinearPolynomial.new(ctx, I(w)) for w in g["witness"]], t)
    assert t.into_proof().hex() == g["proof"]


@pytest.mark.parametrize("idx", range(2))
def test_golden_hyperplonk_lasso(hl, ctx, pp5, idx):
    """Lasso as HyperPlonk's lookup argument against the committed proof of the specification"""
    from halo2_lasso_amd import hyperplonk as g_hp
    g = GOLDEN["hyperplonk_lasso"][idx]
    nv = g["num_vars"]
    table = hl.LassoTable.range(g["c"], g["l"]) if g["kind"] == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND, g["c"], g["l"])
    perms = [[tuple(c) for c in cyc] for cyc in g["permutations"]]
    info = g_hp.vanilla_plonk_with_lasso_circuit_info(nv, len(g["instances"][0]), [I(a) for a in g["preprocess_polys"]],
                                                      perms, table)
    pp = g_hp.HyperPlonk.preprocess(pp5, info)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, [I(a) for a in g["instances"]],
                          [hl.MultilinearPolynomial.new(ctx, I(w)) for w in g["witness"]], t)
    assert t.into_proof().hex() == g["proof"]


def test_golden_zeromorph(hl, ctx):
    g = GOLDEN["zeromorph"]
    nv, s = g["num_vars"], int(g["s"], 16)
    pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, s, g["param_size"]), 1 << nv)
    polys = [hl.MultilinearPolynomial.new(ctx, I(a)) for a in g["polys"]]
    assert [hex(c) for c in hl.Zeromorph.commit(pp, polys[0])] == g["commitment"]
    t = hl.Keccak256Transcript()
    hl.Zeromorph.open(pp, polys[0], I(g["point"]), t)
    assert t.into_proof().hex() == g["open_proof"]
    t = hl.Keccak256Transcript()
    hl.Zeromorph.batch_commit_and_write(pp, polys, t)
    pts = [t.squeeze_challenges(nv) for _ in range(2)]
    vals = [polys[p].evaluate(pts[q]) for p, q in g["pairs"]]
    t.write_field_elements(vals)
    hl.Zeromorph.batch_open(pp, nv, polys, pts, [hl.Evaluation(p, q, v) for (p, q), v in zip(g["pairs"], vals)], t)
    assert t.into_proof().hex() == g["batch_proof"]
