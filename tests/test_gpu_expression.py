"""GPU: ClassicSumCheck<EvaluationsProver> over general Expressions -- the reference's four sum-check
scenarios (piop/sum_check.rs:196-350: lagrange, rotation, vanilla plonk, vanilla plonk with lookup),
HIP path vs the oracle byte for byte, then the oracle's verifier + final-point check (sum_check.rs:159-175)."""
import random

import pytest

from oracle.pyref import expression as oex, sum_check as o_sc, hyperplonk as o_hp
from oracle.pyref.bh import BooleanHypercube
from oracle.pyref.field import R_MOD as P
from oracle.pyref.transcript import Keccak256Transcript as OT

pytestmark = pytest.mark.gpu


def both(hl, build):
    """the same expression in the oracle's and in the product's AST"""
    return build(oex, oex.Poly, oex.distribute_powers), build(hl.expression, hl.expression.Polynomial,
                                                              hl.expression.distribute_powers)


def run(hl, ctx, num_vars, o_expr, g_expr, tables, challenges, y, claim=0):
    ot = OT()
    ox, oev = o_sc.prove(o_sc.EvaluationsProver, num_vars, o_sc.VirtualPolynomial(o_expr, tables, challenges, [y]), claim, ot)
    polys = [hl.MultilinearPolynomial.new(ctx, t) for t in tables]
    t = hl.Keccak256Transcript()
    x, ev = hl.sum_check_prove_expression(ctx, num_vars, g_expr, polys, challenges, [y], claim, t)
    assert (x, ev) == (ox, oev)
    assert t.into_proof() == ot.into_proof()
    final, vx = o_sc.verify(o_sc.Evaluations, num_vars, oex.degree(o_expr), claim, OT(t.into_proof()))
    assert vx == x
    # re-evaluate the expression at the final point (sum_check.rs:159-175)
    evals = {(i, 0): v for i, v in enumerate(ev)}
    for (poly, rot) in oex.used_query(o_expr):
        if rot != 0:
            evals[(poly, rot)] = o_hp.rotation_eval(x, rot, o_hp.evaluate_for_rotation(tables[poly], x, rot))
    assert final == o_hp.evaluate_expression(o_expr, num_vars, evals, challenges, [y], x)


@pytest.mark.parametrize("num_vars", [2, 3])
def test_sum_check_lagrange(hl, ctx, num_vars):
    """sum_check.rs:196-246: gates l_i - poly_i with poly_i one-hot on row bh[i]"""
    rng = random.Random(num_vars)

    def build(m, poly, dp):
        gates = [m.Lagrange(i) - poly(i) for i in range(1 << num_vars)]
        return dp(gates, m.Challenge(0)) * m.EqXY(0)
    o_expr, g_expr = both(hl, build)
    tables = []
    for b in BooleanHypercube(num_vars).iter():
        t = [0] * (1 << num_vars)
        t[b] = 1
        tables.append(t)
    run(hl, ctx, num_vars, o_expr, g_expr, tables, [rng.randrange(P)], [rng.randrange(P) for _ in range(num_vars)])


@pytest.mark.parametrize("num_vars", [2, 3, 5, 8])
def test_sum_check_rotation(hl, ctx, num_vars):
    """sum_check.rs:248-301: poly_k queried at rotation (n-1-k) equals poly_{k+1} at rotation (n-2-k)"""
    rng = random.Random(100 + num_vars)
    rots = list(range(num_vars - 1, -num_vars, -1))

    def build(m, poly, dp):
        ps = [poly(i, r) for i, r in enumerate(rots)]
        gates = [ps[i + 1] - ps[i] for i in range(len(ps) - 1)]
        return dp(gates, m.Challenge(0)) * m.EqXY(0)
    o_expr, g_expr = both(hl, build)
    bh = BooleanHypercube(num_vars)
    cur = [rng.randrange(P) for _ in range(1 << num_vars)]
    tables = [cur]
    for _ in range(2 * num_vars - 2):
        cur = [cur[bh.rotate(b, 1)] for b in range(1 << num_vars)]
        tables.append(cur)
    run(hl, ctx, num_vars, o_expr, g_expr, tables, [rng.randrange(P)], [rng.randrange(P) for _ in range(num_vars)])


def _assignment(num_vars, with_lookup, rng):
    """rand_vanilla_plonk[_with_lookup]_assignment (util.rs:172-214,318-374)"""
    gen = o_hp.rand_vanilla_plonk_with_lookup_circuit if with_lookup else o_hp.rand_vanilla_plonk_circuit
    info, instances, witness = gen(num_vars, rng)
    polys = o_hp.instance_polys(num_vars, instances) + info.preprocess_polys + witness
    beta, gamma, alpha = (rng.randrange(P) for _ in range(3))
    perm_idx = info.permutation_polys()
    perm = o_hp.permutation_polys(num_vars, perm_idx, info.permutations)
    z = o_hp.permutation_z_polys(1, list(zip(perm_idx, perm)), polys, beta, gamma)
    extra = []
    if with_lookup:
        comp = o_hp.lookup_compressed_polys(info.lookups, polys, [], [pow(beta, i, P) for i in range(3)])
        m = [o_hp.lookup_m_poly(c) for c in comp]
        h = [o_hp.lookup_h_poly(c, mm, gamma) for c, mm in zip(comp, m)]
        extra = m + h
    return info, polys + perm + extra + z, [beta, gamma, alpha]


@pytest.mark.parametrize("with_lookup", [False, True])
@pytest.mark.parametrize("num_vars", [2, 3, 6])
def test_zero_check_vanilla_plonk(hl, ctx, num_vars, with_lookup):
    """sum_check.rs:303-349 (vanilla_plonk / vanilla_plonk_with_lookup expressions, degree 5)"""
    from halo2_lasso_amd import hyperplonk as g_hp
    rng = random.Random(7 * num_vars + with_lookup)
    info, tables, challenges = _assignment(num_vars, with_lookup, rng)
    _, o_expr = o_hp.compose(info)
    if with_lookup:
        q_lookup, t_l, t_r, t_o = (hl.expression.Polynomial(i) for i in range(6, 10))
        w = [hl.expression.Polynomial(i) for i in range(10, 13)]
        lookups = [[(q_lookup * w[0], t_l), (q_lookup * w[1], t_r), (q_lookup * w[2], t_o)]]
        base = 10
    else:
        lookups, base = [], 6
    pi, q_l, q_r, q_m, q_o, q_c = (hl.expression.Polynomial(i) for i in range(6))
    w_l, w_r, w_o = (hl.expression.Polynomial(base + i) for i in range(3))
    gate = q_l * w_l + q_r * w_r + q_m * w_l * w_r + q_o * w_o + q_c + pi
    g_info = g_hp.PlonkishCircuitInfo(num_vars, info.num_instances, info.preprocess_polys, [3], [0], [gate], lookups,
                                      info.permutations, 4)
    num_z, g_expr = g_hp.compose(g_info)
    assert num_z == 1 and g_expr.degree() == oex.degree(o_expr) == 5
    run(hl, ctx, num_vars, o_expr, g_expr, tables, challenges, [rng.randrange(P) for _ in range(num_vars)])


def test_monomial_fallback_path_still_matches_golden():
    """the expanded-monomial round kernel (used when an expression does not fit the register program) is selected
    with LH_EXPR_MONOMIALS=1 at library load: run the HyperPlonk golden parity tests in a fresh process with it"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LH_EXPR_MONOMIALS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_golden.py",
                        "tests/test_gpu_expression.py", "-k", "(hyperplonk or sum_check or zero_check) and not runtime_compiled"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 15, r.stdout[-500:]


@pytest.mark.heavy(est=45)
def test_runtime_compiled_round_kernel_matches_golden():
    """Large general-expression sum-checks run the register program as straight-line code compiled at run time
    (csrc/jit.cpp; by default from 2^16 rows).  LH_EXPR_JIT_MIN_VARS=1 selects it for every size: the golden and
    oracle parity tests of the HyperPlonk path are repeated in a fresh process with it.  (`-k` must not select this
    test or the fallback test above again.)"""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LH_EXPR_JIT_MIN_VARS="1", LH_HP_DEBUG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-s", "-m", "gpu", "tests/test_gpu_golden.py",
                        "tests/test_gpu_expression.py", "tests/test_gpu_hyperplonk.py",
                        "-k", "(hyperplonk or sum_check or zero_check) and not fallback and not runtime_compiled"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 15, r.stdout[-500:]
    # the compiled form really ran (made here, or by an earlier test of this session and found in the disk cache)
    assert re.search(r"\[expr\] (compiled|loaded from the disk cache) a", r.stderr + r.stdout)


@pytest.mark.heavy(est=40)
@pytest.mark.parametrize("env", [{"LH_EXPR_EF_MIN_VARS": "2"}, {"LH_EXPR_EF_MIN_VARS": "2", "LH_EXPR_JIT_MIN_VARS": "2"}])
def test_eq_factored_expression_rounds_match_the_oracle(env):
    """The zero-check shape  linear part + kappa eq(y, .) C  (preprocessor.rs:43-57) runs its rounds factored from 2^14 rows on
    (csrc/expr.cpp: C's program times one eq-level entry per pair, the linear part's pair sums beside it, one evaluation
    point fewer; a claim that is not the true sum - most of this file's random tables - keeps the factored rounds and pays
    the point).  Here every test of this file and the golden vectors run again in a child process with the threshold at
    2^2 rows: interpreted, then runtime-compiled - byte for byte the oracle's messages either way."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_expression.py", "tests/test_gpu_golden.py",
                          "tests/test_gpu_hyperplonk.py", "-m", "gpu", "-x", "-q",
                          "-k", "(lagrange or rotation or zero_check or golden or hyperplonk) and not runtime_compiled and not fallback "
                                "and not eq_factored and not sharded"],
                         cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
