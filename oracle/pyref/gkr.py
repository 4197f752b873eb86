"""Layered GKR arguments.  TEST INFRASTRUCTURE ONLY.

(1) `prove/verify_fractional_sum_check` follow reference
    plonkish_backend/src/piop/gkr/fractional_sum_check.rs:27-296 line by line in schedule.
(2) `prove/verify_grand_product` is the product-only specialisation (p == 1 dropped:
    v = l * r, layer expression eq * sum_b lambda^b l_b r_b) that the Lasso memory check uses
    (SURVEY.md Appendix C); it has NO reference code -- same layering / transcript schedule as
    (1), extended to batch trees of different depth (a shallower tree joins the top layers only).
"""
from .field import R_MOD as P
from . import expression as ex
from . import sum_check as sc
from .poly import eq_xy_eval


class GkrError(Exception):
    pass


# ------------------------------------------------------------------ (1) fractional sum-check
def _layer_bottom(p, q):
    mid = len(p) // 2  # fractional_sum_check.rs:42-47
    return [p[:mid], p[mid:], q[:mid], q[mid:]]


def _layer_up(layer):
    p_l, p_r, q_l, q_r = layer  # fractional_sum_check.rs:62-85
    n = len(p_l)
    v_p = [(p_l[i] * q_r[i] + p_r[i] * q_l[i]) % P for i in range(n)]
    v_q = [q_l[i] * q_r[i] % P for i in range(n)]
    h = n // 2
    return [v_p[:h], v_p[h:], v_q[:h], v_q[h:]]


def frac_sum_check_expression(num_batching):
    """fractional_sum_check.rs:272-281"""
    exprs = []
    for b in range(num_batching):
        p_l, p_r, q_l, q_r = (ex.Poly(4 * b + k) for k in range(4))
        exprs += [p_l * q_r + p_r * q_l, q_l * q_r]
    return ex.distribute_powers(exprs, ex.Challenge(0)) * ex.EqXY(0)


def _frac_claim(ps, qs, gamma):
    """fractional_sum_check.rs:283-288"""
    acc, power = 0, 1
    for p, q in zip(ps, qs):
        acc = (acc + p * power) % P
        power = power * gamma % P
        acc = (acc + q * power) % P
        power = power * gamma % P
    return acc


def _frac_down(evals, mu):
    """fractional_sum_check.rs:290-296"""
    ps, qs = [], []
    for i in range(0, len(evals), 4):
        p_l, p_r, q_l, q_r = evals[i:i + 4]
        ps.append((p_l + mu * (p_r - p_l)) % P)
        qs.append((q_l + mu * (q_r - q_l)) % P)
    return ps, qs


def prove_fractional_sum_check(claimed_p_0s, claimed_q_0s, ps, qs, transcript):
    """fractional_sum_check.rs:89-190 -> (p_xs, q_xs, x)"""
    B = len(claimed_p_0s)
    assert B and B == len(claimed_q_0s) == len(ps) == len(qs)
    assert all(len(t) == len(ps[0]) for t in list(ps) + list(qs))
    layers = [[_layer_bottom(p, q) for p, q in zip(ps, qs)]]
    while len(layers[-1][0][0]) > 1:
        layers.append([_layer_up(l) for l in layers[-1]])
    p_0s = [(l[0][0] * l[3][0] + l[1][0] * l[2][0]) % P for l in layers[-1]]
    q_0s = [l[2][0] * l[3][0] % P for l in layers[-1]]
    for claimed, computed in ((claimed_p_0s, p_0s), (claimed_q_0s, q_0s)):
        for c, v in zip(claimed, computed):
            if c is not None:
                transcript.common_field_element(v)
            else:
                transcript.write_field_element(v)
    expression = frac_sum_check_expression(B)
    claimed_p, claimed_q, y = p_0s, q_0s, []
    for layer in reversed(layers):
        polys = [t for l in layer for t in l]
        num_vars = len(polys[0]).bit_length() - 1
        if num_vars == 0:
            x, evals = [], [t[0] for t in polys]
        else:
            gamma = transcript.squeeze_challenge()
            claim = _frac_claim(claimed_p, claimed_q, gamma)
            vp = sc.VirtualPolynomial(expression, polys, [gamma], [y])
            x, evals = sc.prove(sc.EvaluationsProver, num_vars, vp, claim, transcript)
        transcript.write_field_elements(evals)
        mu = transcript.squeeze_challenge()
        claimed_p, claimed_q = _frac_down(evals, mu)
        y = x + [mu]
    # the reference's `sanity-check` feature (fractional_sum_check.rs:184-187): the final claims are the inputs at x
    from .poly import evaluate as _evaluate
    assert all(_evaluate(t, y) == v for t, v in zip(list(ps) + list(qs), claimed_p + claimed_q))
    return claimed_p, claimed_q, y


def verify_fractional_sum_check(num_vars, claimed_p_0s, claimed_q_0s, transcript):
    """fractional_sum_check.rs:193-270"""
    B = len(claimed_p_0s)
    roots = []
    for claimed in (claimed_p_0s, claimed_q_0s):
        row = []
        for c in claimed:
            if c is not None:
                transcript.common_field_element(c)
                row.append(c % P)
            else:
                row.append(transcript.read_field_element())
        roots.append(row)
    claimed_p, claimed_q = roots
    expression = frac_sum_check_expression(B)
    y = []
    for nv in range(num_vars):
        if nv == 0:
            evals = transcript.read_field_elements(4 * B)
            for b in range(B):
                p_l, p_r, q_l, q_r = evals[4 * b:4 * b + 4]
                if claimed_p[b] != (p_l * q_r + p_r * q_l) % P or claimed_q[b] != q_l * q_r % P:
                    raise GkrError("Unmatched between sum_check output and query evaluation")
            x = []
        else:
            gamma = transcript.squeeze_challenge()
            claim = _frac_claim(claimed_p, claimed_q, gamma)
            x_eval, x = sc.verify(sc.Evaluations, nv, ex.degree(expression), claim, transcript)
            evals = transcript.read_field_elements(4 * B)
            if x_eval != ex.evaluate_fe(expression, [eq_xy_eval(x, y)], evals, [gamma]):
                raise GkrError("Unmatched between sum_check output and query evaluation")
        mu = transcript.squeeze_challenge()
        claimed_p, claimed_q = _frac_down(evals, mu)
        y = x + [mu]
    return claimed_p, claimed_q, y


# ------------------------------------------------------------------ (2) grand product
def grand_product_expression(num_batching):
    exprs = [ex.Poly(2 * b) * ex.Poly(2 * b + 1) for b in range(num_batching)]
    return ex.distribute_powers(exprs, ex.Challenge(0)) * ex.EqXY(0)


def _gp_claim(claims, lam):
    acc, power = 0, 1
    for c in claims:
        acc = (acc + c * power) % P
        power = power * lam % P
    return acc


def _product_tree(v):
    """layers[h] = (l, r) halves of the level with 2^(h+1) nodes; layers[0] is the 0-variable
    top layer, layers[-1] the leaves split at mid (as Layer::bottom)."""
    levels = [list(v)]
    while len(levels[-1]) > 2:
        cur = levels[-1]
        mid = len(cur) // 2
        levels.append([cur[i] * cur[mid + i] % P for i in range(mid)])
    levels.reverse()
    return [(lv[:len(lv) // 2], lv[len(lv) // 2:]) for lv in levels]


def prove_grand_product(vs, transcript):
    """Batched product-tree GKR.  vs: list of leaf vectors (power-of-two lengths >= 2, any mix
    of depths).  Writes the roots, then per layer top-down: [squeeze lambda, sum-check]
    (skipped for the 0-variable layer), write (l, r) evals of every ACTIVE tree, squeeze mu.
    Returns (roots, [(claim_b, point_b)]): the MLE claim of every leaf vector."""
    B = len(vs)
    assert B and all(len(v) >= 2 and len(v) & (len(v) - 1) == 0 for v in vs)
    trees = [_product_tree(v) for v in vs]
    depth = [len(t) for t in trees]
    roots = [t[0][0][0] * t[0][1][0] % P for t in trees]
    transcript.write_field_elements(roots)
    claims, y = list(roots), []
    out = [None] * B
    for h in range(max(depth)):
        active = [b for b in range(B) if depth[b] > h]
        polys = [half for b in active for half in trees[b][h]]
        if h == 0:
            x, evals = [], [t[0] for t in polys]
        else:
            lam = transcript.squeeze_challenge()
            claim = _gp_claim([claims[b] for b in active], lam)
            vp = sc.VirtualPolynomial(grand_product_expression(len(active)), polys, [lam], [y])
            x, evals = sc.prove(sc.EvaluationsProver, h, vp, claim, transcript)
        transcript.write_field_elements(evals)
        mu = transcript.squeeze_challenge()
        y = x + [mu]
        for k, b in enumerate(active):
            l, r = evals[2 * k], evals[2 * k + 1]
            claims[b] = (l + mu * (r - l)) % P
            if depth[b] == h + 1:
                out[b] = (claims[b], list(y))
    return roots, out


def verify_grand_product(num_vars_list, transcript):
    """Returns (roots, [(claim_b, point_b)])."""
    B = len(num_vars_list)
    depth = list(num_vars_list)
    roots = transcript.read_field_elements(B)
    claims, y = list(roots), []
    out = [None] * B
    for h in range(max(depth)):
        active = [b for b in range(B) if depth[b] > h]
        if h == 0:
            evals = transcript.read_field_elements(2 * len(active))
            for k, b in enumerate(active):
                if claims[b] != evals[2 * k] * evals[2 * k + 1] % P:
                    raise GkrError("grand product: root mismatch")
            x = []
        else:
            lam = transcript.squeeze_challenge()
            claim = _gp_claim([claims[b] for b in active], lam)
            expression = grand_product_expression(len(active))
            x_eval, x = sc.verify(sc.Evaluations, h, ex.degree(expression), claim, transcript)
            evals = transcript.read_field_elements(2 * len(active))
            if x_eval != ex.evaluate_fe(expression, [eq_xy_eval(x, y)], evals, [lam]):
                raise GkrError("grand product: layer %d mismatch" % h)
        mu = transcript.squeeze_challenge()
        y = x + [mu]
        for k, b in enumerate(active):
            l, r = evals[2 * k], evals[2 * k + 1]
            claims[b] = (l + mu * (r - l)) % P
            if depth[b] == h + 1:
                out[b] = (claims[b], list(y))
    return roots, out
