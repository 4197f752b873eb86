"""Zeromorph over univariate KZG.  TEST INFRASTRUCTURE ONLY.

Restates reference plonkish_backend/src/pcs/univariate/kzg.rs:23-36,175-299,366-378 (setup / trim / commit_coeffs /
open / verify of UnivariateKzg) and pcs/multilinear/zeromorph.rs:86-322 (trim, commit, open, verify,
eval_and_quotient_scalars), with batch_open / batch_verify through the same additive reduction as multilinear KZG
(pcs/multilinear.rs:134-276).  Setup takes the trapdoor `s` explicitly; the verifier checks the pairing equation
e(c, -[s^offset]_2) e(pi, [s]_2 - x [1]_2) = 1 in its trapdoor form  s^offset * c == (s - x) * pi  (the product's host
verifier uses the real pairing; tests/test_verifier.py ties the two together).
"""
from .field import R_MOD as P
from . import curve, kzg, sum_check as sc, expression as ex
from .field import batch_invert
from .poly import eq_xy, eq_xy_eval


class Params:
    """UnivariateKzgParam (univariate/kzg.rs:38-66): powers_of_s_g1[i] = s^i G1 (and the trapdoor, test only)"""

    def __init__(self, s, powers_g1):
        self.s, self.powers_g1 = s % P, powers_g1


def setup(s, poly_size):
    """univariate/kzg.rs:175-218"""
    fb = curve.FixedBase(curve.G1_GEN)
    return Params(s, [fb.mul(pow(s, i, P)) for i in range(poly_size)])


class ProverParam:
    """ZeromorphKzgProverParam (zeromorph.rs:29-40): commit_pp = powers[..poly_size], open_pp = powers[offset..]"""

    def __init__(self, commit_powers, open_powers):
        self.commit_powers, self.open_powers = commit_powers, open_powers

    @property
    def degree(self):
        return len(self.commit_powers) - 1


class VerifierParam:
    def __init__(self, s, offset):
        self.s, self.offset = s, offset  # stands for (g1, g2, [s]_2, [s^offset]_2)


def trim(param, poly_size):
    """zeromorph.rs:90-108, univariate/kzg.rs:220-240"""
    if len(param.powers_g1) < poly_size:
        raise kzg.PcsError("Too large poly_size to trim to")
    offset = len(param.powers_g1) - poly_size
    return ProverParam(param.powers_g1[:poly_size], param.powers_g1[offset:]), VerifierParam(param.s, offset)


def commit_coeffs(powers, coeffs):
    """univariate/kzg.rs:24-31"""
    return curve.msm(coeffs, powers[:len(coeffs)])


def commit(pp, evals):
    """zeromorph.rs:110-120: the evaluation table is committed as a coefficient vector"""
    if pp.degree + 1 < len(evals):
        raise kzg.PcsError("Too large degree of poly to commit")
    return commit_coeffs(pp.commit_powers, evals)


def batch_commit_and_write(pp, polys, transcript):
    comms = [commit(pp, p) for p in polys]
    transcript.write_commitments(comms)
    return comms


def eval_and_quotient_scalars(y, x, z, u):
    """zeromorph.rs:258-296"""
    n = len(u)
    squares = [x % P]
    for _ in range(n):
        squares.append(squares[-1] * squares[-1] % P)
    offsets, state = [], 1
    for sq in reversed(squares[:-1]):   # .rev().skip(1)
        state = state * sq % P
        offsets.append(state)
    offsets.reverse()
    v_numer = (squares[n] - 1) % P
    vs = [v_numer * d % P for d in batch_invert([(sq - 1) % P for sq in squares])]
    q_scalars, power_of_y = [], 1
    for i in range(n):
        q_scalars.append((-(power_of_y * offsets[i] + z * (squares[i] * vs[i + 1] - u[i] * vs[i]))) % P)
        power_of_y = power_of_y * y % P
    return (-vs[0] * z) % P, q_scalars


def _div_by_linear(f, x):
    """(f - f(x)) / (X - x): quotient coefficients (univariate.rs:144-166 with divisor X - x)"""
    q = [0] * (len(f) - 1)
    carry = 0
    for i in range(len(f) - 1, 0, -1):
        carry = (f[i] + carry * x) % P
        q[i - 1] = carry
    return q


def open_(pp, evals, point, eval_, transcript):
    """zeromorph.rs:134-199; `eval_` only shifts the constant term of f, which the quotient does not see"""
    n = len(point)
    if pp.degree + 1 < len(evals):
        raise kzg.PcsError("Too large degree of poly to open")
    qs, _ = kzg.quotients(evals, point)
    transcript.write_commitments([commit_coeffs(pp.commit_powers, q) for q in qs])
    y = transcript.squeeze_challenge()
    q_hat = [0] * (1 << n)
    power_of_y = 1
    for idx, q in enumerate(qs):
        off = (1 << n) - (1 << idx)
        for j, v in enumerate(q):
            q_hat[off + j] = (q_hat[off + j] + power_of_y * v) % P
        power_of_y = power_of_y * y % P
    transcript.write_commitment(commit_coeffs(pp.commit_powers, q_hat))
    x = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    eval_scalar, q_scalars = eval_and_quotient_scalars(y, x, z, point)
    f = [(z * a + b) % P for a, b in zip(evals, q_hat)]
    f[0] = (f[0] + eval_scalar * eval_) % P
    for q, s in zip(qs, q_scalars):
        for j, v in enumerate(q):
            f[j] = (f[j] + s * v) % P
    while len(f) > 1 and f[-1] == 0:    # UnivariatePolynomial::new truncates leading zeros
        f.pop()
    quotient = _div_by_linear(f, x)
    transcript.write_commitment(commit_coeffs(pp.open_powers, quotient))


def verify(vp, comm, point, eval_, transcript):
    """zeromorph.rs:215-247"""
    n = len(point)
    q_comms = transcript.read_commitments(n)
    y = transcript.squeeze_challenge()
    q_hat_comm = transcript.read_commitment()
    x = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    eval_scalar, q_scalars = eval_and_quotient_scalars(y, x, z, point)
    scalars = [1, z, eval_scalar * eval_ % P] + q_scalars
    bases = [q_hat_comm, comm, curve.G1_GEN] + q_comms
    c = None
    for s_, b in zip(scalars, bases):
        c = curve.add(c, curve.mul(b, s_))
    pi = transcript.read_commitment()
    if curve.mul(c, pow(vp.s, vp.offset, P)) != curve.mul(pi, (vp.s - x) % P):
        raise kzg.PcsError("Invalid Zeromorph KZG open")


def batch_open(pp, num_vars, polys, points, evals, transcript):
    """additive::batch_open (pcs/multilinear.rs:134-235) with Pcs = Zeromorph; g_prime_eval is passed as zero
    (multilinear.rs:224-226, no sanity-check feature)"""
    ell = (len(evals) - 1).bit_length() if len(evals) > 1 else 0
    t = transcript.squeeze_challenges(ell)
    eq_xt = eq_xy(t) if ell else []
    if not eq_xt:
        raise kzg.PcsError("batch_open needs >= 2 evaluations")
    merged = kzg._merged(polys, points, evals, eq_xt)
    expression = ex.sum_exprs(ex.EqXY(j) * ex.Poly(j) * 1 for j in range(len(points)))
    vp = sc.VirtualPolynomial(expression, merged, [], points)
    tilde_gs_sum = sum(ev.value * w for ev, w in zip(evals, eq_xt)) % P
    challenges, _ = sc.prove(sc.CoefficientsProver, num_vars, vp, tilde_gs_sum, transcript)
    g_prime = [0] * (1 << num_vars)
    for m, pt in zip(merged, points):
        w = eq_xy_eval(challenges, pt)
        g_prime = [(a + w * v) % P for a, v in zip(g_prime, m)]
    open_(pp, g_prime, challenges, 0, transcript)


def batch_verify(vp, num_vars, comms, points, evals, transcript):
    """additive::batch_verify (pcs/multilinear.rs:237-276) with Pcs = Zeromorph"""
    ell = (len(evals) - 1).bit_length() if len(evals) > 1 else 0
    t = transcript.squeeze_challenges(ell)
    eq_xt = eq_xy(t)
    tilde_gs_sum = sum(ev.value * w for ev, w in zip(evals, eq_xt)) % P
    g_prime_eval, challenges = sc.verify(sc.Coefficients, num_vars, 2, tilde_gs_sum, transcript)
    eq_evals = [eq_xy_eval(challenges, pt) for pt in points]
    scalars = [eq_evals[ev.point] * w % P for ev, w in zip(evals, eq_xt)]
    g_prime_comm = curve.msm(scalars, [comms[ev.poly] for ev in evals])
    verify(vp, g_prime_comm, challenges, g_prime_eval, transcript)
