"""Multilinear KZG (PST) commit / open / batch_open.  TEST INFRASTRUCTURE ONLY.

Follows reference plonkish_backend/src/pcs/multilinear/kzg.rs:166-361 and
pcs/multilinear.rs:72-276.  Differences from the reference, both outside the prover's
observable output:
  * `setup` takes the trapdoor `s` explicitly (reference draws it from an RNG, kzg.rs:169-171);
  * `verify` checks the opening identity in G1 with the trapdoor,
        C - v*G == sum_i (s_i - x_i) * pi_i,
    which is the discrete-log image of the pairing check at kzg.rs:341-360.
"""
from .field import R_MOD as P
from . import curve
from . import expression as ex
from . import sum_check as sc
from .poly import eq_xy, eq_xy_eval


class PcsError(Exception):
    pass


class Params:
    def __init__(self, ss, eqs):
        self.ss = list(ss)    # trapdoor (test only)
        self.eqs = eqs        # eqs[k][b] = eq_k(b; s_0..s_{k-1}) * G, k = 0..n

    @property
    def num_vars(self):
        return len(self.eqs) - 1

    def trim(self, num_vars):
        """kzg.rs:230-250"""
        if self.num_vars < num_vars:
            raise PcsError("Too many variates to trim")
        return Params(self.ss[:num_vars], self.eqs[:num_vars + 1])


def setup(ss):
    """kzg.rs:166-228: eqs[k] expands eqs[k-1] with s_{k-1} as the NEW TOP bit (kzg.rs:178-194:
    hi = s_i * last, lo = last - hi)."""
    fb = curve.FixedBase(curve.G1_GEN)
    scal = [[1]]
    for s_i in ss:
        last = scal[-1]
        hi = [s_i * e % P for e in last]
        lo = [(e - h) % P for e, h in zip(last, hi)]
        scal.append(lo + hi)
    return Params(ss, [[fb.mul(e) for e in lvl] for lvl in scal])


def commit(pp, evals):
    """kzg.rs:252-257"""
    nv = len(evals).bit_length() - 1
    if nv > pp.num_vars:
        raise PcsError("Too many variates of poly to commit")
    return curve.msm(evals, pp.eqs[nv])


def batch_commit_and_write(pp, polys, transcript):
    """pcs.rs:62-75"""
    comms = [commit(pp, p) for p in polys]
    for c in comms:
        transcript.write_commitment(c)
    return comms


def quotients(evals, point):
    """pcs/multilinear.rs:72-107: top variable first; returns ([q_0..q_{n-1}], remainder)."""
    n = len(point)
    assert len(evals) == 1 << n
    rem = list(evals)
    qs = [None] * n
    for i in range(n - 1, -1, -1):
        lo, hi = rem[:1 << i], rem[1 << i:1 << (i + 1)]
        qs[i] = [(h - l) % P for h, l in zip(hi, lo)]
        rem = [(l + (h - l) * point[i]) % P for h, l in zip(hi, lo)]
    return qs, rem[0]


def open_(pp, evals, point, transcript):
    """kzg.rs:276-302: writes pi_0..pi_{n-1}, pi_i = commit(q_i) with eqs[i]."""
    qs, rem = quotients(evals, point)
    comms = [curve.msm(q, pp.eqs[i]) for i, q in enumerate(qs)]
    # the reference's `sanity-check` feature (kzg.rs:286-297): the remainder of the quotients is the evaluation
    from .poly import evaluate as _evaluate
    assert rem == _evaluate(evals, point)
    transcript.write_commitments(comms)  # identity -> TranscriptError, as transcript.rs:216-219
    return rem


def verify(vp, comm, point, eval_, transcript):
    """kzg.rs:330-361 (trapdoor form, see module docstring)."""
    n = len(point)
    pis = transcript.read_commitments(n)
    lhs = curve.add(comm, curve.neg(curve.mul(curve.G1_GEN, eval_)))
    rhs = None
    for s_i, x_i, pi in zip(vp.ss[:n], point, pis):
        rhs = curve.add(rhs, curve.mul(pi, (s_i - x_i) % P))
    if lhs != rhs:
        raise PcsError("Invalid multilinear KZG open")


class Evaluation:
    def __init__(self, poly, point, value):
        self.poly, self.point, self.value = poly, point, value % P


def _merged(polys, points, evals, eq_xt):
    """pcs/multilinear.rs:155-170 as field identities: scalar_j * merged_j =
    sum_{i: evals[i].point == j} eq_xt[i] * polys[evals[i].poly] (the lazy first-scalar there
    changes no field value)."""
    merged = [None] * len(points)
    for ev, w in zip(evals, eq_xt):
        src = polys[ev.poly]
        if merged[ev.point] is None:
            merged[ev.point] = [w * v % P for v in src]
        else:
            m = merged[ev.point]
            merged[ev.point] = [(a + w * v) % P for a, v in zip(m, src)]
    return merged


def batch_open(pp, num_vars, polys, points, evals, transcript):
    """pcs/multilinear.rs:134-235"""
    for pt in points:
        if len(pt) != num_vars:
            raise PcsError("Invalid point")
    ell = (len(evals) - 1).bit_length() if len(evals) > 1 else 0  # next_power_of_two().ilog2()
    t = transcript.squeeze_challenges(ell)
    eq_xt = eq_xy(t) if ell else []  # eq_xy(&[]) is zero() (multilinear.rs:92-94)
    if not eq_xt:
        raise PcsError("batch_open needs >= 2 evaluations (eq_xy of an empty point is empty)")
    merged = _merged(polys, points, evals, eq_xt)
    expression = ex.sum_exprs(ex.EqXY(j) * ex.Poly(j) * 1 for j in range(len(points)))
    vp = sc.VirtualPolynomial(expression, merged, [], points)
    tilde_gs_sum = sum(ev.value * w for ev, w in zip(evals, eq_xt)) % P
    challenges, _ = sc.prove(sc.CoefficientsProver, num_vars, vp, tilde_gs_sum, transcript)
    g_prime = [0] * (1 << num_vars)
    for m, pt in zip(merged, points):
        w = eq_xy_eval(challenges, pt)
        g_prime = [(a + w * v) % P for a, v in zip(g_prime, m)]
    return open_(pp, g_prime, challenges, transcript)


def batch_verify(vp, num_vars, comms, points, evals, transcript):
    """pcs/multilinear.rs:237-276"""
    ell = (len(evals) - 1).bit_length() if len(evals) > 1 else 0
    t = transcript.squeeze_challenges(ell)
    eq_xt = eq_xy(t)
    tilde_gs_sum = sum(ev.value * w for ev, w in zip(evals, eq_xt)) % P
    g_prime_eval, challenges = sc.verify(sc.Coefficients, num_vars, 2, tilde_gs_sum, transcript)
    eq_evals = [eq_xy_eval(challenges, pt) for pt in points]
    scalars = [eq_evals[ev.point] * w % P for ev, w in zip(evals, eq_xt)]
    g_prime_comm = curve.msm(scalars, [comms[ev.poly] for ev in evals])
    verify(vp, g_prime_comm, challenges, g_prime_eval, transcript)
