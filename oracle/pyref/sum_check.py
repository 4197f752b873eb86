"""ClassicSumCheck with the Evaluations and Coefficients provers.  TEST INFRASTRUCTURE ONLY.

Follows reference plonkish_backend/src/piop/sum_check/classic.rs:25-263,
classic/eval.rs:22-131 and classic/coeff.rs:16-203.  Variable 0 (the LSB of the table index)
is bound first (classic.rs:127-137 -> multilinear.rs:609-616); the returned point is
[r_0 .. r_{n-1}] and the returned evals are table[0] of every bound poly (classic.rs:143-149).
"""
from .field import R_MOD as P, fr_inv
from . import expression as ex
from .poly import eq_xy, fix_var


class SumCheckError(Exception):
    pass


class VirtualPolynomial:
    """piop/sum_check.rs:16-37"""

    def __init__(self, expression, polys, challenges, ys):
        self.expression = expression
        self.polys = [list(p) for p in polys]
        self.challenges = list(challenges)
        self.ys = [list(y) for y in ys]


# ---------------------------------------------------------------- round messages
def _lagrange_at(evals, x):
    """Value at x of the degree-(len-1) polynomial through (i, evals[i]); equals
    `barycentric_interpolate` (arithmetic.rs:108-136) for x outside {0..d}."""
    d = len(evals) - 1
    x %= P
    if x <= d:
        return evals[x] % P
    total = 0
    for j in range(d + 1):
        num, den = 1, 1
        for i in range(d + 1):
            if i != j:
                num = num * (x - i) % P
                den = den * (j - i) % P
        total = (total + evals[j] * num % P * fr_inv(den)) % P
    return total


class Evaluations:
    """classic/eval.rs:22-57: message [p(0) .. p(d)]"""

    @staticmethod
    def sum(msg):
        return (msg[0] + msg[1]) % P

    @staticmethod
    def evaluate(msg, challenge):
        return _lagrange_at(msg, challenge)


class Coefficients:
    """classic/coeff.rs:16-39: message [c_0 .. c_d]"""

    @staticmethod
    def sum(msg):
        return (2 * msg[0] + sum(msg[1:])) % P

    @staticmethod
    def evaluate(msg, challenge):
        acc = 0
        for c in reversed(msg):
            acc = (acc * challenge + c) % P
        return acc


# ---------------------------------------------------------------- prover state
class ProverState:
    """classic.rs:25-150.  Rotated queries, `Identity` and `Lagrange(i)` are materialised as ordinary
    evaluation tables (rotated[b] = poly[bh.rotate(b, rot)], id[b] = b, l_i = one-hot at bh[i]): the
    reference reads them through index maps in round 0 (eval.rs:217-266), materialises the rotated
    copies when it binds round 0 (classic.rs:104-126) and tracks identity / Lagrange values in closed
    form (classic.rs:92-101) -- all three are the multilinear extensions of exactly these tables, so
    every round message is the same field element."""

    def __init__(self, num_vars, sum_, vp):
        from .bh import BooleanHypercube
        assert num_vars > 0
        self.num_vars = num_vars
        self.expression = vp.expression
        self.degree = ex.degree(vp.expression)
        self.sum = sum_ % P
        self.eq_xys = [eq_xy(y) for y in vp.ys]
        self.polys = [list(p) for p in vp.polys]
        self.challenges = vp.challenges
        self.round = 0
        for t in self.eq_xys + self.polys:
            assert len(t) == 1 << num_vars
        bh = BooleanHypercube(num_vars)
        order = bh.iter()
        assert all(abs(r) <= num_vars for r in ex.used_rotation(vp.expression))  # classic.rs:42
        self.rotated = {}
        for (idx, rot) in ex.used_query(vp.expression):
            if rot != 0:
                self.rotated[(idx, rot)] = [self.polys[idx][bh.rotate(b, rot)] for b in range(1 << num_vars)]
        self.identity = list(range(1 << num_vars)) if any(
            isinstance(l, ex.Identity) for l in ex.leaves(vp.expression)) else None
        self.lagranges = {}
        for i in ex.used_lagrange(vp.expression):
            t = [0] * (1 << num_vars)
            t[order[i % (1 << num_vars)]] = 1
            self.lagranges[i] = t

    def size(self):
        return 1 << (self.num_vars - self.round - 1)

    def next_round(self, sum_, challenge):
        self.sum = sum_
        self.eq_xys = [fix_var(t, challenge) for t in self.eq_xys]
        self.polys = [fix_var(t, challenge) for t in self.polys]
        self.rotated = {k: fix_var(t, challenge) for k, t in self.rotated.items()}
        self.lagranges = {k: fix_var(t, challenge) for k, t in self.lagranges.items()}
        if self.identity is not None:
            self.identity = fix_var(self.identity, challenge)
        self.round += 1

    def into_evals(self):
        assert self.round == self.num_vars
        return [p[0] for p in self.polys]

    def pair_values(self, b, which):
        """values of every leaf at table index 2b + which"""
        i = 2 * b + which
        q = {(idx, 0): t[i] for idx, t in enumerate(self.polys)}
        q.update({k: t[i] for k, t in self.rotated.items()})
        return ([t[i] for t in self.eq_xys], q, self.identity[i] if self.identity is not None else 0,
                {k: t[i] for k, t in self.lagranges.items()})


class EvaluationsProver:
    """classic/eval.rs:68-131: evals[X] = sum_b expr(tables at (X, b)) for X = 1..d,
    evals[0] = claim - evals[1] (eval.rs:129)."""

    message = Evaluations

    def __init__(self, state):
        assert state.degree >= 2  # eval.rs:316 debug_assert
        self.state0 = state

    def prove_round(self, state):
        d = state.degree
        evals = [0] * (d + 1)
        for b in range(state.size()):
            eq0, q0, id0, l0 = state.pair_values(b, 0)
            eq1, q1, id1, l1 = state.pair_values(b, 1)
            for X in range(1, d + 1):
                lin = lambda v0, v1: (v1 + (X - 1) * (v1 - v0)) % P
                eqv = [lin(a, c) for a, c in zip(eq0, eq1)]
                qv = {k: lin(q0[k], q1[k]) for k in q0}
                lv = {k: lin(l0[k], l1[k]) for k in l0}
                val = ex.evaluate_general(state.expression, eqv, qv, state.challenges, lin(id0, id1), lv)
                evals[X] = (evals[X] + val) % P
        evals[0] = (state.sum - evals[1]) % P
        return evals


class CoefficientsProver:
    """classic/coeff.rs:62-203: flattens the expression into constant + sum of
    scalar * (lhs * rhs) and emits [c0, c1, c2] with c1 := claim - (2 c0 + c2)."""

    message = Coefficients

    def __init__(self, state):
        ch = state.challenges

        def neg(t):
            c, prods = t
            return ((-c) % P, [((-s) % P, ps) for s, ps in prods])

        def add(l, r):
            return ((l[0] + r[0]) % P, l[1] + r[1])

        def mul(l, r):
            (lc, lp), (rc, rp) = l, r
            out = []
            for c, prods in ((lc, rp), (rc, lp)):
                if c != 0:
                    out += [(c * s % P, list(ps)) for s, ps in prods]
            for ls, lps in lp:
                for rs, rps in rp:
                    out.append((ls * rs % P, list(lps) + list(rps)))
            return (lc * rc % P, out)

        def scaled(t, s):
            c, prods = t
            return (c * s % P, [(x * s % P, ps) for x, ps in prods])

        self.constant, self.products = ex.evaluate(
            state.expression,
            lambda c: (c, []),
            lambda i: (0, [(1, [("eq", i)])]),
            lambda i: (0, [(1, [("poly", i)])]),
            lambda i: (ch[i] % P, []),
            neg, add, mul, scaled)

    @staticmethod
    def _table(state, leaf):
        kind, i = leaf
        return state.eq_xys[i] if kind == "eq" else state.polys[i]

    def prove_round(self, state):
        assert state.degree == 2
        coeffs = [0, 0, 0]
        coeffs[0] = state.size() * self.constant % P
        for scalar, leaves in self.products:
            if len(leaves) != 2:
                raise NotImplementedError  # coeff.rs:143
            lhs, rhs = (self._table(state, l) for l in leaves)
            c0 = c2 = 0
            for b in range(state.size()):
                l0, l1, r0, r1 = lhs[2 * b], lhs[2 * b + 1], rhs[2 * b], rhs[2 * b + 1]
                c0 += l0 * r0
                c2 += (l1 - l0) * (r1 - r0)
            coeffs[0] = (coeffs[0] + scalar * c0) % P
            coeffs[2] = (coeffs[2] + scalar * c2) % P
        coeffs[1] = (state.sum - Coefficients.sum(coeffs)) % P
        return coeffs


# ---------------------------------------------------------------- protocol
def prove(prover_cls, num_vars, vp, sum_, transcript):
    """ClassicSumCheck::prove, classic.rs:208-240 -> (challenges, evals)."""
    state = ProverState(num_vars, sum_, vp)
    prover = prover_cls(state)
    challenges = []
    for _ in range(num_vars):
        msg = prover.prove_round(state)
        transcript.write_field_elements(msg)
        r = transcript.squeeze_challenge()
        challenges.append(r)
        state.next_round(prover_cls.message.evaluate(msg, r), r)
    return challenges, state.into_evals()


def verify(message_cls, num_vars, degree, sum_, transcript):
    """ClassicSumCheck::verify, classic.rs:242-263 -> (final claim, challenges)."""
    msgs, challenges = [], []
    for _ in range(num_vars):
        msgs.append(transcript.read_field_elements(degree + 1))
        challenges.append(transcript.squeeze_challenge())
    s = sum_ % P
    for rnd, (msg, r) in enumerate(zip(msgs, challenges)):
        if s != message_cls.sum(msg):
            raise SumCheckError("Consistency failure at round %d" % rnd)
        s = message_cls.evaluate(msg, r)
    return s, challenges
