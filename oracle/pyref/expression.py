"""Expression AST subset used by the sum-check provers.  TEST INFRASTRUCTURE ONLY.

Follows reference plonkish_backend/src/util/expression.rs:14-574: CommonPolynomial
{Identity, Lagrange(i), EqXY(idx)}, Polynomial(Query{poly, rotation}), Challenge, Negated, Sum,
Product, Scaled, DistributePowers; `evaluate` (:109-169), `degree` (:171-182), `used_*` (:184-243).
"""
from .field import R_MOD as P


class Expr:
    def __add__(self, o):
        return Sum(self, _wrap(o))

    def __mul__(self, o):
        if isinstance(o, int):
            return Scaled(self, o % P)
        return Product(self, o)

    def __neg__(self):
        return Negated(self)

    def __sub__(self, o):
        return Sum(self, Negated(_wrap(o)))


def _wrap(o):
    return Constant(o % P) if isinstance(o, int) else o


class Constant(Expr):
    def __init__(self, v):
        self.v = v % P


class EqXY(Expr):
    def __init__(self, idx):
        self.idx = idx


class Identity(Expr):
    """CommonPolynomial::Identity"""


class Lagrange(Expr):
    """CommonPolynomial::Lagrange(i)"""

    def __init__(self, i):
        self.i = i


class Poly(Expr):
    """Expression::Polynomial(Query::new(poly, Rotation(rotation)))"""

    def __init__(self, idx, rotation=0):
        self.idx, self.rotation = idx, rotation

    @property
    def query(self):
        return (self.idx, self.rotation)


class Challenge(Expr):
    def __init__(self, idx):
        self.idx = idx


class Negated(Expr):
    def __init__(self, a):
        self.a = a


class Sum(Expr):
    def __init__(self, a, b):
        self.a, self.b = a, b


class Product(Expr):
    def __init__(self, a, b):
        self.a, self.b = a, b


class Scaled(Expr):
    def __init__(self, a, s):
        self.a, self.s = a, s % P


class DistributePowers(Expr):
    def __init__(self, exprs, base):
        self.exprs, self.base = list(exprs), base


def distribute_powers(exprs, base):
    """expression.rs:92-105"""
    exprs = list(exprs)
    assert exprs
    return exprs[0] if len(exprs) == 1 else DistributePowers(exprs, base)


def sum_exprs(exprs):
    """`impl Sum for Expression` (expression.rs): reduce with +, empty -> Constant(0)."""
    exprs = list(exprs)
    if not exprs:
        return Constant(0)
    acc = exprs[0]
    for e in exprs[1:]:
        acc = Sum(acc, e)
    return acc


def evaluate(e, constant, common_poly, poly, challenge, negated, sum_, product, scaled):
    """expression.rs:109-169"""
    ev = lambda x: evaluate(x, constant, common_poly, poly, challenge, negated, sum_, product, scaled)
    if isinstance(e, Constant):
        return constant(e.v)
    if isinstance(e, EqXY):
        return common_poly(e.idx)
    if isinstance(e, (Identity, Lagrange)):
        return common_poly(e)
    if isinstance(e, Poly):
        return poly(e.idx) if e.rotation == 0 else poly(e)
    if isinstance(e, Challenge):
        return challenge(e.idx)
    if isinstance(e, Negated):
        return negated(ev(e.a))
    if isinstance(e, Sum):
        a = ev(e.a)
        b = ev(e.b)
        return sum_(a, b)
    if isinstance(e, Product):
        a = ev(e.a)
        b = ev(e.b)
        return product(a, b)
    if isinstance(e, Scaled):
        return scaled(ev(e.a), e.s)
    if isinstance(e, DistributePowers):
        if len(e.exprs) == 1:
            return ev(e.exprs[0])
        base = ev(e.base)
        acc = ev(e.exprs[0])
        power = base
        for sub in e.exprs[1:]:
            acc = sum_(acc, product(power, ev(sub)))
            power = product(power, base)
        return acc
    raise TypeError(e)


def degree(e):
    """expression.rs:171-182"""
    return evaluate(e, lambda _: 0, lambda _: 1, lambda _: 1, lambda _: 0, lambda a: a,
                    max, lambda a, b: a + b, lambda a, _: a)


def evaluate_fe(e, eq_vals, poly_vals, challenges):
    """Field evaluation given values for every EqXY / Poly / Challenge leaf."""
    return evaluate(
        e,
        lambda c: c,
        lambda i: eq_vals[i],
        lambda i: poly_vals[i],
        lambda i: challenges[i],
        lambda a: (-a) % P,
        lambda a, b: (a + b) % P,
        lambda a, b: a * b % P,
        lambda a, s: a * s % P,
    )


# ------------------------------------------------------------------ general form (rotations, identity, lagrange)
def _walk(e, out):
    if isinstance(e, (Negated,)):
        _walk(e.a, out)
    elif isinstance(e, (Sum, Product)):
        _walk(e.a, out)
        _walk(e.b, out)
    elif isinstance(e, Scaled):
        _walk(e.a, out)
    elif isinstance(e, DistributePowers):
        for sub in e.exprs:
            _walk(sub, out)
        _walk(e.base, out)
    else:
        out.append(e)


def leaves(e):
    out = []
    _walk(e, out)
    return out


def used_query(e):
    """expression.rs:194-196: BTreeSet<Query>, ordered by (poly, rotation)"""
    return sorted({l.query for l in leaves(e) if isinstance(l, Poly)})


def used_lagrange(e):
    return sorted({l.i for l in leaves(e) if isinstance(l, Lagrange)})


def used_rotation(e):
    return sorted({l.rotation for l in leaves(e) if isinstance(l, Poly)})


def evaluate_general(e, eq_vals, query_vals, challenges, identity, lagranges):
    """Field value given eq_xy evals, a {(poly, rotation): value} map, the identity value and a
    {i: value} map of Lagrange values (piop/sum_check.rs:60-98 `evaluate`)."""
    def common(x):
        if isinstance(x, Identity):
            return identity % P
        if isinstance(x, Lagrange):
            return lagranges[x.i] % P
        return eq_vals[x]
    return evaluate(
        e, lambda c: c, common,
        lambda q: query_vals[q.query] if isinstance(q, Poly) else query_vals[(q, 0)],
        lambda i: challenges[i], lambda a: (-a) % P, lambda a, b: (a + b) % P,
        lambda a, b: a * b % P, lambda a, s: a * s % P)
