"""Host mirror of plonkish_backend::util::expression (util/expression.rs:14-574): the `Expression` AST
with the reference's operator semantics, and its flattening into the `lh_expr` node array of the C-ABI.
Pure host bookkeeping (a circuit has a few hundred nodes); all evaluation happens on the GPU.
"""
import ctypes as C

from . import _ffi

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
_MONT = 1 << 256


def _fr_bytes(x):
    return (x % R_MOD * _MONT % R_MOD).to_bytes(32, "little")


class Expression:
    """Operators as impl_expression_ops! (expression.rs:488-519): `*` Expression -> Product, `*` int -> Scaled,
    `+` -> Sum, `-` -> Sum(lhs, Negated(rhs)), unary `-` -> Negated."""

    def __add__(self, o):
        return Sum(self, _wrap(o))

    def __sub__(self, o):
        return Sum(self, Negated(_wrap(o)))

    def __mul__(self, o):
        return Scaled(self, o % R_MOD) if isinstance(o, int) else Product(self, o)

    def __neg__(self):
        return Negated(self)

    # ---- expression.rs:171-243
    def degree(self):
        return _fold(self, lambda e: 0 if isinstance(e, (Constant, Challenge)) else 1, lambda a: a, max,
                     lambda a, b: a + b, lambda a: a)

    def leaves(self):
        out = []
        _walk(self, out)
        return out

    def used_query(self):
        return sorted({(l.poly, l.rotation) for l in self.leaves() if isinstance(l, Polynomial)})

    def used_lagrange(self):
        return sorted({l.i for l in self.leaves() if isinstance(l, Lagrange)})

    # ---- C-ABI
    def to_c(self):
        """-> (lh_expr, keepalive): nodes in topological order, DistributePowers lowered as
        Expression::evaluate does (expression.rs:155-167)."""
        nodes = []

        def emit(op, a=0, b=0, scalar=0):
            nodes.append((op, a, b, scalar))
            return len(nodes) - 1

        def go(e):
            if isinstance(e, Constant):
                return emit(0, scalar=e.value)
            if isinstance(e, Identity):
                return emit(1)
            if isinstance(e, Lagrange):
                return emit(2, e.i)
            if isinstance(e, EqXY):
                return emit(3, e.idx)
            if isinstance(e, Polynomial):
                return emit(4, e.poly, e.rotation)
            if isinstance(e, Challenge):
                return emit(5, e.idx)
            if isinstance(e, Negated):
                return emit(6, go(e.a))
            if isinstance(e, Sum):
                a = go(e.a)
                return emit(7, a, go(e.b))
            if isinstance(e, Product):
                a = go(e.a)
                return emit(8, a, go(e.b))
            if isinstance(e, Scaled):
                return emit(9, go(e.a), scalar=e.scalar)
            if isinstance(e, DistributePowers):
                if len(e.exprs) == 1:
                    return go(e.exprs[0])
                base = go(e.base)
                acc = go(e.exprs[0])
                power = base
                for k, sub in enumerate(e.exprs[1:]):
                    if k:
                        power = emit(8, power, base)
                    acc = emit(7, acc, emit(8, power, go(sub)))
                return acc  # the root must stay the LAST node
            raise TypeError(e)

        go(self)
        arr = (_ffi.lh_expr_node * len(nodes))()
        for k, (op, a, b, scalar) in enumerate(nodes):
            arr[k].op, arr[k].a, arr[k].b = op, a, b
            C.memmove(C.byref(arr[k].scalar), _fr_bytes(scalar), 32)
        ce = _ffi.lh_expr()
        ce.nodes = C.cast(arr, C.POINTER(_ffi.lh_expr_node))
        ce.num_nodes = len(nodes)
        return ce, arr


def _wrap(o):
    return Constant(o) if isinstance(o, int) else o


class Constant(Expression):
    def __init__(self, value):
        self.value = value % R_MOD


class Identity(Expression):
    pass


class Lagrange(Expression):
    def __init__(self, i):
        self.i = i


class EqXY(Expression):
    def __init__(self, idx):
        self.idx = idx


class Polynomial(Expression):
    """Expression::Polynomial(Query::new(poly, Rotation(rotation)))"""

    def __init__(self, poly, rotation=0):
        self.poly, self.rotation = poly, rotation


class Challenge(Expression):
    def __init__(self, idx):
        self.idx = idx


class Negated(Expression):
    def __init__(self, a):
        self.a = a


class Sum(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b


class Product(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b


class Scaled(Expression):
    def __init__(self, a, scalar):
        self.a, self.scalar = a, scalar % R_MOD


class DistributePowers(Expression):
    def __init__(self, exprs, base):
        self.exprs, self.base = list(exprs), base


def distribute_powers(exprs, base):
    """expression.rs:92-105"""
    exprs = list(exprs)
    assert exprs
    return exprs[0] if len(exprs) == 1 else DistributePowers(exprs, base)


def sum_exprs(exprs):
    """impl Sum (expression.rs:537-542)"""
    exprs = list(exprs)
    if not exprs:
        return Constant(0)
    acc = exprs[0]
    for e in exprs[1:]:
        acc = acc + e
    return acc


def product_exprs(exprs):
    """impl Product (expression.rs:550-555)"""
    exprs = list(exprs)
    if not exprs:
        return Constant(1)
    acc = exprs[0]
    for e in exprs[1:]:
        acc = acc * e
    return acc


def _walk(e, out):
    if isinstance(e, Negated):
        _walk(e.a, out)
    elif isinstance(e, (Sum, Product)):
        _walk(e.a, out)
        _walk(e.b, out)
    elif isinstance(e, Scaled):
        _walk(e.a, out)
    elif isinstance(e, DistributePowers):
        for s in e.exprs:
            _walk(s, out)
        _walk(e.base, out)
    else:
        out.append(e)


def _fold(e, leaf, neg, add, mul, scaled):
    f = lambda x: _fold(x, leaf, neg, add, mul, scaled)
    if isinstance(e, Negated):
        return neg(f(e.a))
    if isinstance(e, Sum):
        return add(f(e.a), f(e.b))
    if isinstance(e, Product):
        return mul(f(e.a), f(e.b))
    if isinstance(e, Scaled):
        return scaled(f(e.a))
    if isinstance(e, DistributePowers):
        if len(e.exprs) == 1:
            return f(e.exprs[0])
        base = f(e.base)
        acc, power = f(e.exprs[0]), base
        for s in e.exprs[1:]:
            acc = add(acc, mul(power, f(s)))
            power = mul(power, base)
        return acc
    return leaf(e)
