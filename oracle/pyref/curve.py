"""BN254 G1 (y^2 = x^3 + 3 over Fq) in Python big-ints.  TEST INFRASTRUCTURE ONLY.

Restates what the reference takes from halo2_curves 0.3.3 `bn256::{G1Affine, G1}`
(call sites plonkish_backend/src/util/arithmetic/msm.rs:129-179,
pcs/multilinear/kzg.rs:204-207).  Points are affine tuples (x, y) or None = identity;
only the affine result of an MSM is observable, so the summation order
and window scheme here are free.
"""
from .field import Q_MOD, R_MOD, CURVE_B, fq_inv

G1_GEN = (1, 2)
P = Q_MOD


def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - CURVE_B) % P == 0


def neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % P)


# Jacobian (X, Y, Z), Z == 0 is the identity
def _jdbl(p):
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jadd_affine(p, q):
    """p Jacobian, q affine tuple (not identity)."""
    X1, Y1, Z1 = p
    if Z1 == 0:
        return (q[0], q[1], 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = q[0] * Z1Z1 % P
    S2 = q[1] * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    r = (S2 - Y1) % P
    if H == 0:
        if r == 0:
            return _jdbl(p)
        return (1, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (r * r - HHH - 2 * V) % P
    Y3 = (r * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def _jadd(p, q):
    if p[2] == 0:
        return q
    if q[2] == 0:
        return p
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    H = (U2 - U1) % P
    r = (S2 - S1) % P
    if H == 0:
        if r == 0:
            return _jdbl(p)
        return (1, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = U1 * HH % P
    X3 = (r * r - HHH - 2 * V) % P
    Y3 = (r * (V - X3) - S1 * HHH) % P
    Z3 = Z1 * Z2 * H % P
    return (X3, Y3, Z3)


def _to_affine(p):
    X, Y, Z = p
    if Z == 0:
        return None
    zi = fq_inv(Z)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return _to_affine(_jadd_affine((a[0], a[1], 1), b))


def mul(pt, k):
    k %= R_MOD
    if pt is None or k == 0:
        return None
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jdbl(acc)
        if bit == "1":
            acc = _jadd_affine(acc, pt)
    return _to_affine(acc)


class FixedBase:
    """Windowed fixed-base multiplication (role of reference msm.rs:16-81 `window_table` /
    `fixed_base_msm`; only the resulting points are observable)."""

    def __init__(self, base=G1_GEN, window=8):
        self.w = window
        self.tables = []
        cur = (base[0], base[1], 1)
        for _ in range((254 + window - 1) // window):
            row, acc = [None], (1, 1, 0)
            for _ in range((1 << window) - 1):
                acc = _jadd(acc, cur)
                row.append(acc)
            self.tables.append(row)
            for _ in range(window):
                cur = _jdbl(cur)

    def mul(self, k):
        k %= R_MOD
        acc = (1, 1, 0)
        i = 0
        while k:
            d = k & ((1 << self.w) - 1)
            if d:
                acc = _jadd(acc, self.tables[i][d])
            k >>= self.w
            i += 1
        return _to_affine(acc)


def msm(scalars, bases):
    """variable_base_msm (reference msm.rs:84-181): returns the affine sum, None = identity.
    Bucket method with 8-bit windows; the partition/window choice is unobservable."""
    assert len(scalars) == len(bases)
    w = 8
    scalars = [s % R_MOD for s in scalars]
    nwin = (max([s.bit_length() for s in scalars] + [1]) + w - 1) // w
    total = (1, 1, 0)
    for win in range(nwin - 1, -1, -1):
        for _ in range(w):
            total = _jdbl(total)
        buckets = [(1, 1, 0)] * ((1 << w) - 1)
        for s, b in zip(scalars, bases):
            d = (s >> (w * win)) & ((1 << w) - 1)
            if d and b is not None:
                buckets[d - 1] = _jadd_affine(buckets[d - 1], b)
        run = (1, 1, 0)
        acc = (1, 1, 0)
        for bkt in reversed(buckets):
            run = _jadd(run, bkt)
            acc = _jadd(acc, run)
        total = _jadd(total, acc)
    return _to_affine(total)
