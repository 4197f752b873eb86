"""BN254 G2 and the optimal ate pairing in Python big-ints.  TEST INFRASTRUCTURE ONLY.

Restates what the reference takes from halo2_curves 0.3.3 `bn256::{G2Affine, Gt, multi_miller_loop}`
through `MultiMillerLoop::pairings_product_is_identity` (plonkish_backend/src/util/arithmetic.rs:24-33,
call site pcs/multilinear/kzg.rs:330-361).  Only "is the product of pairings the identity" is observable,
so the Miller-loop shape here is free: Fq12 is the flat extension Fq[w]/(w^12 - 18 w^6 + 82) (w^6 = 9 + u),
lines are affine and evaluated on untwisted points, and the final exponentiation is one big power.
The C++ product verifier uses a 2-3-2 tower instead; tests compare the two through `to_tower`.
"""
from .field import Q_MOD, R_MOD, fq_inv

P = Q_MOD
ATE_LOOP = 29793968203157093288  # 6x + 2, x = 4965661367192848881
FINAL_EXP = (P ** 12 - 1) // R_MOD


# ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2 + 1), tuples (c0, c1)
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    n = fq_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * n % P, (-a[1]) * n % P)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


XI = (9, 1)
B2 = f2_mul((3, 0), f2_inv(XI))  # twist: y^2 = x^3 + 3/(9+u)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


# ------------------------------------------------------------------ G2 affine ((x0,x1),(y0,y1)) or None
def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == (0, 0)


def g2_neg(pt):
    return None if pt is None else (pt[0], f2_neg(pt[1]))


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if a[1] != b[1] or a[1] == (0, 0):
            return None
        lam = f2_mul(f2_scale(f2_mul(a[0], a[0]), 3), f2_inv(f2_scale(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))


def g2_mul(pt, k):
    k %= R_MOD
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, pt)
        pt = g2_add(pt, pt)
        k >>= 1
    return acc


# ------------------------------------------------------------------ Fq12 flat: 12 coefficients, w^12 = 18 w^6 - 82
def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [c % P for c in t[:12]]


F12_ONE = [1] + [0] * 11


def f12_pow(a, e):
    out = F12_ONE
    for bit in bin(e)[2:]:
        out = f12_mul(out, out)
        if bit == "1":
            out = f12_mul(out, a)
    return out


def f12_inv(a):
    """a^(q^12 - 2) would be slow; invert through the norm chain instead: solve a * x = 1 by linear algebra."""
    # 12x12 system over Fq: columns = a * w^j
    cols = []
    for j in range(12):
        e = [0] * 12
        e[j] = 1
        cols.append(f12_mul(a, e))
    m = [[cols[j][i] for j in range(12)] + [1 if i == 0 else 0] for i in range(12)]
    for c in range(12):
        piv = next(r for r in range(c, 12) if m[r][c])
        m[c], m[piv] = m[piv], m[c]
        inv = fq_inv(m[c][c])
        m[c] = [v * inv % P for v in m[c]]
        for r in range(12):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(v - f * w) % P for v, w in zip(m[r], m[c])]
    return [m[i][12] for i in range(12)]


def _embed_fq2(c):
    """Fq2 element e + f u (u = w^6 - 9) as a flat Fq12"""
    out = [0] * 12
    out[0], out[6] = (c[0] - 9 * c[1]) % P, c[1]
    return out


def _shift(a, k):
    """a * w^k"""
    e = [0] * 12
    e[k] = 1
    return f12_mul(a, e)


def untwist(pt):
    """E'(Fq2) -> E(Fq12): (x, y) -> (x w^2, y w^3)"""
    return (_shift(_embed_fq2(pt[0]), 2), _shift(_embed_fq2(pt[1]), 3))


def _f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def _line(p1, p2, t):
    """line through p1, p2 (points over Fq12) evaluated at t (py-style affine; vertical lines included)"""
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_mul(_f12_sub(y2, y1), f12_inv(_f12_sub(x2, x1)))
    elif y1 == y2:
        m = f12_mul([3 * c % P for c in f12_mul(x1, x1)], f12_inv([2 * c % P for c in y1]))
    else:
        return _f12_sub(xt, x1)
    return _f12_sub(f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))


def _ec12_add(p1, p2):
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2 and y1 == y2:
        m = f12_mul([3 * c % P for c in f12_mul(x1, x1)], f12_inv([2 * c % P for c in y1]))
    else:
        m = f12_mul(_f12_sub(y2, y1), f12_inv(_f12_sub(x2, x1)))
    x3 = _f12_sub(_f12_sub(f12_mul(m, m), x1), x2)
    return (x3, _f12_sub(f12_mul(m, _f12_sub(x1, x3)), y1))


def miller_loop(q, p):
    """f_{6x+2,Q}(P) with the two Frobenius lines; q in G2 (affine over Fq2), p in G1 (affine ints)"""
    if q is None or p is None:
        return F12_ONE
    Q = untwist(q)
    Pt = ([p[0]] + [0] * 11, [p[1]] + [0] * 11)
    R, f = Q, F12_ONE
    for bit in bin(ATE_LOOP)[3:]:
        f = f12_mul(f12_mul(f, f), _line(R, R, Pt))
        R = _ec12_add(R, R)
        if bit == "1":
            f = f12_mul(f, _line(R, Q, Pt))
            R = _ec12_add(R, Q)
    Q1 = (f12_pow(Q[0], P), f12_pow(Q[1], P))
    nQ2 = (f12_pow(Q1[0], P), [(-c) % P for c in f12_pow(Q1[1], P)])
    f = f12_mul(f, _line(R, Q1, Pt))
    R = _ec12_add(R, Q1)
    f = f12_mul(f, _line(R, nQ2, Pt))
    return f


def pairing(q, p):
    return f12_pow(miller_loop(q, p), FINAL_EXP)


def pairings_product_is_identity(pairs):
    """util/arithmetic.rs:24-33: pairs of (G1 affine, G2 affine)"""
    f = F12_ONE
    for g1, g2 in pairs:
        f = f12_mul(f, miller_loop(g2, g1))
    return f12_pow(f, FINAL_EXP) == F12_ONE


def to_tower(a):
    """flat coefficients -> 6 Fq2 coefficients of w^0..w^5 (the C++ tower's basis): a_k = e_k - 9 f_k, a_{k+6} = f_k"""
    return [((a[k] + 9 * a[k + 6]) % P, a[k + 6]) for k in range(6)]
