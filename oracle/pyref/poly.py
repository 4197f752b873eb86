"""Multilinear polynomials in evaluation form.  TEST INFRASTRUCTURE ONLY.

Follows reference plonkish_backend/src/poly/multilinear.rs.  Index bit i of an evaluation
table corresponds to variable i (LSB = variable 0).
"""
from .field import R_MOD as P


def eq_xy(y):
    """multilinear.rs:91-127: eq(y)[b] = prod_i (b_i ? y_i : 1 - y_i), bit i <-> y_i."""
    evals = [1]
    for y_i in reversed(y):
        nxt = [0] * (2 * len(evals))
        for k, e in enumerate(evals):
            hi = e * y_i % P
            nxt[2 * k + 1] = hi
            nxt[2 * k] = (e - hi) % P
        evals = nxt
    return evals


def fix_var(evals, x_i):
    """multilinear.rs:179-183,599-618 (`merge_into`, distance 1): binds variable 0:
    out[b] = e[2b] + (e[2b+1] - e[2b]) * x."""
    return [((evals[2 * b + 1] - evals[2 * b]) * x_i + evals[2 * b]) % P
            for b in range(len(evals) // 2)]


def fix_last_var(evals, x_i):
    """Binds the TOP variable (multilinear.rs:158-177 `fix_last_vars`, one variable)."""
    h = len(evals) // 2
    return [(evals[b] + (evals[h + b] - evals[b]) * x_i) % P for b in range(h)]


def evaluate(evals, x):
    """multilinear.rs:137-156 (value only; the 0/1 shortcut there is an optimisation)."""
    assert len(evals) == 1 << len(x)
    for x_i in x:
        evals = fix_var(evals, x_i)
    return evals[0] % P


def eq_xy_eval(x, y):
    """piop/sum_check.rs:112-121"""
    assert len(x) == len(y) and len(x) > 0
    acc = 1
    for a, b in zip(x, y):
        acc = acc * ((2 * a * b + 1 - a - b) % P) % P
    return acc


def identity_eval(x):
    """piop/sum_check.rs:123-125"""
    return sum(x_i << i for i, x_i in enumerate(x)) % P
