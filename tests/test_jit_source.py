"""CPU: the source csrc/jit.cpp emits for a register program computes the program.  The generator keeps products PENDING and
emits sums of products as shared-reduction dot products (ff.cuh dot); here random programs - register reuse as the
expression compiler produces it (destination = an operand register), chains, negations, moves, overwritten operands - go
through lh_debug_jit_source, the emitted statements are parsed and evaluated over the integers mod r, and the result is
compared with the program run instruction by instruction."""
import ctypes as C
import random
import re

import pytest

from halo2_lasso_amd import _ffi
from oracle.pyref.field import R_MOD as P

ADD, SUB, MUL, NEG, MOV = range(5)
REG, ATOM, CONST = range(3)


def word(op, dst, a, b=(REG, 0)):
    return [op | dst << 4 | a[0] << 8 | b[0] << 10, a[1] | b[1] << 16]


def run_program(prog, atoms, consts, result_reg):
    regs = [0] * 16
    val = lambda o: regs[o[1]] if o[0] == REG else atoms[o[1]] if o[0] == ATOM else consts[o[1]]
    for op, dst, a, b in prog:
        x = val(a)
        if op == NEG:
            regs[dst] = -x % P
        elif op == MOV:
            regs[dst] = x
        else:
            y = val(b)
            regs[dst] = (x + y) % P if op == ADD else (x - y) % P if op == SUB else x * y % P
    return regs[result_reg]


def source_of(prog, num_regs, result_reg, degree=3):
    lib = C.CDLL(_ffi.LIB_PATH)
    fn = lib.lh_debug_jit_source
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    words = []
    for op, dst, a, b in prog:
        words += word(op, dst, a, b)
    arr = (C.c_uint32 * len(words))(*words)
    n = C.c_size_t(0)
    assert fn(arr, len(prog), num_regs, result_reg, degree, None, 0, C.byref(n)) == 0
    buf = C.create_string_buffer(n.value + 1)
    assert fn(arr, len(prog), num_regs, result_reg, degree, buf, n.value + 1, C.byref(n)) == 0
    return buf.value.decode()


def eval_source(src, atoms, consts):
    """the body of the kernel's inner loop: `const Fr tK = at_x(...)` loads, `rK = ...;` statements, dot blocks"""
    body = src[src.index("if (b < a.size) {"):src.index("acc = add(acc, r")]
    env = {}

    def split2(args):  # "A, B" at nesting depth 0
        depth = 0
        for i, ch in enumerate(args):
            depth += ch == "("
            depth -= ch == ")"
            if ch == "," and depth == 0:
                return args[:i], args[i + 1:]
        raise AssertionError(args)

    def term(t):
        t = t.strip()
        if t == "Fr::zero()":
            return 0
        m = re.fullmatch(r"sub\(Fr::zero\(\), (.+)\)", t)
        if m:
            return -term(m.group(1)) % P
        m = re.fullmatch(r"([rf]\d+)", t)
        if m:
            return env.get(m.group(1), 0)
        m = re.fullmatch(r"mul\((.+)\)", t)
        if m:
            x, y = split2(m.group(1))
            return term(x) * term(y) % P
        m = re.fullmatch(r"t(\d+)", t)
        if m:
            return atoms[int(m.group(1))]
        m = re.fullmatch(r"a\.consts\[(\d+)\]", t)
        assert m, t
        return consts[int(m.group(1))]

    def split_all(args):
        out, depth, cur = [], 0, ""
        for ch in args:
            depth += ch == "("
            depth -= ch == ")"
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        out.append(cur)
        return out

    stmts = 0
    lines = [ln.strip() for ln in body.splitlines()]
    i = 0
    while i < len(lines):
        ln = lines[i]
        i += 1
        if not ln or ln.startswith(("if (", "Fr r", "Fr f", "const Fr t", "}")) or ln == "{":
            if ln == "{":  # a dot block: xa, xb, assignment to a temporary
                ma = re.fullmatch(r"const Fr xa\[(\d+)\] = \{(.*)\};", lines[i])
                mb = re.fullmatch(r"const Fr xb\[(\d+)\] = \{(.*)\};", lines[i + 1])
                md = re.fullmatch(r"(f\d+) = dot<FrParams, (\d+)>\(xa, xb\);", lines[i + 2])
                assert ma and mb and md, lines[i:i + 3]
                xa, xb = [term(t) for t in split_all(ma.group(2))], [term(t) for t in split_all(mb.group(2))]
                assert len(xa) == len(xb) == int(ma.group(1)) == int(md.group(2)) and 2 <= len(xa) <= 4
                env[md.group(1)] = sum(x * y for x, y in zip(xa, xb)) % P
                stmts += 1
                i += 3
            continue
        m = re.fullmatch(r"const Fr (f\d+) = (.+);", ln)
        if m:  # a single pending product materialised
            env[m.group(1)] = term(m.group(2))
            stmts += 1
            continue
        m = re.fullmatch(r"r(\d+) = (mul|add|sub)\((.*)\);", ln)
        if m:
            a, b = split2(m.group(3))
            x, y = term(a), term(b)
            env["r" + m.group(1)] = x * y % P if m.group(2) == "mul" else (x + y) % P if m.group(2) == "add" else (x - y) % P
        else:
            m = re.fullmatch(r"r(\d+) = (.+);", ln)
            assert m, ln
            env["r" + m.group(1)] = term(m.group(2))
        stmts += 1
    res = re.search(r"acc = add\(acc, r(\d+)\)", src)
    return env.get("r" + res.group(1), 0), stmts, body.count("dot<FrParams")


def random_program(rng, num_atoms, num_consts, length):
    """register programs in the style of expr.cpp's builder: a binary instruction reuses an operand register as its
    destination when it has one, fresh registers come from a free list - plus arbitrary overwrites to stress the hazards"""
    live, prog = [], []
    free = list(range(8))

    def leaf():
        return (ATOM, rng.randrange(num_atoms)) if rng.random() < 0.8 else (CONST, rng.randrange(num_consts))

    def operand():
        if live and rng.random() < 0.55:
            return (REG, rng.choice(live))
        return leaf()

    for _ in range(length):
        kind = rng.random()
        if kind < 0.08 and live:
            r = rng.choice(live)
            prog.append((NEG, r, (REG, r), (REG, 0)))
            continue
        if kind < 0.12:
            dst = free.pop() if free and rng.random() < 0.7 else rng.randrange(8)
            prog.append((MOV, dst, operand(), (REG, 0)))
            if dst not in live:
                live.append(dst)
            continue
        a, b = operand(), operand()
        op = rng.choice([MUL, MUL, MUL, ADD, ADD, SUB])
        wild = rng.random() < 0.1  # an arbitrary destination: overwrites whatever is there
        if wild:
            dst = rng.randrange(8)
        elif a[0] == REG:
            dst = a[1]
        elif b[0] == REG:
            dst = b[1]
        else:
            dst = free.pop() if free else rng.randrange(8)
        prog.append((op, dst, a, b))
        for o in (a, b):  # the builder releases operand registers that are not the destination
            if o[0] == REG and o[1] != dst and o[1] in live and rng.random() < 0.7:
                live.remove(o[1])
                free.append(o[1])
        if dst in free:
            free.remove(dst)
        if dst not in live:
            live.append(dst)
    result = rng.choice(live) if live else 0
    return prog, result


@pytest.mark.parametrize("seed", range(300))
def test_emitted_source_computes_the_program(seed):
    rng = random.Random(seed)
    num_atoms, num_consts = 6, 3
    prog, result = random_program(rng, num_atoms, num_consts, rng.randrange(4, 60))
    atoms = [rng.randrange(P) for _ in range(num_atoms)]
    consts = [rng.randrange(P) for _ in range(num_consts)]
    want = run_program(prog, atoms, consts, result)
    got, _, _ = eval_source(source_of(prog, 8, result), atoms, consts)
    assert got == want


def test_sums_of_products_become_dot_products():
    """t0 t1 + t2 t3 - t4 t5 + c0 t0 (the shape of a gate constraint): ONE reduction instead of four products"""
    prog = [(MUL, 0, (ATOM, 0), (ATOM, 1)), (MUL, 1, (ATOM, 2), (ATOM, 3)), (ADD, 0, (REG, 0), (REG, 1)),
            (MUL, 1, (ATOM, 4), (ATOM, 5)), (SUB, 0, (REG, 0), (REG, 1)), (MUL, 1, (CONST, 0), (ATOM, 0)),
            (ADD, 0, (REG, 0), (REG, 1)), (MUL, 0, (REG, 0), (ATOM, 2))]  # ... times a selector: the sum is needed as a value
    rng = random.Random(7)
    atoms, consts = [rng.randrange(P) for _ in range(6)], [rng.randrange(P)]
    src = source_of(prog, 2, 0)
    got, stmts, dots = eval_source(src, atoms, consts)
    assert got == run_program(prog, atoms, consts, 0)
    assert dots == 1 and "dot<FrParams, 4>" in src and src.count("mul(") == 1  # the dot, then the product with the selector
