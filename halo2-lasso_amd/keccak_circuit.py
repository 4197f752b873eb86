"""Keccak-f as a PLONKish circuit whose bitwise operations are Lasso lookups (BASELINE.json configs[4]: "HyperPlonk+Lasso
end-to-end prove of Keccak-f[1600] circuit").  Host side only (numpy): the straight-line program, its witness, the fixed
columns and the copy constraints; `synthetic.keccak_f` puts them on the device, `hyperplonk.keccak_circuit_info` holds the
gates.  The reference has no Keccak circuit (its only hash circuit is SHA-256, benchmark/src/halo2/circuit.rs:389-479)
and no Lasso: the layout below is this build's; the permutation itself is FIPS-202's (tests/test_keccak_circuit.py checks
the program against a plain Keccak-f and every constraint on the witness).

Cells are UNITS of `ub` bits (bytes for Keccak-f[1600]; nibbles in the small parity circuits): a lane of w bits is w/ub
units, little-endian.  One row = one operation on units, operands u = c_x + s_x x and v = c_y + s_y y (fixed columns
c, s; witness cells x, y), result cell o:

  XOR row   d_X = 2^ub u + v,  o = a_X        Lasso lookup: a_X = T_xor[d_X] on every row (one chunk of 2 ub bits)
  AND row   d_A = 2^ub u + v,  o = a_A        Lasso lookup: a_A = T_and[d_A]
  LIN row   o = u + v                         (no lookup: recombines the two masked halves of a rotated unit)

theta and iota are XOR rows (iota: v a constant), chi is  o = B ^ (~B' & B'')  as an AND row with u = (2^ub - 1) - x and
an XOR row, rho splits every unit with two constant-mask AND rows and recombines neighbours with a LIN row
(o = lo 2^s + hi 2^-(ub-s): the high part is a multiple of 2^(ub-s), so the field quotient is the integer one); rotations
by whole units and pi are wiring.  Every operand cell is tied to the result cell it comes from by a copy constraint
(permutation argument over x, y, o); the input units of every permutation enter through an AND row with x = y (which
forces x into range: T_and[257 x] = x only for a unit x).  Operands of lookups are therefore constants or lookup results.

Polys: pi | q_xor q_and q_lin c_x s_x c_y s_y | x y o d_X a_X d_A a_A   (indices 0 | 1..7 | 8..14).
"""
import numpy as np

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617

# FIPS-202: rotation offsets r[x][y] and round constants
RHO = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
      0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
      0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
      0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
      0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]

IDLE, XOR, AND, LIN = 0, 1, 2, 3
COL_X, COL_Y, COL_O = 0, 1, 2


def num_rounds(w):
    return 12 + 2 * (w.bit_length() - 1)


class Program:
    """rows of one permutation: kind, operand sources (cell id = column * rows + row, -1: none), fixed columns as
    (c_x, s_x, c_y, s_y) with s in {0, 1, -1} on lookup rows and (shift, ub - shift) coded on LIN rows"""

    def __init__(self, w, ub, rounds):
        assert w % ub == 0 and ub in (4, 8) and w in (4, 8, 16, 32, 64)
        self.w, self.ub, self.nb, self.rounds = w, ub, w // ub, rounds
        self.kind, self.src_x, self.src_y = [], [], []
        self.cx, self.sx, self.cy, self.sy = [], [], [], []
        self.inputs, self.outputs = [], []   # rows whose x cell is an input unit / result cells of the final state

    # ---- emitting rows; a unit is referred to by the row whose o cell holds it
    def _emit(self, kind, a, b, cx, sx, cy, sy, x_free=False):
        r = len(self.kind)
        self.kind.append(kind)
        self.src_x.append(-1 if a is None or x_free else ("o", a))
        self.src_y.append(-1 if b is None else b if isinstance(b, tuple) else ("o", b))
        self.cx.append(cx), self.sx.append(sx), self.cy.append(cy), self.sy.append(sy)
        return r

    def input_unit(self):
        r = len(self.kind)
        self._emit(AND, r, ("x", r), 0, 1, 0, 1, x_free=True)  # o = x & y with y tied to x: x is a unit
        self.inputs.append(r)
        return r

    def xor(self, a, b):
        return self._emit(XOR, a, b, 0, 1, 0, 1)

    def xor_const(self, a, k):
        return self._emit(XOR, a, None, 0, 1, k, 0)

    def and_const(self, a, mask):
        return self._emit(AND, a, None, 0, 1, mask, 0)

    def andn(self, a, b):  # (~a) & b
        return self._emit(AND, a, b, (1 << self.ub) - 1, -1, 0, 1)

    def lin(self, lo, hi, s):  # lo * 2^s + hi / 2^(ub - s)
        return self._emit(LIN, lo, hi, 0, ("shl", s), 0, ("shr", self.ub - s))

    def rot(self, lane, r):
        r %= self.w
        q, s = divmod(r, self.ub)
        nb = self.nb
        if s == 0:
            return [lane[(k - q) % nb] for k in range(nb)]
        lo_mask = (1 << (self.ub - s)) - 1
        hi_mask = ((1 << self.ub) - 1) ^ lo_mask
        lo = [self.and_const(u, lo_mask) for u in lane]
        hi = [self.and_const(u, hi_mask) for u in lane]
        return [self.lin(lo[(k - q) % nb], hi[(k - q - 1) % nb], s) for k in range(nb)]


def keccak_program(w=64, ub=8, rounds=None):
    """the rows of one Keccak-f[25 w] (the first `rounds` rounds; default all of them)"""
    p = Program(w, ub, num_rounds(w) if rounds is None else rounds)
    nb, umask = p.nb, (1 << ub) - 1
    A = [[[p.input_unit() for _ in range(nb)] for _y in range(5)] for _x in range(5)]  # A[x][y]: lane = units
    # (FIPS-202 state order: lane (x, y) is lane index x + 5 y; inputs were created x-major, the driver maps them)
    p.input_order = [(x, y, k) for x in range(5) for y in range(5) for k in range(nb)]
    for rnd in range(p.rounds):
        C = []
        for x in range(5):
            acc = A[x][0]
            for y in range(1, 5):
                acc = [p.xor(acc[k], A[x][y][k]) for k in range(nb)]
            C.append(acc)
        D = []
        for x in range(5):
            r1 = p.rot(C[(x + 1) % 5], 1)
            D.append([p.xor(C[(x - 1) % 5][k], r1[k]) for k in range(nb)])
        A = [[[p.xor(A[x][y][k], D[x][k]) for k in range(nb)] for y in range(5)] for x in range(5)]
        B = [[None] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = p.rot(A[x][y], RHO[x][y])
        A = [[[p.xor(B[x][y][k], p.andn(B[(x + 1) % 5][y][k], B[(x + 2) % 5][y][k])) for k in range(nb)]
              for y in range(5)] for x in range(5)]
        rc = RC[rnd] & ((1 << w) - 1)
        for k in range(nb):
            byte = (rc >> (ub * k)) & umask
            if byte:
                A[0][0][k] = p.xor_const(A[0][0][k], byte)
    p.outputs = [A[x][y][k] for (x, y, k) in p.input_order]
    p.num_rows = len(p.kind)
    return p


def evaluate(prog, states):
    """states: (B, 25) array of lane values (lane index x + 5 y, FIPS-202).  Returns (x, y, o) value arrays of shape
    (rows, B) - the operand CELLS (before c + s * cell) and results - and the (B, 25) output lanes."""
    B = states.shape[0]
    R, ub, umask = prog.num_rows, prog.ub, (1 << prog.ub) - 1
    xs, ys, os_ = (np.zeros((R, B), dtype=np.int64) for _ in range(3))
    in_pos = {r: i for i, r in enumerate(prog.inputs)}
    for r in range(R):
        sx_, sy_ = prog.src_x[r], prog.src_y[r]
        if r in in_pos:
            x_, y_, k_ = prog.input_order[in_pos[r]]
            xv = (states[:, x_ + 5 * y_].astype(np.uint64) >> np.uint64(ub * k_)).astype(np.int64) & umask
        else:
            xv = os_[sx_[1]] if sx_ != -1 else np.zeros(B, dtype=np.int64)
        yv = np.zeros(B, dtype=np.int64) if sy_ == -1 else (xv if sy_[0] == "x" else os_[sy_[1]])
        xs[r], ys[r] = xv, yv
        kind = prog.kind[r]
        if kind == LIN:
            os_[r] = (xv << prog.sx[r][1]) + (yv >> prog.sy[r][1])
        else:
            u, v = prog.cx[r] + prog.sx[r] * xv, prog.cy[r] + prog.sy[r] * yv
            os_[r] = (u ^ v) if kind == XOR else (u & v)
    out = np.zeros((B, 25), dtype=np.uint64)
    for i, (x_, y_, k_) in enumerate(prog.input_order):
        out[:, x_ + 5 * y_] |= os_[prog.outputs[i]].astype(np.uint64) << np.uint64(ub * k_)
    return xs, ys, os_, out


class KeccakColumns:
    """integer columns of a circuit of 2^k rows holding `instances` permutations back to back (the rest idle)"""


def build_columns(prog, k, states):
    """-> KeccakColumns: fixed columns (small ints, with the field constants of LIN rows kept symbolic in `lin_shift`),
    witness columns, and the permutation as target (column, row) per cell of x, y, o"""
    size, R, B, ub = 1 << k, prog.num_rows, states.shape[0], prog.ub
    # row 0 takes no part in copy constraints (preprocessor.rs:172-203 asserts it) and the last row stays empty as in
    # the reference's generators: the permutations occupy rows 1 .. B R
    row0 = 1
    assert row0 + B * R <= size - 1, "circuit too small: %d rows per permutation" % R
    xs, ys, os_, out = evaluate(prog, states)
    kind = np.array(prog.kind, dtype=np.int64)
    col = KeccakColumns()
    col.k, col.size, col.rows_per_instance, col.instances, col.ub, col.outputs = k, size, R, B, ub, out
    col.first_row = row0
    n = B * R

    def tile(a):  # program-space vector -> circuit column (instances back to back, idle rows zero)
        full = np.zeros(size, dtype=np.int64)
        full[row0:row0 + n] = np.tile(np.asarray(a, dtype=np.int64), B)
        return full

    def cells(a):  # (R, B) -> circuit column
        full = np.zeros(size, dtype=np.int64)
        full[row0:row0 + n] = a.T.reshape(-1)
        return full

    col.q_xor, col.q_and, col.q_lin = tile(kind == XOR), tile(kind == AND), tile(kind == LIN)
    lin = kind == LIN
    col.cx, col.cy = tile(prog.cx), tile(prog.cy)
    col.sx = tile([0 if isinstance(s, tuple) else s for s in prog.sx])            # -1, 0, 1 on lookup rows
    col.sy = tile([0 if isinstance(s, tuple) else s for s in prog.sy])
    col.lin_shl = tile([s[1] if isinstance(s, tuple) else 0 for s in prog.sx])    # LIN rows: s_x = 2^shl
    col.lin_shr = tile([s[1] if isinstance(s, tuple) else 0 for s in prog.sy])    # LIN rows: s_y = 2^-shr
    col.x, col.y, col.o = cells(xs), cells(ys), cells(os_)
    u = col.cx + col.sx * col.x
    v = col.cy + col.sy * col.y
    d = (u << ub) + v
    col.d_xor, col.d_and = np.where(col.q_xor == 1, d, 0), np.where(col.q_and == 1, d, 0)
    col.a_xor, col.a_and = np.where(col.q_xor == 1, col.o, 0), np.where(col.q_and == 1, col.o, 0)
    assert not lin.any() or (col.lin_shl[col.q_lin == 1] > 0).all()

    # ---- copy constraints: every operand cell with a source joins the cycle of that source cell
    def cell_id(c, r):
        return c * R + r
    src, dst = [], []
    for r in range(R):
        for c, s in ((COL_X, prog.src_x[r]), (COL_Y, prog.src_y[r])):
            if s == -1:
                continue
            src.append(cell_id(COL_X if s[0] == "x" else COL_O, s[1]))
            dst.append(cell_id(c, r))
    src, dst = np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64)
    order = np.argsort(src, kind="stable")
    src, dst = src[order], dst[order]
    nxt = np.arange(3 * R, dtype=np.int64)             # program-space permutation: cell -> next cell of its cycle
    first = np.ones(len(src), dtype=bool)
    first[1:] = src[1:] != src[:-1]
    last = np.ones(len(src), dtype=bool)
    last[:-1] = src[1:] != src[:-1]
    nxt[src[first]] = dst[first]                        # source -> its first consumer
    nxt[dst[~last]] = dst[1:][~last[:-1]]               # consumer -> next consumer of the same source
    nxt[dst[last]] = src[last]                          # last consumer -> back to the source
    # replicate over the instances: (column, row) of instance t = (column, t R + row)
    t_off = (row0 + np.arange(B, dtype=np.int64) * R)[:, None]
    col.perm_col, col.perm_row = [], []
    for c in range(3):
        tgt = nxt[c * R:(c + 1) * R]
        pc = np.full(size, c, dtype=np.int64)
        pr = np.arange(size, dtype=np.int64)
        pc[row0:row0 + n] = np.tile(tgt // R, B)
        pr[row0:row0 + n] = (tgt % R + t_off).reshape(-1)
        col.perm_col.append(pc), col.perm_row.append(pr)
    return col


def check_columns(col):
    """every constraint of the circuit on the integer columns (tests): gates, lookups, copy constraints"""
    ub, umask = col.ub, (1 << col.ub) - 1
    u = col.cx + col.sx * col.x
    v = col.cy + col.sy * col.y
    lk = (col.q_xor + col.q_and) == 1
    assert ((u[lk] >= 0) & (u[lk] <= umask) & (v[lk] >= 0) & (v[lk] <= umask)).all()
    assert (col.d_xor[col.q_xor == 1] == ((u << ub) + v)[col.q_xor == 1]).all()
    assert (col.d_and[col.q_and == 1] == ((u << ub) + v)[col.q_and == 1]).all()
    assert (col.o[col.q_xor == 1] == col.a_xor[col.q_xor == 1]).all() and (col.o[col.q_and == 1] == col.a_and[col.q_and == 1]).all()
    # lookups hold on EVERY row
    assert (col.a_xor == ((col.d_xor >> ub) ^ (col.d_xor & umask))).all()
    assert (col.a_and == ((col.d_and >> ub) & (col.d_and & umask))).all()
    ln = col.q_lin == 1
    hi = col.y[ln]
    assert (hi % (1 << col.lin_shr[ln]) == 0).all()
    assert (col.o[ln] == (col.x[ln] << col.lin_shl[ln]) + (hi >> col.lin_shr[ln])).all() and (col.o[ln] <= umask).all()
    vals = [col.x, col.y, col.o]
    seen = np.zeros((3, col.size), dtype=np.int64)
    for c in range(3):
        tgt = np.stack(vals)[col.perm_col[c], col.perm_row[c]]
        assert (tgt == vals[c]).all(), "copy constraint between unequal cells"
        np.add.at(seen, (col.perm_col[c], col.perm_row[c]), 1)
    assert (seen == 1).all(), "the copy constraints are not a permutation"
    return True


def reference_keccak_f(lanes, w=64, rounds=None):
    """plain Keccak-f[25 w] on a list of 25 lane integers (FIPS-202 order x + 5 y): the permutation the circuit encodes"""
    mask = (1 << w) - 1
    rot = lambda v, r: ((v << (r % w)) | (v >> (w - r % w))) & mask if r % w else v
    a = [[lanes[x + 5 * y] for y in range(5)] for x in range(5)]
    for rnd in range(num_rounds(w) if rounds is None else rounds):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rot(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = rot(a[x][y], RHO[x][y])
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & mask & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= RC[rnd] & mask
    return [a[i % 5][i // 5] for i in range(25)]


def field_columns(col):
    """-> (preprocess, witness): the 7 + 7 polys as lists of integers mod r (the small-circuit path: PlonkishCircuitInfo
    with host tables; synthetic.keccak_f builds the same columns on the device)"""
    lin = col.q_lin == 1
    inv2 = pow(2, R_MOD - 2, R_MOD)
    s_x = [(1 << int(a)) if l else int(s) % R_MOD for l, a, s in zip(lin, col.lin_shl, col.sx)]
    s_y = [pow(inv2, int(a), R_MOD) if l else int(s) % R_MOD for l, a, s in zip(lin, col.lin_shr, col.sy)]
    as_list = lambda a: [int(v) for v in a]
    pre = [as_list(col.q_xor), as_list(col.q_and), as_list(col.q_lin), as_list(col.cx), s_x, as_list(col.cy), s_y]
    wit = [as_list(c) for c in (col.x, col.y, col.o, col.d_xor, col.a_xor, col.d_and, col.a_and)]
    return pre, wit


def copy_cycles(col, first_poly=8):
    """the copy constraints as cycles of (poly, row) (PlonkishCircuitInfo.permutations, backend.rs:66-69)"""
    seen = [np.zeros(col.size, dtype=bool) for _ in range(3)]
    cycles = []
    for c in range(3):
        moved = np.nonzero((col.perm_col[c] != c) | (col.perm_row[c] != np.arange(col.size)))[0]
        for r in moved:
            if seen[c][r]:
                continue
            cyc, cc, rr = [], c, int(r)
            while not seen[cc][rr]:
                seen[cc][rr] = True
                cyc.append((first_poly + cc, rr))
                cc, rr = int(col.perm_col[cc][rr]), int(col.perm_row[cc][rr])
            cycles.append(cyc)
    return cycles
