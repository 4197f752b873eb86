"""Synthetic circuits for measurement (bench.py --workload hyperplonk, tools/hp_bench.py).

`vanilla_plonk_with_lookup(ctx, k, seed)` builds a satisfied circuit of the reference's
`vanilla_plonk_with_lookup` shape (backend/hyperplonk/util.rs:63-86,216-316): 13 polys
(pi | q_l q_r q_m q_o q_c q_lookup t_l t_r t_o | w_l w_r w_o), add / mul gates, one 3-column lookup on a quarter of
the rows, copy constraints between the two halves of the table, no instances.  Built with numpy and the device
field ops so that 2^20..2^24 rows take seconds; the reference's generator (a Python loop in the oracle) is used
for the small parity circuits instead.
"""
import numpy as np

from . import hyperplonk as hp


class SyntheticCircuit:
    """host arrays are (2^k, 4) uint64 Montgomery limbs, exactly the bytes the device holds"""


def vanilla_plonk_with_lookup(ctx, k, seed=None):
    import halo2_lasso_amd as hl
    size = 1 << k
    lib = ctx.lib
    rng = np.random.default_rng(k if seed is None else seed)

    def rand_fr(n):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < r: a uniform-looking field element in Montgomery form
        return a

    def const_fr(v, n):
        return np.tile(np.frombuffer(hl.fr_to_bytes(v), dtype=np.uint64), (n, 1))

    def up(a):
        return hl.MultilinearPolynomial(ctx, ctx.upload(np.ascontiguousarray(a).tobytes()), k)

    def down(p):
        return np.frombuffer(p.buf.read(), dtype=np.uint64).reshape(size, 4).copy()

    def binop(fn, a, b):
        out = hl.MultilinearPolynomial(ctx, ctx.alloc(32 * size), k)
        hl._check(fn(ctx.h, a.ptr, b.ptr, size, out.ptr))
        return out

    rows = np.arange(size)
    live = rows < size - 1                     # the reference leaves the last row empty
    is_add, is_mul = live & (rows % 4 < 2), live & (rows % 4 == 2)
    is_lookup = live & (rows % 4 == 3)
    gate = is_add | is_mul
    zero, one, minus1 = const_fr(0, size), const_fr(1, size), const_fr(hl.R_MOD - 1, size)
    sel = lambda m, v: np.where(m[:, None], v, zero)
    q_l = q_r = sel(is_add, one)
    q_m, q_o, q_c = sel(is_mul, one), sel(gate, minus1), sel(gate, rand_fr(size))
    q_lookup = sel(is_lookup, one)
    t_l, t_r, t_o = (sel(rows >= 2, rand_fr(size)) for _ in range(3))
    t_idx = rng.integers(1, size, size=size)
    w_l, w_r = sel(gate, rand_fr(size)), sel(gate, rand_fr(size))
    w_l[is_lookup], w_r[is_lookup] = t_l[t_idx[is_lookup]], t_r[t_idx[is_lookup]]
    d_ql, d_qm, d_qc = up(q_l), up(q_m), up(q_c)

    def out_column(wl, wr):
        """w_o = q_l w_l + q_r w_r + q_m w_l w_r + q_c on gate rows (q_o = -1), t_o[t] on lookup rows"""
        a, b = up(wl), up(wr)
        lin = binop(lib.lh_fr_mul, d_ql, binop(lib.lh_fr_add, a, b))
        quad = binop(lib.lh_fr_mul, d_qm, binop(lib.lh_fr_mul, a, b))
        wo = down(binop(lib.lh_fr_add, binop(lib.lh_fr_add, lin, quad), d_qc))
        wo[is_lookup] = t_o[t_idx[is_lookup]]
        return wo

    # copies: gate rows of the second half take w_l from w_o and w_r from w_r of the row half a table earlier
    w_o = out_column(w_l, w_r)
    half = size // 2
    dst = rows[gate & (rows > half)]           # source row = dst - half >= 1
    src = dst - half
    w_l[dst], w_r[dst] = w_o[src], w_r[src]
    w_o = out_column(w_l, w_r)
    # permutation polys over (w_l, w_r, w_o) = polys 10, 11, 12 (preprocessor.rs:172-203): 2-cycles swap ids
    ident = lambda p: (np.uint64(p) << np.uint64(k)) + rows.astype(np.uint64)
    perm = [ident(0), ident(1), ident(2)]
    perm[0][dst], perm[2][src] = ident(2)[src], ident(0)[dst]      # (w_o, src) <-> (w_l, dst)
    perm[1][dst], perm[1][src] = ident(1)[src], ident(1)[dst]      # (w_r, src) <-> (w_r, dst)
    d_perm, h_perm = [], []
    for p in perm:
        out = hl.MultilinearPolynomial(ctx, ctx.alloc(32 * size), k)
        staged = ctx.upload(p.tobytes())
        hl._check(lib.lh_fr_from_u64(ctx.h, staged.ptr, size, out.ptr))
        ctx.sync()
        d_perm.append(out)

    c = SyntheticCircuit()
    c.k, c.num_copies, c.num_lookups = k, len(dst), int(is_lookup.sum())
    # nine (device-resident) preprocess polys: compose() only needs their count
    c.info = hp.vanilla_plonk_with_lookup_circuit_info(k, 0, [[]] * 9, [[(10, 1)], [(11, 1)], [(12, 1)]])
    c.h_preprocess = [q_l, q_r, q_m, q_o, q_c, q_lookup, t_l, t_r, t_o]
    c.h_witness = [w_l, w_r, w_o]
    c.d_preprocess = [up(a) for a in c.h_preprocess]
    c.d_permutation = d_perm
    c.d_witness = [up(a) for a in c.h_witness]
    c.instances = [[]]
    return c


def vanilla_plonk_with_lasso(ctx, k, kind="and", seed=None):
    """The BASELINE.json configs[4] stand-in at scale (hyperplonk.vanilla_plonk_with_lasso_circuit_info): vanilla add / mul
    gates on half of the rows, a 32-bit AND / XOR / range lookup proven by Lasso on the other half (w_o = a on lookup rows),
    copy constraints between the two halves of the table, no instances.  16 polys for the bitwise tables:
    pi | q_l q_r q_m q_o q_c q_lookup | w_l w_r w_o d_0..d_3 a."""
    import halo2_lasso_amd as hl
    size = 1 << k
    lib = ctx.lib
    rng = np.random.default_rng(1000 + k if seed is None else seed)
    table = hl.LassoTable.range(2, 16) if kind == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, 4, 16)
    c = table.c

    def rand_fr(n):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a

    def const_fr(v, n):
        return np.tile(np.frombuffer(hl.fr_to_bytes(v), dtype=np.uint64), (n, 1))

    def up(a):
        return hl.MultilinearPolynomial(ctx, ctx.upload(np.ascontiguousarray(a).tobytes()), k)

    def down(p):
        return np.frombuffer(p.buf.read(), dtype=np.uint64).reshape(size, 4).copy()

    def binop(fn, a, b):
        out = hl.MultilinearPolynomial(ctx, ctx.alloc(32 * size), k)
        hl._check(fn(ctx.h, a.ptr, b.ptr, size, out.ptr))
        return out

    def fr_of_u64(v):
        out = hl.MultilinearPolynomial(ctx, ctx.alloc(32 * size), k)
        staged = ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).tobytes())
        hl._check(lib.lh_fr_from_u64(ctx.h, staged.ptr, size, out.ptr))
        ctx.sync()
        return out

    rows = np.arange(size)
    live = rows < size - 1
    is_add, is_mul = live & (rows % 4 == 0), live & (rows % 4 == 1)
    is_lookup = live & (rows % 4 >= 2)
    gate = is_add | is_mul
    zero, one, minus1 = const_fr(0, size), const_fr(1, size), const_fr(hl.R_MOD - 1, size)
    sel = lambda m, v: np.where(m[:, None], v, zero)
    q_l = q_r = sel(is_add, one)
    q_m, q_o, q_c = sel(is_mul, one), sel(gate, minus1), sel(gate, rand_fr(size))
    q_lookup = sel(is_lookup, one)
    # chunk columns: indices on lookup rows, 0 elsewhere (T[0] = 0 for range / AND / XOR: a = 0 there)
    dims = [np.where(is_lookup, rng.integers(0, 1 << table.l, size=size, dtype=np.uint64), 0).astype(np.uint64)
            for _ in range(c)]
    if kind == "range":
        out = sum(d << np.uint64(16 * j) for j, d in enumerate(dims))
    else:
        x, y = [d >> np.uint64(8) for d in dims], [d & np.uint64(0xff) for d in dims]
        out = sum(((xj & yj) if kind == "and" else (xj ^ yj)) << np.uint64(8 * j) for j, (xj, yj) in enumerate(zip(x, y)))
    d_dims = [fr_of_u64(d) for d in dims]
    d_a = fr_of_u64(out)
    a_host = down(d_a)
    w_l, w_r = sel(live, rand_fr(size)), sel(live, rand_fr(size))
    d_ql, d_qm, d_qc = up(q_l), up(q_m), up(q_c)

    def out_column(wl, wr):
        al, ar = up(wl), up(wr)
        lin = binop(lib.lh_fr_mul, d_ql, binop(lib.lh_fr_add, al, ar))
        quad = binop(lib.lh_fr_mul, d_qm, binop(lib.lh_fr_mul, al, ar))
        wo = down(binop(lib.lh_fr_add, binop(lib.lh_fr_add, lin, quad), d_qc))
        wo[is_lookup] = a_host[is_lookup]
        return wo

    w_o = out_column(w_l, w_r)
    half = size // 2
    dst = rows[gate & (rows > half)]
    src = dst - half
    w_l[dst], w_r[dst] = w_o[src], w_r[src]
    w_o = out_column(w_l, w_r)
    ident = lambda p: (np.uint64(p) << np.uint64(k)) + rows.astype(np.uint64)
    perm = [ident(0), ident(1), ident(2)]
    perm[0][dst], perm[2][src] = ident(2)[src], ident(0)[dst]
    perm[1][dst], perm[1][src] = ident(1)[src], ident(1)[dst]
    d_perm = [fr_of_u64(p) for p in perm]

    circ = SyntheticCircuit()
    circ.k, circ.num_copies, circ.num_lookups, circ.table, circ.kind = k, len(dst), int(is_lookup.sum()), table, kind
    circ.info = hp.vanilla_plonk_with_lasso_circuit_info(k, 0, [[]] * 6, [[(7, 1)], [(8, 1)], [(9, 1)]], table)
    circ.h_preprocess = [q_l, q_r, q_m, q_o, q_c, q_lookup]
    circ.h_witness = [w_l, w_r, w_o] + [down(d) for d in d_dims] + [a_host]
    circ.d_preprocess = [up(a) for a in circ.h_preprocess]
    circ.d_permutation = d_perm
    circ.d_witness = [up(w_l), up(w_r), up(w_o)] + d_dims + [d_a]
    circ.instances = [[]]
    return circ


def keccak_f(ctx, k, w=64, ub=8, rounds=None, seed=None, instances=None, states=None):
    """BASELINE.json configs[4]: Keccak-f[25 w] permutations (default Keccak-f[1600], 35013 rows each) packed into a
    circuit of 2^k rows, every XOR / AND of the permutation a Lasso lookup proven inside HyperPlonk::prove
    (keccak_circuit.py: layout; hyperplonk.keccak_circuit_info: gates).  Random input states; `outputs` holds the
    permuted states.  15 polys, two Lasso lookups (XOR and AND over units of `ub` bits), ~2 copy constraints per row."""
    import halo2_lasso_amd as hl
    from . import keccak_circuit as kc
    size = 1 << k
    lib = ctx.lib
    prog = kc.keccak_program(w, ub, rounds)
    fit = (size - 2) // prog.num_rows
    if fit < 1:
        raise ValueError("a Keccak-f[%d] of %d rounds needs %d rows: 2^%d is too small" % (25 * w, prog.rounds, prog.num_rows, k))
    B = fit if instances is None else instances
    rng = np.random.default_rng(4000 + k if seed is None else seed)
    if states is not None:  # the caller's input states, (B, 25) lanes in FIPS-202 order
        states = np.asarray(states, dtype=np.uint64)
        B = states.shape[0]
    else:
        states = rng.integers(0, 1 << min(w, 63), size=(B, 25), dtype=np.uint64)
        if w == 64:
            states = states * np.uint64(2) + rng.integers(0, 2, size=(B, 25), dtype=np.uint64)
    col = kc.build_columns(prog, k, states)

    def up(a):
        return hl.MultilinearPolynomial(ctx, ctx.upload(np.ascontiguousarray(a).tobytes()), k)

    def down(p):
        return np.frombuffer(p.buf.read(), dtype=np.uint64).reshape(size, 4).copy()

    def fr_of_u64(v):
        out = hl.MultilinearPolynomial(ctx, ctx.alloc(32 * size), k)
        staged = ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).tobytes())
        hl._check(lib.lh_fr_from_u64(ctx.h, staged.ptr, size, out.ptr))
        ctx.sync()
        return out

    def fr_of_values(vals):
        """a column of few distinct field values (python ints mod r) -> host Montgomery limbs"""
        vals = np.asarray(vals, dtype=object)
        distinct = sorted(set(int(v) % hl.R_MOD for v in vals))
        table = np.stack([np.frombuffer(hl.fr_to_bytes(v), dtype=np.uint64) for v in distinct])
        index = {v: i for i, v in enumerate(distinct)}
        return table[np.fromiter((index[int(v) % hl.R_MOD] for v in vals), dtype=np.int64, count=len(vals))]

    lin = col.q_lin == 1
    inv2 = pow(2, hl.R_MOD - 2, hl.R_MOD)
    s_x = np.where(lin, 0, col.sx).astype(object)
    s_y = np.where(lin, 0, col.sy).astype(object)
    for shl in np.unique(col.lin_shl[lin]):
        s_x[lin & (col.lin_shl == shl)] = 1 << int(shl)
    for shr in np.unique(col.lin_shr[lin]):
        s_y[lin & (col.lin_shr == shr)] = pow(inv2, int(shr), hl.R_MOD)
    h_pre = [fr_of_values(c) for c in (col.q_xor, col.q_and, col.q_lin, col.cx, s_x, col.cy, s_y)]
    d_wit = [fr_of_u64(c) for c in (col.x, col.y, col.o, col.d_xor, col.a_xor, col.d_and, col.a_and)]
    d_perm = [fr_of_u64((col.perm_col[c].astype(np.uint64) << np.uint64(k)) + col.perm_row[c].astype(np.uint64)) for c in range(3)]
    unit_bits = 2 * ub
    t_xor, t_and = hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 1, unit_bits), hl.LassoTable.bitwise(hl.SUBTABLE_AND, 1, unit_bits)

    circ = SyntheticCircuit()
    circ.k, circ.columns, circ.program, circ.states, circ.outputs = k, col, prog, states, col.outputs
    circ.num_copies = int(sum(((col.perm_row[c] != np.arange(size)) | (col.perm_col[c] != c)).sum() for c in range(3)))
    circ.num_lookups = int(col.q_xor.sum() + col.q_and.sum())
    circ.num_permutations, circ.rows_per_permutation, circ.lane_bits, circ.unit_bits = B, prog.num_rows, w, ub
    circ.tables = (t_xor, t_and)
    circ.info = hp.keccak_circuit_info(k, [[]] * 7, [[(8, 1)], [(9, 1)], [(10, 1)]], t_xor, t_and)
    circ.h_preprocess = h_pre
    circ.d_preprocess = [up(a) for a in h_pre]
    circ.d_permutation = d_perm
    circ.d_witness = d_wit
    circ.h_witness = [down(d) for d in d_wit]
    circ.instances = [[]]
    return circ


def prover_param(pcs_pp, circuit, pcs_vp=None):
    """HyperPlonk.preprocess for a SyntheticCircuit whose polys already live on the device -> pp or (pp, vp)"""
    import halo2_lasso_amd as hl
    pp = hp.HyperPlonkProverParam()
    pp.pcs, pp.num_vars, pp.info = pcs_pp, circuit.k, circuit.info
    pp.preprocess_polys, pp.permutation_polys = circuit.d_preprocess, circuit.d_permutation
    pp.num_permutation_z_polys, pp.expression = hp.compose(circuit.info)
    if pcs_vp is None:
        return pp
    vp = hp.HyperPlonkVerifierParam()
    vp.pcs, vp.num_vars, vp.info = pcs_vp, circuit.k, circuit.info
    vp.num_permutation_z_polys, vp.expression = pp.num_permutation_z_polys, pp.expression
    pcs = hp._pcs_of(pcs_pp)
    vp.preprocess_comms = pcs.batch_commit(pcs_pp, pp.preprocess_polys)
    vp.permutation_comms = pcs.batch_commit(pcs_pp, pp.permutation_polys)
    return pp, vp
