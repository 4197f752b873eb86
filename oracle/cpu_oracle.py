"""ctypes wrapper of the C++ CPU oracle (oracle/cpu/oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Field elements / points cross as Montgomery bytes (the C-ABI's layout), so buffers downloaded from the
GPU (tables, SRS) can be handed over unchanged.
"""
import ctypes as C
import os

from .pyref.field import R_MOD, Q_MOD, MONT_R

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle_cpu.so")
_lib = None
_RI = pow(MONT_R, -1, R_MOD)
_QI = pow(MONT_R, -1, Q_MOD)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s missing: run `make -C oracle/cpu` (or __graft_entry__.build())" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.orc_last_error.restype = C.c_char_p
        l.orc_tr_new.restype = C.c_void_p
        l.orc_tr_proof.restype = C.c_size_t
        for name in ("orc_tr_free", "orc_tr_write_fe", "orc_tr_common_fe", "orc_tr_squeeze"):
            getattr(l, name).restype = None
        _lib = l
    return _lib


def fr_bytes(xs):
    """ints -> Montgomery bytes; byte strings (already Montgomery, e.g. downloaded from the GPU) pass through"""
    if isinstance(xs, (bytes, bytearray)):
        return bytes(xs)
    return b"".join((x % R_MOD * MONT_R % R_MOD).to_bytes(32, "little") for x in xs)


def fr_ints(b):
    return [int.from_bytes(b[i:i + 32], "little") * _RI % R_MOD for i in range(0, len(b), 32)]


def g1_bytes(pts):
    return b"".join(bytes(64) if p is None else b"".join((v * MONT_R % Q_MOD).to_bytes(32, "little") for v in p)
                    for p in pts)


def g1_point(b):
    x = int.from_bytes(b[:32], "little") * _QI % Q_MOD
    y = int.from_bytes(b[32:64], "little") * _QI % Q_MOD
    return None if x == 0 and y == 0 else (x, y)


def _count(p):
    """number of field elements of an int list or a Montgomery byte string"""
    return len(p) // 32 if isinstance(p, (bytes, bytearray)) else len(p)


def _chk(rc):
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())


class Transcript:
    def __init__(self):
        self.h = C.c_void_p(lib().orc_tr_new())

    def __del__(self):
        try:
            lib().orc_tr_free(self.h)
        except Exception:
            pass

    def into_proof(self):
        p = C.POINTER(C.c_uint8)()
        n = lib().orc_tr_proof(self.h, C.byref(p))
        return C.string_at(p, n)

    def write_field_element(self, x):
        lib().orc_tr_write_fe(self.h, fr_bytes([x]))

    def write_field_elements(self, xs):
        for x in xs:
            self.write_field_element(x)

    def common_field_element(self, x):
        lib().orc_tr_common_fe(self.h, fr_bytes([x]))

    def squeeze_challenge(self):
        out = C.create_string_buffer(32)
        lib().orc_tr_squeeze(self.h, out)
        return fr_ints(out.raw)[0]

    def squeeze_challenges(self, n):
        return [self.squeeze_challenge() for _ in range(n)]

    def write_commitment(self, pt):
        _chk(lib().orc_tr_write_comm(self.h, g1_bytes([pt])))

    def write_commitments(self, pts):
        for p in pts:
            self.write_commitment(p)


def _ptrs(bufs):
    keep = [C.create_string_buffer(b, len(b)) if isinstance(b, (bytes, bytearray)) else b for b in bufs]
    arr = (C.c_void_p * max(len(keep), 1))(*[C.cast(k, C.c_void_p).value for k in keep])
    return arr, keep


def set_threads(n):
    lib().orc_set_threads(n)


def num_threads():
    return lib().orc_num_threads()


def setup(ss):
    """-> flat SRS bytes (level k at offset 2^k - 1), Montgomery affine points"""
    n = len(ss)
    out = C.create_string_buffer(64 * ((2 << n) - 1))
    _chk(lib().orc_setup(fr_bytes(ss), C.c_size_t(n), out))
    return out.raw


def msm(scalars, bases_bytes):
    out = C.create_string_buffer(64)
    _chk(lib().orc_msm(fr_bytes(scalars), bases_bytes, C.c_size_t(_count(scalars)), out))
    return g1_point(out.raw)


def commit(srs, srs_nv, poly):
    out = C.create_string_buffer(64)
    nv = _count(poly).bit_length() - 1
    _chk(lib().orc_commit(srs, C.c_size_t(srs_nv), fr_bytes(poly), C.c_size_t(nv), out))
    return g1_point(out.raw)


def open_(tr, srs, srs_nv, poly, point):
    out = C.create_string_buffer(32)
    _chk(lib().orc_open(tr.h, srs, C.c_size_t(srs_nv), fr_bytes(poly), C.c_size_t(len(point)), fr_bytes(point), out))
    return fr_ints(out.raw)[0]


def batch_open(tr, srs, srs_nv, nv, polys, points, evals_struct_array, num_evals):
    arr, keep = _ptrs([fr_bytes(p) for p in polys])
    flat = fr_bytes([v for p in points for v in p])
    _chk(lib().orc_batch_open(tr.h, srs, C.c_size_t(srs_nv), C.c_size_t(nv), arr, C.c_size_t(len(polys)), flat,
                              C.c_size_t(len(points)), evals_struct_array, C.c_size_t(num_evals)))


def sumcheck_prove(tr, kind, nv, sop_struct, polys, ys, claim):
    arr, keep = _ptrs([fr_bytes(p) for p in polys])
    x, ev = C.create_string_buffer(32 * nv), C.create_string_buffer(32 * max(len(polys), 1))
    _chk(lib().orc_sumcheck_prove(tr.h, kind, C.c_size_t(nv), C.byref(sop_struct), arr, C.c_size_t(len(polys)),
                                  fr_bytes([v for y in ys for v in y]), C.c_size_t(len(ys)), fr_bytes([claim]), x, ev))
    return fr_ints(x.raw), fr_ints(ev.raw)[:len(polys)]


def frac_gkr_prove(tr, ps, qs):
    B, nv = len(ps), _count(ps[0]).bit_length() - 1
    pa, k1 = _ptrs([fr_bytes(p) for p in ps])
    qa, k2 = _ptrs([fr_bytes(q) for q in qs])
    px, qx, x = (C.create_string_buffer(32 * B), C.create_string_buffer(32 * B), C.create_string_buffer(32 * nv))
    _chk(lib().orc_frac_gkr_prove(tr.h, C.c_size_t(B), C.c_size_t(nv), pa, qa, px, qx, x))
    return fr_ints(px.raw), fr_ints(qx.raw), fr_ints(x.raw)


def grand_product_prove(tr, leaves):
    B = len(leaves)
    nvs = [_count(v).bit_length() - 1 for v in leaves]
    la, keep = _ptrs([fr_bytes(v) for v in leaves])
    roots, claims = C.create_string_buffer(32 * B), C.create_string_buffer(32 * B)
    pts = C.create_string_buffer(32 * sum(nvs))
    _chk(lib().orc_grand_product_prove(tr.h, C.c_size_t(B), la, (C.c_size_t * B)(*nvs), roots, claims, pts))
    flat, out, off = fr_ints(pts.raw), [], 0
    for nv in nvs:
        out.append(flat[off:off + nv])
        off += nv
    return fr_ints(roots.raw), list(zip(fr_ints(claims.raw), out))


def lasso_prove(tr, srs, srs_nv, table_struct, n, dims_u32_bytes):
    """dims_u32_bytes: list of bytes objects / buffers holding u32[2^n]."""
    arr, keep = _ptrs(dims_u32_bytes)
    _chk(lib().orc_lasso_prove(tr.h, srs, C.c_size_t(srs_nv), C.byref(table_struct), C.c_size_t(n), arr))


# ------------------------------------------------------------------ HyperPlonk (oracle/cpu/oracle.cpp: hyperplonk_prove)
class _ExprNode(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", C.c_int32), ("b", C.c_int32), ("reserved", C.c_uint32), ("scalar", C.c_uint8 * 32)]


class _Expr(C.Structure):
    _fields_ = [("nodes", C.POINTER(_ExprNode)), ("n", C.c_size_t)]


class _HpLookup(C.Structure):
    _fields_ = [("inputs", C.POINTER(_Expr)), ("tables", C.POINTER(_Expr)), ("width", C.c_size_t)]


class _HpParam(C.Structure):
    _fields_ = [("num_vars", C.c_size_t),
                ("num_instance_polys", C.c_size_t), ("num_instances", C.POINTER(C.c_size_t)),
                ("num_preprocess_polys", C.c_size_t), ("preprocess_polys", C.POINTER(C.c_void_p)),
                ("num_witness_polys", C.c_size_t), ("num_challenges", C.c_size_t),
                ("num_lookups", C.c_size_t), ("lookups", C.POINTER(_HpLookup)),
                ("num_permutation_polys", C.c_size_t), ("permutation_poly_index", C.POINTER(C.c_size_t)),
                ("permutation_polys", C.POINTER(C.c_void_p)),
                ("num_permutation_z_polys", C.c_size_t),
                ("expression", _Expr),
                ("num_lasso_lookups", C.c_size_t), ("lasso_lookups", C.c_void_p)]


def flatten_expression(e):
    """pyref expression -> node tuples (op, a, b, scalar) in topological order, root last; DistributePowers lowered as
    Expression::evaluate does (expression.rs:155-167)"""
    from .pyref import expression as ex
    nodes = []

    def emit(op, a=0, b=0, scalar=0):
        nodes.append((op, a, b, scalar % R_MOD))
        return len(nodes) - 1

    def go(x):
        if isinstance(x, ex.Constant):
            return emit(0, scalar=x.v)
        if isinstance(x, ex.Identity):
            return emit(1)
        if isinstance(x, ex.Lagrange):
            return emit(2, x.i)
        if isinstance(x, ex.EqXY):
            return emit(3, x.idx)
        if isinstance(x, ex.Poly):
            return emit(4, x.idx, x.rotation)
        if isinstance(x, ex.Challenge):
            return emit(5, x.idx)
        if isinstance(x, ex.Negated):
            return emit(6, go(x.a))
        if isinstance(x, ex.Sum):
            a = go(x.a)
            return emit(7, a, go(x.b))
        if isinstance(x, ex.Product):
            a = go(x.a)
            return emit(8, a, go(x.b))
        if isinstance(x, ex.Scaled):
            return emit(9, go(x.a), scalar=x.s)
        if isinstance(x, ex.DistributePowers):
            if len(x.exprs) == 1:
                return go(x.exprs[0])
            base, acc = go(x.base), go(x.exprs[0])
            power = base
            for k, sub in enumerate(x.exprs[1:]):
                if k:
                    power = emit(8, power, base)
                acc = emit(7, acc, emit(8, power, go(sub)))
            return acc
        raise TypeError(x)

    go(e)
    return nodes


def _c_expr(nodes, keep):
    arr = (_ExprNode * len(nodes))()
    for k, (op, a, b, scalar) in enumerate(nodes):
        arr[k].op, arr[k].a, arr[k].b = op, a, b
        C.memmove(C.byref(arr[k], _ExprNode.scalar.offset), fr_bytes([scalar]), 32)
    keep.append(arr)
    e = _Expr()
    e.nodes, e.n = C.cast(arr, C.POINTER(_ExprNode)), len(nodes)
    return e


def _poly_bytes(p):
    return p if isinstance(p, (bytes, bytearray)) else fr_bytes(p)


def hyperplonk_prove(tr, srs, srs_nv, num_vars, num_instances, preprocess_polys, num_witness_polys, num_challenges,
                     lookups, permutation_poly_index, permutation_polys, num_permutation_z_polys, expression,
                     instances, witness, lasso_lookups=()):
    """Polys are int lists or Montgomery byte strings (e.g. downloaded from the GPU); expressions are node lists from
    `flatten_expression`; lookups: list of lists of (input nodes, table nodes); lasso_lookups: list of
    (lh_lasso_table-layout ctypes struct, output_poly, chunk_polys) - lookups proven by the Lasso argument
    (oracle/pyref/hyperplonk.py LassoLookup)."""
    keep = []
    pp = _HpParam()
    pp.num_vars = num_vars
    pp.num_instance_polys = len(num_instances)
    ni = (C.c_size_t * max(len(num_instances), 1))(*num_instances)
    pp.num_instances = ni
    pre, k1 = _ptrs([_poly_bytes(p) for p in preprocess_polys])
    pp.num_preprocess_polys, pp.preprocess_polys = len(preprocess_polys), C.cast(pre, C.POINTER(C.c_void_p))
    pp.num_witness_polys, pp.num_challenges = num_witness_polys, num_challenges
    lk = (_HpLookup * max(len(lookups), 1))()
    for i, lookup in enumerate(lookups):
        ins = (_Expr * len(lookup))(*[_c_expr(a, keep) for a, _ in lookup])
        tabs = (_Expr * len(lookup))(*[_c_expr(b, keep) for _, b in lookup])
        keep += [ins, tabs]
        lk[i].inputs, lk[i].tables, lk[i].width = ins, tabs, len(lookup)
    pp.num_lookups, pp.lookups = len(lookups), lk
    pidx = (C.c_size_t * max(len(permutation_poly_index), 1))(*permutation_poly_index)
    perm, k2 = _ptrs([_poly_bytes(p) for p in permutation_polys])
    pp.num_permutation_polys, pp.permutation_poly_index = len(permutation_polys), pidx
    pp.permutation_polys = C.cast(perm, C.POINTER(C.c_void_p))
    pp.num_permutation_z_polys = num_permutation_z_polys
    pp.expression = _c_expr(expression, keep)
    if lasso_lookups:
        tsize = C.sizeof(lasso_lookups[0][0])
        stride = tsize + 8 + 8 * 8  # table | output_poly | chunk_polys[8]: the layout of lh_hp_lasso_lookup
        buf = C.create_string_buffer(stride * len(lasso_lookups))
        for k, (table, out_poly, chunk_polys) in enumerate(lasso_lookups):
            C.memmove(C.addressof(buf) + k * stride, C.byref(table), tsize)
            tail = (C.c_size_t * 9)(out_poly, *(list(chunk_polys) + [0] * (8 - len(chunk_polys))))
            C.memmove(C.addressof(buf) + k * stride + tsize, tail, 72)
        keep.append(buf)
        pp.num_lasso_lookups, pp.lasso_lookups = len(lasso_lookups), C.cast(buf, C.c_void_p)
    inst, k3 = _ptrs([fr_bytes(i) if len(i) else bytes(32) for i in instances])
    wit, k4 = _ptrs([_poly_bytes(w) for w in witness])
    _chk(lib().orc_hyperplonk_prove(tr.h, srs, C.c_size_t(srs_nv), C.byref(pp), inst, wit))


# ------------------------------------------------------------------ Zeromorph over univariate KZG (oracle.cpp: zm_*)
def usetup(s, poly_size):
    """-> powers_of_s_g1 as Montgomery bytes (64 per point)"""
    out = C.create_string_buffer(64 * poly_size)
    _chk(lib().orc_usetup(fr_bytes([s]), C.c_size_t(poly_size), out))
    return out.raw


def zm_commit(powers, poly_size, poly):
    out = C.create_string_buffer(64)
    nv = _count(poly).bit_length() - 1
    _chk(lib().orc_zm_commit(powers, C.c_size_t(len(powers) // 64), C.c_size_t(poly_size), fr_bytes(poly), C.c_size_t(nv), out))
    return g1_point(out.raw)


def zm_open(tr, powers, poly_size, poly, point):
    _chk(lib().orc_zm_open(tr.h, powers, C.c_size_t(len(powers) // 64), C.c_size_t(poly_size), fr_bytes(poly),
                           C.c_size_t(len(point)), fr_bytes(point)))


def zm_batch_open(tr, powers, poly_size, nv, polys, points, evals_struct_array, num_evals):
    arr, keep = _ptrs([fr_bytes(p) for p in polys])
    flat = fr_bytes([v for p in points for v in p])
    _chk(lib().orc_zm_batch_open(tr.h, powers, C.c_size_t(len(powers) // 64), C.c_size_t(poly_size), C.c_size_t(nv), arr,
                                 C.c_size_t(len(polys)), flat, C.c_size_t(len(points)), evals_struct_array,
                                 C.c_size_t(num_evals)))


def lasso_prove_zm(tr, powers, poly_size, table_struct, n, dims_u32_bytes):
    arr, keep = _ptrs(dims_u32_bytes)
    _chk(lib().orc_lasso_prove_zm(tr.h, powers, C.c_size_t(len(powers) // 64), C.c_size_t(poly_size),
                                  C.byref(table_struct), C.c_size_t(n), arr))
