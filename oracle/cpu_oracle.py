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
    _chk(lib().orc_msm(fr_bytes(scalars), bases_bytes, C.c_size_t(len(scalars)), out))
    return g1_point(out.raw)


def commit(srs, srs_nv, poly):
    out = C.create_string_buffer(64)
    nv = len(poly).bit_length() - 1
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
    B, nv = len(ps), len(ps[0]).bit_length() - 1
    pa, k1 = _ptrs([fr_bytes(p) for p in ps])
    qa, k2 = _ptrs([fr_bytes(q) for q in qs])
    px, qx, x = (C.create_string_buffer(32 * B), C.create_string_buffer(32 * B), C.create_string_buffer(32 * nv))
    _chk(lib().orc_frac_gkr_prove(tr.h, C.c_size_t(B), C.c_size_t(nv), pa, qa, px, qx, x))
    return fr_ints(px.raw), fr_ints(qx.raw), fr_ints(x.raw)


def grand_product_prove(tr, leaves):
    B = len(leaves)
    nvs = [len(v).bit_length() - 1 for v in leaves]
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
