"""halo2-lasso_amd: MI355X-native prover hot path of DoHoonKim8/halo2-lasso.

Host-side mirror (Python, over the C-ABI in include/lasso_hip.h) of the reference's interface for
the path: names and argument meaning follow plonkish_backend --
  util::transcript::Keccak256Transcript, poly::multilinear::MultilinearPolynomial,
  piop::sum_check::{ClassicSumCheck, EvaluationsProver, CoefficientsProver},
  piop::gkr::prove_fractional_sum_check, util::arithmetic::variable_base_msm,
  pcs::multilinear::MultilinearKzg, and the Lasso prover (no reference code).
All compute happens in liblasso_hip.so on the GPU; this module only marshals.
"""
import ctypes as C

from . import _ffi
from ._ffi import (lh_fr, lh_g1, lh_sop, lh_evaluation, lh_lasso_table, lh_transcript,
                   LH_SC_EVALUATIONS, LH_SC_COEFFICIENTS)

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
_MONT = 1 << 256
_MONT_INV_R = pow(_MONT, -1, R_MOD)
_MONT_INV_Q = pow(_MONT, -1, Q_MOD)

SUBTABLE_IDENTITY, SUBTABLE_AND, SUBTABLE_XOR = 0, 1, 2


# ------------------------------------------------------------------ errors (plonkish_backend/src/lib.rs:12-20)
class Error(Exception):
    code = None


class InvalidSumcheck(Error):
    code = _ffi.LH_ERR_INVALID_SUMCHECK


class InvalidPcsParam(Error):
    code = _ffi.LH_ERR_INVALID_PCS_PARAM


class InvalidPcsOpen(Error):
    code = _ffi.LH_ERR_INVALID_PCS_OPEN


class InvalidSnark(Error):
    code = _ffi.LH_ERR_INVALID_SNARK


class Serialization(Error):
    code = _ffi.LH_ERR_SERIALIZATION


class TranscriptError(Error):
    code = _ffi.LH_ERR_TRANSCRIPT


class DeviceError(Error):
    code = _ffi.LH_ERR_DEVICE


class ArgumentError(Error):  # an assert!/panic in the reference
    code = _ffi.LH_ERR_ARG


_ERRORS = {e.code: e for e in (InvalidSumcheck, InvalidPcsParam, InvalidPcsOpen, InvalidSnark,
                               Serialization, TranscriptError, DeviceError, ArgumentError)}


def _check(rc):
    if rc != 0:
        msg = _ffi.load().lh_last_error()
        raise _ERRORS.get(rc, Error)((msg or b"").decode() or "lh_status %d" % rc)


# ------------------------------------------------------------------ marshalling
def fr_to_bytes(x):
    """Python int -> 32 bytes of a Montgomery `bn256::Fr`."""
    return (x % R_MOD * _MONT % R_MOD).to_bytes(32, "little")


def fr_from_bytes(b):
    return int.from_bytes(bytes(b), "little") * _MONT_INV_R % R_MOD


def frs_to_bytes(xs):
    return b"".join(fr_to_bytes(x) for x in xs)


def frs_from_bytes(b):
    return [fr_from_bytes(b[i:i + 32]) for i in range(0, len(b), 32)]


def g1_to_bytes(pt):
    """(x, y) ints or None (identity = (0,0)) -> 64 bytes of a Montgomery `bn256::G1Affine`."""
    if pt is None:
        return bytes(64)
    return b"".join((v % Q_MOD * _MONT % Q_MOD).to_bytes(32, "little") for v in pt)


def g1_from_bytes(b):
    b = bytes(b)
    x = int.from_bytes(b[:32], "little") * _MONT_INV_Q % Q_MOD
    y = int.from_bytes(b[32:], "little") * _MONT_INV_Q % Q_MOD
    return None if x == 0 and y == 0 else (x, y)


def g2_to_bytes(pt):
    """((x0, x1), (y0, y1)) or None -> 128 bytes of a Montgomery `bn256::G2Affine` (x.c0 | x.c1 | y.c0 | y.c1)."""
    if pt is None:
        return bytes(128)
    return b"".join((v % Q_MOD * _MONT % Q_MOD).to_bytes(32, "little") for c in pt for v in c)


def g2_from_bytes(b):
    b = bytes(b)
    v = [int.from_bytes(b[32 * i:32 * i + 32], "little") * _MONT_INV_Q % Q_MOD for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def _fr_array(xs):
    arr = (lh_fr * max(len(xs), 1))()
    C.memmove(arr, frs_to_bytes(xs), 32 * len(xs))
    return arr


def _fr_list(arr, n):
    return frs_from_bytes(C.string_at(arr, 32 * n))


# ------------------------------------------------------------------ context / device memory
def parse_cpulist(text):
    """the kernel's CPU list format ("0-63,128-191") as a set"""
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


class Context:
    """One per process per GPU (`lh_ctx`)."""

    def __init__(self, device_id=0):
        self.lib = _ffi.load()
        h = C.c_void_p()
        _check(self.lib.lh_ctx_create(device_id, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.lh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self.lib.lh_ctx_sync(self.h))

    def host_cpus(self):
        """(PCI address of the ctx's device, the CPUs on its NUMA node as a set) - lh_ctx_host_cpus; the set is empty where
        the system does not say"""
        bus, cpus = C.create_string_buffer(64), C.create_string_buffer(4096)
        _check(self.lib.lh_ctx_host_cpus(self.h, bus, 64, cpus, 4096))
        return bus.value.decode(), parse_cpulist(cpus.value.decode())

    def bind_host(self):
        """Bind the CALLING thread to the CPUs next to the ctx's device - what a one-process-per-GPU deployment does with
        numactl.  Threads the library starts AFTERWARDS from this thread inherit the mask (its host pool, which also sizes
        itself by the mask; the helper ctx's worker); threads that exist already keep theirs, and threads created while
        bound stay bound when the caller restores its own mask: call this BEFORE the first proof of the process.  Returns
        the previous affinity mask (give it to os.sched_setaffinity to undo), or None when nothing was changed: no such
        list, or none of its CPUs is in the present mask."""
        import os
        _, local = self.host_cpus()
        before = os.sched_getaffinity(0)
        want = local & before
        if not want or want == before:
            return None
        os.sched_setaffinity(0, want)
        return before

    @property
    def stream(self):
        return self.lib.lh_ctx_stream(self.h)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, data):
        buf = DeviceBuffer(self, len(data))
        buf.write(data)
        return buf


class DeviceBuffer:
    def __init__(self, ctx, nbytes, ptr=None):
        self.ctx, self.nbytes, self.owned = ctx, nbytes, ptr is None
        if ptr is None:
            p = C.c_void_p()
            _check(ctx.lib.lh_alloc(ctx.h, nbytes, C.byref(p)))
            ptr = p.value
        self.ptr = ptr

    def write(self, data, offset=0):
        data = bytes(data)
        assert offset + len(data) <= self.nbytes
        _check(self.ctx.lib.lh_upload(self.ctx.h, self.ptr + offset, data, len(data)))

    def read(self, nbytes=None, offset=0):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = C.create_string_buffer(nbytes)
        _check(self.ctx.lib.lh_download(self.ctx.h, out, self.ptr + offset, nbytes))
        return out.raw

    def free(self):
        if self.owned and self.ptr and self.ctx.h:
            self.ctx.lib.lh_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr_array(bufs):
    arr = (C.c_void_p * max(len(bufs), 1))()
    for i, b in enumerate(bufs):
        arr[i] = b.ptr if hasattr(b, "ptr") else b
    return arr


# ------------------------------------------------------------------ transcript
class Keccak256Transcript:
    """util::transcript::Keccak256Transcript<Cursor<Vec<u8>>> living in the library: `Keccak256Transcript()`
    writes (InMemoryTranscript::new), `Keccak256Transcript.from_proof(bytes)` reads (transcript.rs:110-123)."""

    def __init__(self, proof=None):
        self.lib = _ffi.load()
        p = C.POINTER(lh_transcript)()
        if proof is None:
            _check(self.lib.lh_keccak_transcript_new(C.byref(p)))
        else:
            _check(self.lib.lh_keccak_transcript_from_proof(bytes(proof), len(proof), C.byref(p)))
        self.p = p

    @classmethod
    def from_proof(cls, proof):
        return cls(proof)

    def read_field_element(self):
        out = lh_fr()
        _check(self._vt().read_field_element(self._vt().user, C.byref(out)))
        return fr_from_bytes(bytes(out))

    def read_field_elements(self, n):
        return [self.read_field_element() for _ in range(n)]

    def read_commitment(self):
        out = lh_g1()
        _check(self._vt().read_commitment(self._vt().user, C.byref(out)))
        return g1_from_bytes(bytes(out))

    def read_commitments(self, n):
        return [self.read_commitment() for _ in range(n)]

    def remaining(self):
        n = C.c_size_t()
        _check(self.lib.lh_keccak_transcript_remaining(self.p, C.byref(n)))
        return n.value

    def _vt(self):
        return self.p.contents

    def write_field_element(self, fe):
        _check(self._vt().write_field_element(self._vt().user, C.byref(_fr_array([fe])[0])))

    def write_field_elements(self, fes):
        for fe in fes:
            self.write_field_element(fe)

    def common_field_element(self, fe):
        _check(self._vt().common_field_element(self._vt().user, C.byref(_fr_array([fe])[0])))

    def squeeze_challenge(self):
        out = lh_fr()
        _check(self._vt().squeeze_challenge(self._vt().user, C.byref(out)))
        return fr_from_bytes(bytes(out))

    def squeeze_challenges(self, n):
        return [self.squeeze_challenge() for _ in range(n)]

    def write_commitment(self, pt):
        g = lh_g1()
        C.memmove(C.byref(g), g1_to_bytes(pt), 64)
        _check(self._vt().write_commitment(self._vt().user, C.byref(g)))

    def write_commitments(self, pts):
        for pt in pts:
            self.write_commitment(pt)

    def into_proof(self):
        ptr, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        _check(self.lib.lh_keccak_transcript_proof(self.p, C.byref(ptr), C.byref(n)))
        return C.string_at(ptr, n.value)

    def __del__(self):
        try:
            if self.p:
                self.lib.lh_keccak_transcript_free(self.p)
                self.p = None
        except Exception:
            pass


# ------------------------------------------------------------------ poly::multilinear
class MultilinearPolynomial:
    """Device-resident evaluation table of 2^num_vars Fr (poly/multilinear.rs:20-24)."""

    def __init__(self, ctx, buf, num_vars):
        self.ctx, self.buf, self.num_vars = ctx, buf, num_vars

    @property
    def ptr(self):
        return self.buf.ptr

    def __len__(self):
        return 1 << self.num_vars

    @classmethod
    def new(cls, ctx, evals):
        n = len(evals)
        assert n and n & (n - 1) == 0
        return cls(ctx, ctx.upload(frs_to_bytes(evals)), n.bit_length() - 1)

    @classmethod
    def from_u32(cls, ctx, values):
        import array
        n = len(values)
        assert n and n & (n - 1) == 0
        src = ctx.upload(array.array("I", values).tobytes())
        out = ctx.alloc(32 * n)
        _check(ctx.lib.lh_fr_from_u32(ctx.h, src.ptr, n, out.ptr))
        ctx.sync()
        return cls(ctx, out, n.bit_length() - 1)

    @classmethod
    def eq_xy(cls, ctx, y):
        out = ctx.alloc(32 << len(y))
        _check(ctx.lib.lh_eq_xy(ctx.h, _fr_array(y), len(y), out.ptr))
        return cls(ctx, out, len(y))

    def evals(self):
        return frs_from_bytes(self.buf.read(32 << self.num_vars))

    def fix_var(self, x):
        out = self.ctx.alloc(32 << (self.num_vars - 1))
        _check(self.ctx.lib.lh_fix_var(self.ctx.h, self.ptr, 1 << self.num_vars, _fr_array([x]), out.ptr))
        return MultilinearPolynomial(self.ctx, out, self.num_vars - 1)

    def evaluate(self, x):
        return evaluate_polys(self.ctx, [self], x)[0]


def evaluate_polys(ctx, polys, x):
    out = (lh_fr * len(polys))()
    _check(ctx.lib.lh_evaluate(ctx.h, _ptr_array(polys), len(polys), len(x), _fr_array(x), out))
    return _fr_list(out, len(polys))


# ------------------------------------------------------------------ piop::sum_check
EvaluationsProver = LH_SC_EVALUATIONS
CoefficientsProver = LH_SC_COEFFICIENTS


class SumOfProducts:
    """[eq_xy(ys[global_eq]) *] sum_m coeff_m * prod_k table[id]; ids < len(polys) are polys,
    ids >= len(polys) are eq_xy(ys[id - len(polys)]) -- the Expression shapes of
    fractional_sum_check.rs:272-281 and pcs/multilinear.rs:182-190."""

    def __init__(self, terms, global_eq=-1):
        self.terms, self.global_eq = [(c % R_MOD, list(f)) for c, f in terms], global_eq

    def to_c(self):
        s = lh_sop()
        s.num_terms, s.global_eq = len(self.terms), self.global_eq
        for m, (coeff, facs) in enumerate(self.terms):
            C.memmove(C.byref(s.coeff[m]), fr_to_bytes(coeff), 32)
            s.num_factors[m] = len(facs)
            for k, f in enumerate(facs):
                s.factor[m][k] = f
        return s


class ClassicSumCheck:
    @staticmethod
    def prove(ctx, prover, num_vars, expression, polys, ys, sum_, transcript):
        """ClassicSumCheck::<P>::prove (classic.rs:208-240) -> (challenges, evals)."""
        if len(expression.terms) > _ffi.LH_SC_MAX_TERMS:
            raise ArgumentError("too many terms")
        ys_flat = [v for y in ys for v in y]
        ch = (lh_fr * max(num_vars, 1))()
        ev = (lh_fr * max(len(polys), 1))()
        sop = expression.to_c()
        _check(ctx.lib.lh_sumcheck_prove(ctx.h, prover, num_vars, C.byref(sop), _ptr_array(polys), len(polys),
                                         _fr_array(ys_flat), len(ys), _fr_array([sum_]), transcript.p, ch, ev))
        return _fr_list(ch, num_vars), _fr_list(ev, len(polys))


def sum_check_prove_expression(ctx, num_vars, expression, polys, challenges, ys, sum_, transcript):
    """ClassicSumCheck::<EvaluationsProver>::prove over VirtualPolynomial{expression, polys, challenges, ys}
    (piop/sum_check.rs:16-37, classic.rs:208-240) for a general `expression.Expression`
    (rotations, Identity, Lagrange) -> (challenges x, evals of every poly at x)."""
    ce, keep = expression.to_c()
    ys_flat = [v for y in ys for v in y]
    ch = (lh_fr * max(num_vars, 1))()
    ev = (lh_fr * max(len(polys), 1))()
    _check(ctx.lib.lh_sumcheck_prove_expr(ctx.h, num_vars, C.byref(ce), _ptr_array(polys), len(polys),
                                          _fr_array(challenges), len(challenges), _fr_array(ys_flat), len(ys),
                                          _fr_array([sum_]), transcript.p, ch, ev))
    return _fr_list(ch, num_vars), _fr_list(ev, len(polys))


# ------------------------------------------------------------------ piop::gkr
def prove_fractional_sum_check(ctx, claimed_p_0s, claimed_q_0s, ps, qs, transcript):
    """fractional_sum_check.rs:89-190 -> (p_xs, q_xs, x)."""
    B = len(ps)
    if not (B == len(qs) == len(claimed_p_0s) == len(claimed_q_0s)):
        raise ArgumentError("length mismatch")  # :105-107 assert_eq
    num_vars = ps[0].num_vars if B else 0
    for p in list(ps) + list(qs):
        if p.num_vars != num_vars:
            raise ArgumentError("num_vars mismatch")  # :108-110

    def opt(claims):
        keep = [_fr_array([c]) if c is not None else None for c in claims]
        arr = (C.POINTER(lh_fr) * max(B, 1))()
        for i, k in enumerate(keep):
            arr[i] = C.cast(k, C.POINTER(lh_fr)) if k is not None else None
        return arr, keep

    cp, keep_p = opt(claimed_p_0s)
    cq, keep_q = opt(claimed_q_0s)
    p_xs, q_xs, x = (lh_fr * max(B, 1))(), (lh_fr * max(B, 1))(), (lh_fr * max(num_vars, 1))()
    _check(ctx.lib.lh_gkr_fractional_prove(ctx.h, B, num_vars, cp, cq, _ptr_array(ps), _ptr_array(qs),
                                           transcript.p, p_xs, q_xs, x))
    return _fr_list(p_xs, B), _fr_list(q_xs, B), _fr_list(x, num_vars)


def prove_grand_product(ctx, leaves, transcript):
    """Batched product-tree GKR (Lasso memory check) -> (roots, [(claim, point)])."""
    B = len(leaves)
    nv = (C.c_size_t * max(B, 1))(*[p.num_vars for p in leaves])
    total = sum(p.num_vars for p in leaves)
    roots, claims, points = (lh_fr * max(B, 1))(), (lh_fr * max(B, 1))(), (lh_fr * max(total, 1))()
    _check(ctx.lib.lh_grand_product_prove(ctx.h, B, _ptr_array(leaves), nv, transcript.p, roots, claims, points))
    pts, flat, off = [], _fr_list(points, total), 0
    for p in leaves:
        pts.append(flat[off:off + p.num_vars])
        off += p.num_vars
    return _fr_list(roots, B), list(zip(_fr_list(claims, B), pts))


# ------------------------------------------------------------------ util::arithmetic::msm
def variable_base_msm(ctx, scalars, bases, n=None):
    """msm.rs:84-181 on device buffers -> affine (x, y) ints or None for the identity."""
    n = len(scalars) if n is None else n
    out = lh_g1()
    _check(ctx.lib.lh_msm(ctx.h, scalars.ptr, bases.ptr, n, C.byref(out)))
    return g1_from_bytes(bytes(out))


def variable_base_msm_u32(ctx, scalars, bases, n):
    out = lh_g1()
    _check(ctx.lib.lh_msm_u32(ctx.h, scalars.ptr, bases.ptr, n, C.byref(out)))
    return g1_from_bytes(bytes(out))


# ------------------------------------------------------------------ pcs::multilinear::kzg
class Evaluation:
    """pcs.rs:132-155"""

    def __init__(self, poly, point, value):
        self.poly, self.point, self.value = poly, point, value % R_MOD


class MultilinearKzgParams:
    def __init__(self, ctx, handle, borrowed=False):
        self.ctx, self.h, self.borrowed = ctx, handle, borrowed

    def view(self, ctx):
        """The same SRS (device memory, owned by this object - keep it alive) for use through another ctx of the same
        device: a second proof in flight on its own stream (bench.py two_proofs_in_flight / sharded_two_in_flight)."""
        return MultilinearKzgParams(ctx, self.h, borrowed=True)

    @property
    def num_vars(self):
        return self.ctx.lib.lh_srs_num_vars(self.h)

    def eqs_bytes(self):
        """the flat SRS as the device holds it: level k (2^k points of 64 bytes) at offset 2^k - 1"""
        total = (2 << self.num_vars) - 1
        out = C.create_string_buffer(64 * total)
        _check(self.ctx.lib.lh_srs_download(self.ctx.h, self.h, out))
        return out.raw

    def eqs(self):
        """All levels as lists of affine points (level k has 2^k)."""
        n = self.num_vars
        total = (2 << n) - 1
        out = C.create_string_buffer(64 * total)
        _check(self.ctx.lib.lh_srs_download(self.ctx.h, self.h, out))
        pts = [g1_from_bytes(out.raw[64 * i:64 * i + 64]) for i in range(total)]
        return [pts[(1 << k) - 1:(2 << k) - 1] for k in range(n + 1)]

    def free(self):
        if self.h and self.ctx.h and not self.borrowed:
            self.ctx.lib.lh_srs_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MultilinearKzg:
    """pcs/multilinear/kzg.rs:150-361 (prover side)."""

    @staticmethod
    def setup(ctx, ss):
        """kzg.rs:166-228 with the trapdoor `ss` explicit."""
        h = C.c_void_p()
        _check(ctx.lib.lh_mkzg_setup(ctx.h, _fr_array(ss), len(ss), C.byref(h)))
        return MultilinearKzgParams(ctx, h)

    @staticmethod
    def upload(ctx, eqs_levels):
        flat = b"".join(g1_to_bytes(p) for lvl in eqs_levels for p in lvl)
        return MultilinearKzg.upload_bytes(ctx, flat, len(eqs_levels) - 1)

    @staticmethod
    def upload_bytes(ctx, eqs_flat, num_vars):
        """eqs_flat: (2^(num_vars+1) - 1) * 64 bytes in the device layout (e.g. from params_io.read_*)"""
        if len(eqs_flat) != 64 * ((2 << num_vars) - 1):
            raise ArgumentError("flat SRS has the wrong length")
        h = C.c_void_p()
        _check(ctx.lib.lh_srs_upload(ctx.h, eqs_flat, num_vars, C.byref(h)))
        return MultilinearKzgParams(ctx, h)

    @staticmethod
    def commit(pp, poly):
        out = lh_g1()
        _check(pp.ctx.lib.lh_mkzg_commit(pp.ctx.h, pp.h, poly.ptr, poly.num_vars, C.byref(out)))
        return g1_from_bytes(bytes(out))

    @staticmethod
    def batch_commit(pp, polys):
        if not polys:
            return []
        out = (lh_g1 * len(polys))()
        _check(pp.ctx.lib.lh_mkzg_batch_commit(pp.ctx.h, pp.h, _ptr_array(polys), len(polys), polys[0].num_vars, out))
        raw = C.string_at(out, 64 * len(polys))
        return [g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range(len(polys))]

    verify = staticmethod(lambda vp, comm, point, eval_, transcript: mkzg_verify(vp, comm, point, eval_, transcript))
    batch_verify = staticmethod(lambda vp, num_vars, comms, points, evals, transcript:
                                mkzg_batch_verify(vp, num_vars, comms, points, evals, transcript))

    @staticmethod
    def batch_commit_and_write(pp, polys, transcript):
        comms = MultilinearKzg.batch_commit(pp, polys)
        transcript.write_commitments(comms)
        return comms

    @staticmethod
    def open(pp, poly, point, transcript):
        out = lh_fr()
        _check(pp.ctx.lib.lh_mkzg_open(pp.ctx.h, pp.h, poly.ptr, poly.num_vars, _fr_array(point), transcript.p,
                                       C.byref(out)))
        return fr_from_bytes(bytes(out))

    @staticmethod
    def batch_open(pp, num_vars, polys, points, evals, transcript):
        evs = _evaluations(evals)
        flat = [v for p in points for v in p]
        for p in points:
            if len(p) != num_vars:
                raise InvalidPcsParam("Invalid point (expect point to have %d variates but got %d)" % (num_vars, len(p)))
        _check(pp.ctx.lib.lh_mkzg_batch_open(pp.ctx.h, pp.h, num_vars, _ptr_array(polys), len(polys), _fr_array(flat),
                                             len(points), evs, len(evals), transcript.p))


class MultilinearKzgVerifierParams:
    """MultilinearKzgVerifierParams (kzg.rs:79-101): g1, g2, ss[i] = s_i * g2.  Host only."""

    def __init__(self, handle):
        self.lib, self.h = _ffi.load(), handle

    @classmethod
    def setup(cls, ss):
        """the verifier half of MultilinearKzg.setup for the same trapdoor"""
        h = C.c_void_p()
        _check(_ffi.load().lh_mkzg_vp_setup(_fr_array(ss), len(ss), C.byref(h)))
        return cls(h)

    @classmethod
    def new(cls, g1, g2, ss):
        h = C.c_void_p()
        a, b = _ffi.lh_g1(), _ffi.lh_g2()
        C.memmove(C.byref(a), g1_to_bytes(g1), 64)
        C.memmove(C.byref(b), g2_to_bytes(g2), 128)
        arr = (_ffi.lh_g2 * max(len(ss), 1))()
        C.memmove(arr, b"".join(g2_to_bytes(p) for p in ss), 128 * len(ss))
        _check(_ffi.load().lh_mkzg_vp_new(C.byref(a), C.byref(b), arr, len(ss), C.byref(h)))
        return cls(h)

    @property
    def num_vars(self):
        return self.lib.lh_mkzg_vp_num_vars(self.h)

    def export(self):
        """-> (g1, g2, [ss_i])"""
        n = self.num_vars
        a, b, arr = _ffi.lh_g1(), _ffi.lh_g2(), (_ffi.lh_g2 * max(n, 1))()
        _check(self.lib.lh_mkzg_vp_export(self.h, C.byref(a), C.byref(b), arr))
        raw = C.string_at(arr, 128 * n)
        return g1_from_bytes(bytes(a)), g2_from_bytes(bytes(b)), [g2_from_bytes(raw[128 * i:128 * i + 128]) for i in range(n)]

    def free(self):
        if self.h:
            self.lib.lh_mkzg_vp_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pairings_product_is_identity(pairs):
    """util/arithmetic.rs:24-33: pairs of (G1 affine, G2 affine)"""
    n = len(pairs)
    ps, qs = (_ffi.lh_g1 * max(n, 1))(), (_ffi.lh_g2 * max(n, 1))()
    C.memmove(ps, b"".join(g1_to_bytes(p) for p, _ in pairs), 64 * n)
    C.memmove(qs, b"".join(g2_to_bytes(q) for _, q in pairs), 128 * n)
    out = C.c_int()
    _check(_ffi.load().lh_pairing_check(ps, qs, n, C.byref(out)))
    return bool(out.value)


def _evaluations(evals):
    evs = (lh_evaluation * max(len(evals), 1))()
    for i, e in enumerate(evals):
        evs[i].poly, evs[i].point = e.poly, e.point
        C.memmove(C.byref(evs[i].value), fr_to_bytes(e.value), 32)
    return evs


def _g1_array(pts):
    arr = (lh_g1 * max(len(pts), 1))()
    C.memmove(arr, b"".join(g1_to_bytes(p) for p in pts), 64 * len(pts))
    return arr


def mkzg_verify(vp, comm, point, eval_, transcript):
    """MultilinearKzg::verify (kzg.rs:330-361); raises InvalidPcsOpen"""
    _check(vp.lib.lh_mkzg_verify(vp.h, _g1_array([comm]), _fr_array(point), len(point), _fr_array([eval_]), transcript.p))


def mkzg_batch_verify(vp, num_vars, comms, points, evals, transcript):
    """additive::batch_verify (pcs/multilinear.rs:237-276)"""
    for p in points:
        if len(p) != num_vars:
            raise InvalidPcsParam("Invalid point (expect point to have %d variates but got %d)" % (num_vars, len(p)))
    flat = [v for p in points for v in p]
    _check(vp.lib.lh_mkzg_batch_verify(vp.h, num_vars, _g1_array(comms), len(comms), _fr_array(flat), len(points),
                                       _evaluations(evals), len(evals), transcript.p))


def sum_check_verify(prover, num_vars, degree, sum_, transcript):
    """ClassicSumCheck::<P>::verify (classic.rs:242-272) -> (final claim, challenges)"""
    ev, ch = lh_fr(), (lh_fr * max(num_vars, 1))()
    _check(_ffi.load().lh_sumcheck_verify(prover, num_vars, degree, _fr_array([sum_]), transcript.p, C.byref(ev), ch))
    return fr_from_bytes(bytes(ev)), _fr_list(ch, num_vars)


def lasso_verify(vp, table, num_vars, transcript):
    """Verifier of the Lasso argument (oracle/pyref/lasso.py:219-261).  A Keccak256Transcript must be fully consumed."""
    t = table.to_c()
    fn = vp.lib.lh_lasso_verify_zeromorph if isinstance(vp, ZeromorphVerifierParam) else vp.lib.lh_lasso_verify
    _check(fn(vp.h, C.byref(t), num_vars, transcript.p))
    if isinstance(transcript, Keccak256Transcript) and transcript.remaining():
        raise InvalidSnark("trailing bytes in proof")


# ------------------------------------------------------------------ pcs::multilinear::zeromorph over pcs::univariate::kzg
class UnivariateKzgParams:
    """UnivariateKzgParam (pcs/univariate/kzg.rs:38-66): powers_of_s_g1 resident on the device"""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def size(self):
        return self.ctx.lib.lh_usrs_size(self.h)

    def powers(self):
        n = self.size
        out = C.create_string_buffer(64 * n)
        _check(self.ctx.lib.lh_usrs_download(self.ctx.h, self.h, out))
        return [g1_from_bytes(out.raw[64 * i:64 * i + 64]) for i in range(n)]

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.lh_usrs_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ZeromorphProverParam:
    """ZeromorphKzgProverParam (zeromorph.rs:29-40) = the device SRS + the trim size"""

    def __init__(self, params, poly_size):
        self.params, self.poly_size, self.ctx = params, poly_size, params.ctx


class ZeromorphVerifierParam:
    """ZeromorphKzgVerifierParam (zeromorph.rs:42-65), host only"""

    def __init__(self, handle):
        self.lib, self.h = _ffi.load(), handle

    @classmethod
    def setup(cls, s, param_size, poly_size):
        h = C.c_void_p()
        _check(_ffi.load().lh_zeromorph_vp_setup(_fr_array([s]), param_size, poly_size, C.byref(h)))
        return cls(h)

    @classmethod
    def new(cls, g1, g2, s_g2, s_offset_g2):
        h, a = C.c_void_p(), _ffi.lh_g1()
        C.memmove(C.byref(a), g1_to_bytes(g1), 64)
        gs = []
        for p in (g2, s_g2, s_offset_g2):
            b = _ffi.lh_g2()
            C.memmove(C.byref(b), g2_to_bytes(p), 128)
            gs.append(b)
        _check(_ffi.load().lh_zeromorph_vp_new(C.byref(a), C.byref(gs[0]), C.byref(gs[1]), C.byref(gs[2]), C.byref(h)))
        return cls(h)

    def export(self):
        a, b, c_, d = _ffi.lh_g1(), _ffi.lh_g2(), _ffi.lh_g2(), _ffi.lh_g2()
        _check(self.lib.lh_zeromorph_vp_export(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d)))
        return g1_from_bytes(bytes(a)), g2_from_bytes(bytes(b)), g2_from_bytes(bytes(c_)), g2_from_bytes(bytes(d))

    def free(self):
        if self.h:
            self.lib.lh_zeromorph_vp_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Zeromorph:
    """Zeromorph<UnivariateKzg<Bn256>> (pcs/multilinear/zeromorph.rs:67-256)"""

    @staticmethod
    def setup(ctx, s, poly_size):
        """UnivariateKzg::setup (univariate/kzg.rs:175-218) with the trapdoor explicit"""
        h = C.c_void_p()
        _check(ctx.lib.lh_ukzg_setup(ctx.h, _fr_array([s]), poly_size, C.byref(h)))
        return UnivariateKzgParams(ctx, h)

    @staticmethod
    def upload(ctx, powers):
        h = C.c_void_p()
        _check(ctx.lib.lh_usrs_upload(ctx.h, b"".join(g1_to_bytes(p) for p in powers), len(powers), C.byref(h)))
        return UnivariateKzgParams(ctx, h)

    @staticmethod
    def trim(params, poly_size):
        """zeromorph.rs:90-108 (prover half)"""
        if params.size < poly_size:
            raise InvalidPcsParam("Too large poly_size to trim to (param supports poly_size up to %d but got %d)"
                                  % (params.size, poly_size))
        return ZeromorphProverParam(params, poly_size)

    @staticmethod
    def batch_commit(pp, polys):
        if not polys:
            return []
        out = (lh_g1 * len(polys))()
        _check(pp.ctx.lib.lh_zeromorph_batch_commit(pp.ctx.h, pp.params.h, pp.poly_size, _ptr_array(polys), len(polys),
                                                    polys[0].num_vars, out))
        raw = C.string_at(out, 64 * len(polys))
        return [g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range(len(polys))]

    @staticmethod
    def commit(pp, poly):
        return Zeromorph.batch_commit(pp, [poly])[0]

    @staticmethod
    def batch_commit_and_write(pp, polys, transcript):
        comms = Zeromorph.batch_commit(pp, polys)
        transcript.write_commitments(comms)
        return comms

    @staticmethod
    def open(pp, poly, point, transcript):
        _check(pp.ctx.lib.lh_zeromorph_open(pp.ctx.h, pp.params.h, pp.poly_size, poly.ptr, poly.num_vars,
                                            _fr_array(point), transcript.p))

    @staticmethod
    def batch_open(pp, num_vars, polys, points, evals, transcript):
        for p in points:
            if len(p) != num_vars:
                raise InvalidPcsParam("Invalid point (expect point to have %d variates but got %d)" % (num_vars, len(p)))
        flat = [v for p in points for v in p]
        _check(pp.ctx.lib.lh_zeromorph_batch_open(pp.ctx.h, pp.params.h, pp.poly_size, num_vars, _ptr_array(polys),
                                                  len(polys), _fr_array(flat), len(points), _evaluations(evals),
                                                  len(evals), transcript.p))

    @staticmethod
    def verify(vp, comm, point, eval_, transcript):
        _check(vp.lib.lh_zeromorph_verify(vp.h, _g1_array([comm]), _fr_array(point), len(point), _fr_array([eval_]),
                                          transcript.p))

    @staticmethod
    def batch_verify(vp, num_vars, comms, points, evals, transcript):
        flat = [v for p in points for v in p]
        _check(vp.lib.lh_zeromorph_batch_verify(vp.h, num_vars, _g1_array(comms), len(comms), _fr_array(flat),
                                                len(points), _evaluations(evals), len(evals), transcript.p))


# ------------------------------------------------------------------ Lasso
class LassoTable:
    """Decomposable table: c chunks of l bits, memories (chunk, subtable), g = sum coeff * prod E_i."""

    def __init__(self, num_chunks, chunk_bits, memories, g_terms):
        self.c, self.l, self.memories = num_chunks, chunk_bits, list(memories)
        self.g_terms = [(co % R_MOD, list(f)) for co, f in g_terms]

    @classmethod
    def range(cls, num_chunks=2, chunk_bits=16):
        return cls(num_chunks, chunk_bits, [(j, SUBTABLE_IDENTITY) for j in range(num_chunks)],
                   [(1 << (chunk_bits * j), [j]) for j in range(num_chunks)])

    @classmethod
    def bitwise(cls, kind, num_chunks=4, chunk_bits=16):
        return cls(num_chunks, chunk_bits, [(j, kind) for j in range(num_chunks)],
                   [(1 << (chunk_bits // 2 * j), [j]) for j in range(num_chunks)])

    def to_c(self):
        t = lh_lasso_table()
        t.num_chunks, t.chunk_bits, t.num_memories = self.c, self.l, len(self.memories)
        for i, (j, kind) in enumerate(self.memories):
            t.memory_chunk[i], t.memory_subtable[i] = j, kind
        t.num_terms = len(self.g_terms)
        for m, (co, facs) in enumerate(self.g_terms):
            C.memmove(C.byref(t.g_coeff[m]), fr_to_bytes(co), 32)
            t.g_num_factors[m] = len(facs)
            for k, f in enumerate(facs):
                t.g_factor[m][k] = f
        return t


def lasso_prove(pp, table, num_vars, dims, transcript):
    """dims: c device buffers of u32[2^num_vars] chunk indices. Appends the proof to `transcript`."""
    if len(dims) != table.c:
        raise ArgumentError("expected %d dim columns" % table.c)
    if len(table.memories) > _ffi.LH_LASSO_MAX_MEMORIES or table.c > _ffi.LH_LASSO_MAX_CHUNKS \
            or len(table.g_terms) > _ffi.LH_LASSO_MAX_TERMS:
        raise ArgumentError("table too large")
    t = table.to_c()
    if isinstance(pp, ZeromorphProverParam):
        _check(pp.ctx.lib.lh_lasso_prove_zeromorph(pp.ctx.h, pp.params.h, pp.poly_size, C.byref(t), num_vars,
                                                   _ptr_array(dims), transcript.p))
        return
    _check(pp.ctx.lib.lh_lasso_prove(pp.ctx.h, pp.h, C.byref(t), num_vars, _ptr_array(dims), transcript.p))


def attach_comm(ctx, rank, size, all_gather, shard_bit):
    """Attach a caller-supplied communicator to the context (sharded proving): the transport of the
    CPU / one-GPU tests.  all_gather(send: bytes) -> bytes of size * len(send), rank-major; device-side gathers are
    staged through it.  A multi-GPU node uses attach_comm_rccl."""
    def cb(_user, send, recv, nbytes):
        try:
            out = all_gather(C.string_at(send, nbytes))
            if len(out) != nbytes * size:
                return _ffi.LH_ERR_ARG
            C.memmove(recv, out, len(out))
            return 0
        except Exception:  # the library turns this into an lh_status
            return _ffi.LH_ERR_DEVICE
    comm = _ffi.lh_comm()
    comm.rank, comm.size, comm.user = rank, size, None
    comm.all_gather = _ffi._AG_CB(cb)
    ctx._comm_keepalive = (comm, cb)
    _check(ctx.lib.lh_ctx_set_comm(ctx.h, C.byref(comm), shard_bit))


def rccl_unique_id():
    """ncclGetUniqueId: 128 bytes to hand to every rank's attach_comm_rccl (call on one rank)"""
    buf = C.create_string_buffer(_ffi.LH_RCCL_UNIQUE_ID_BYTES)
    _check(_ffi.load().lh_rccl_unique_id(buf))
    return buf.raw


def attach_comm_rccl(ctx, rank, size, unique_id, shard_bit):
    """RCCL communicator on the ctx's device and stream (collective over the `size` ranks): every exchange of a sharded
    proof is an ncclAllGather enqueued behind the kernels that produce its input."""
    if len(unique_id) != _ffi.LH_RCCL_UNIQUE_ID_BYTES:
        raise ArgumentError("unique id must be %d bytes" % _ffi.LH_RCCL_UNIQUE_ID_BYTES)
    _check(ctx.lib.lh_ctx_set_comm_rccl(ctx.h, rank, size, unique_id, shard_bit))


def comm_stats(ctx):
    """{"device": collectives issued on the device side (RCCL), "host": through the host callback}"""
    out = (C.c_uint64 * 2)()
    _check(ctx.lib.lh_ctx_comm_stats(ctx.h, out))
    return {"device": int(out[0]), "host": int(out[1])}


def comm_phase_stats(ctx, reset=False):
    """collectives and bytes (this rank's contribution) by phase of the Lasso prove that issued them"""
    out = (C.c_uint64 * 16)()
    _check(ctx.lib.lh_ctx_comm_phase_stats(ctx.h, out, 1 if reset else 0))
    names = ["witness", "commit", "surge", "leaves", "gkr", "evals", "open", "outside"]
    return {nm: {"collectives": int(out[2 * i]), "bytes": int(out[2 * i + 1])} for i, nm in enumerate(names) if out[2 * i]}


def memory_stats(ctx):
    """lh_ctx_memory_stats: the workspace arena's high-water mark and reservation, the device's free / total bytes"""
    out = (C.c_uint64 * 4)()
    _check(ctx.lib.lh_ctx_memory_stats(ctx.h, out))
    return {"arena_high_water_bytes": int(out[0]), "arena_reserved_bytes": int(out[1]), "device_free_bytes": int(out[2]),
            "device_total_bytes": int(out[3])}


def attach_comm_loopback(ctx, rank, size, shard_bit):
    """lh_ctx_set_comm_loopback: measurement aid - every peer is a copy of this rank (the transcript is not a valid proof)"""
    _check(ctx.lib.lh_ctx_set_comm_loopback(ctx.h, rank, size, shard_bit))


def detach_comm(ctx):
    _check(ctx.lib.lh_ctx_set_comm(ctx.h, None, 0))
    ctx._comm_keepalive = None


def shard_of(column, rank, size, shard_bit):
    """This rank's shard of a full column (numpy array of 2^n entries): entries whose index bits
    [shard_bit, shard_bit + log2 size) equal `rank`, in index order (local index hi || lo)."""
    rho = size.bit_length() - 1
    n = len(column)
    return column.reshape(n >> (shard_bit + rho), size, 1 << shard_bit)[:, rank, :].reshape(-1).copy()


def shard_poly(poly, rank, size, shard_bit):
    """This rank's shard of a device-resident MultilinearPolynomial (lh_shard_extract): a table of num_vars - rho variables"""
    rho = size.bit_length() - 1
    n_local = (1 << poly.num_vars) >> rho
    out = poly.ctx.alloc(32 * n_local)
    _check(poly.ctx.lib.lh_shard_extract(poly.ctx.h, poly.ptr, n_local, shard_bit, rho, rank, 32, out.ptr))
    return MultilinearPolynomial(poly.ctx, out, poly.num_vars - rho)


def lasso_prove_sharded(pp, table, num_vars, dims_local, transcript):
    """One proof over the ranks of the attached communicator; same bytes as lasso_prove.
    dims_local: THIS RANK'S shard of every chunk column (device u32[2^(num_vars - rho)], see shard_of)."""
    if len(dims_local) != table.c:
        raise ArgumentError("expected %d dim columns" % table.c)
    t = table.to_c()
    _check(pp.ctx.lib.lh_lasso_prove_sharded(pp.ctx.h, pp.h, C.byref(t), num_vars, _ptr_array(dims_local), transcript.p))


def lasso_last_timing(ctx):
    out = (C.c_double * _ffi.LH_LASSO_NUM_PHASES)()
    _check(ctx.lib.lh_lasso_last_timing(ctx.h, out))
    names = ["witness", "commit", "surge", "leaves", "gkr", "evals", "open_n", "open_l", "total"]
    return dict(zip(names, list(out)))


ROUTE_FIELDS = ["open_small_depth", "open_small_passes", "eq_factored_rounds", "standard_rounds", "rw_leaf_rounds",
                "resident_tails", "resident_rounds", "packed_ts_pairs", "derived_commitments", "sorted_dim_reuse",
                "sharded_rounds", "shard_exchanges", "window_table_jobs", "open_precommit", "resident_layers", "pp_folds", "msm_half_batches"]


def lasso_last_route(ctx):
    """lh_lasso_last_route: which routes the last Lasso prove on `ctx` took (include/lasso_hip.h lh_lasso_route)"""
    out = (C.c_uint32 * 24)()
    _check(ctx.lib.lh_lasso_last_route(ctx.h, out))
    return dict(zip(ROUTE_FIELDS, [int(v) for v in out]))


def set_option(ctx, name, value):
    """lh_ctx_set_option: a route switch of the prover (include/lasso_hip.h lists them)"""
    _check(ctx.lib.lh_ctx_set_option(ctx.h, name.encode(), int(value)))


def get_option(ctx, name):
    out = C.c_int64(0)
    _check(ctx.lib.lh_ctx_get_option(ctx.h, name.encode(), C.byref(out)))
    return int(out.value)


def profile_enable(ctx, on=True):
    """lh_profile_enable: True / 1 = every instrumented launch synchronised (a separate, slower prove); 2 = live records of the
    bucket-accumulation launches only, nothing synchronised (usable inside a timed region); False / 0 = off"""
    _check(ctx.lib.lh_profile_enable(ctx.h, int(on)))


def profile_read(ctx):
    """-> list of dicts {name, ms, bytes, muls, items}, one per instrumented launch since enable."""
    n = C.c_size_t()
    _check(ctx.lib.lh_profile_read(ctx.h, None, 0, C.byref(n)))
    arr = (_ffi.lh_prof_rec * max(n.value, 1))()
    _check(ctx.lib.lh_profile_read(ctx.h, arr, n.value, C.byref(n)))
    return [dict(name=r.name.decode(), ms=r.ms, bytes=r.bytes, muls=r.muls, items=r.items) for r in arr[:n.value]]


from . import expression  # noqa: E402,F401  (host mirror of util::expression)
