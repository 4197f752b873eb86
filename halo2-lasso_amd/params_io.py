"""Wire format of the PCS parameter structs, so that a host using the reference can exchange SRS files with this
library.

The reference derives serde `Serialize`/`Deserialize` on its param structs (pcs/multilinear/kzg.rs:25-102,
pcs/univariate/kzg.rs:33-111, pcs/multilinear/zeromorph.rs:22-65) and depends on bincode 1.3.3
(plonkish_backend/Cargo.toml:16).  bincode 1.x default options: struct fields in declaration order, no framing;
`Vec<T>` = u64 little-endian length followed by the elements; integers fixed-width little-endian.

ASSUMPTION (the curve crate is an un-vendored git dependency, its source is not in this environment): halo2curves
0.3.3 built with `derive_serde` serializes a field element as its four raw u64 limbs, i.e. the Montgomery form it
holds in memory -- 32 bytes per Fq, 64 per G1Affine {x, y}, 128 per G2Affine {x: {c0, c1}, y: {c0, c1}}, identity =
all zeros.  That is byte for byte the layout of `lh_g1` / `lh_g2`, so point arrays move without conversion.
`repr="canonical"` writes / reads canonical little-endian integers instead (the layout later halo2curves releases
use for their serde impl) at one Montgomery conversion per coordinate.
"""
import struct

from . import g1_to_bytes, g1_from_bytes, g2_to_bytes, g2_from_bytes, Q_MOD

_MONT = 1 << 256
_MONT_INV_Q = pow(_MONT, -1, Q_MOD)


def _conv(raw, to_canonical):
    """re-encode every 32-byte coordinate of `raw` between Montgomery and canonical little-endian"""
    out = bytearray(len(raw))
    for i in range(0, len(raw), 32):
        v = int.from_bytes(raw[i:i + 32], "little")
        v = v * _MONT_INV_Q % Q_MOD if to_canonical else v * _MONT % Q_MOD
        out[i:i + 32] = v.to_bytes(32, "little")
    return bytes(out)


def _enc(raw, repr_):
    return raw if repr_ == "raw" else _conv(raw, True)


def _dec(raw, repr_):
    return raw if repr_ == "raw" else _conv(raw, False)


class _Reader:
    def __init__(self, data):
        self.d, self.p = memoryview(data), 0

    def take(self, n):
        if self.p + n > len(self.d):
            raise ValueError("unexpected end of parameter file")
        out = bytes(self.d[self.p:self.p + n])
        self.p += n
        return out

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def done(self):
        if self.p != len(self.d):
            raise ValueError("trailing bytes in parameter file")


# ------------------------------------------------------------------ MultilinearKzg{Params, ProverParams, VerifierParams}
def write_multilinear_kzg_params(g1, eqs_flat, num_levels, g2, ss, repr_="raw"):
    """MultilinearKzgParams { g1, eqs: Vec<Vec<G1Affine>>, g2, ss: Vec<G2Affine> } (kzg.rs:25-31).
    g1 / g2 / ss: points as integer tuples; eqs_flat: the library's flat SRS bytes (level k, 2^k points, at offset
    2^k - 1; `MultilinearKzgParams.eqs_bytes()`), num_levels = num_vars + 1."""
    out = [_enc(g1_to_bytes(g1), repr_), struct.pack("<Q", num_levels)]
    for k in range(num_levels):
        out.append(struct.pack("<Q", 1 << k))
        out.append(_enc(eqs_flat[64 * ((1 << k) - 1):64 * ((2 << k) - 1)], repr_))
    out.append(_enc(g2_to_bytes(g2), repr_))
    out.append(struct.pack("<Q", len(ss)))
    out += [_enc(g2_to_bytes(p), repr_) for p in ss]
    return b"".join(out)


def read_multilinear_kzg_params(data, repr_="raw"):
    """-> dict(g1, eqs_flat, num_levels, g2, ss); level k must hold 2^k points (kzg.rs:174-212)"""
    r = _Reader(data)
    g1 = g1_from_bytes(_dec(r.take(64), repr_))
    levels = r.u64()
    flat = []
    for k in range(levels):
        n = r.u64()
        if n != 1 << k:
            raise ValueError("eqs[%d] holds %d points, expected %d" % (k, n, 1 << k))
        flat.append(_dec(r.take(64 * n), repr_))
    g2 = g2_from_bytes(_dec(r.take(128), repr_))
    ss = [g2_from_bytes(_dec(r.take(128), repr_)) for _ in range(r.u64())]
    r.done()
    return dict(g1=g1, eqs_flat=b"".join(flat), num_levels=levels, g2=g2, ss=ss)


def write_multilinear_kzg_prover_params(g1, eqs_flat, num_levels, repr_="raw"):
    """MultilinearKzgProverParams { g1, eqs } (kzg.rs:56-60)"""
    out = [_enc(g1_to_bytes(g1), repr_), struct.pack("<Q", num_levels)]
    for k in range(num_levels):
        out += [struct.pack("<Q", 1 << k), _enc(eqs_flat[64 * ((1 << k) - 1):64 * ((2 << k) - 1)], repr_)]
    return b"".join(out)


def read_multilinear_kzg_prover_params(data, repr_="raw"):
    r = _Reader(data)
    g1 = g1_from_bytes(_dec(r.take(64), repr_))
    levels = r.u64()
    flat = []
    for k in range(levels):
        if r.u64() != 1 << k:
            raise ValueError("eqs[%d]: wrong length" % k)
        flat.append(_dec(r.take(64 << k), repr_))
    r.done()
    return dict(g1=g1, eqs_flat=b"".join(flat), num_levels=levels)


def write_multilinear_kzg_verifier_params(g1, g2, ss, repr_="raw"):
    """MultilinearKzgVerifierParams { g1, g2, ss } (kzg.rs:79-84)"""
    return b"".join([_enc(g1_to_bytes(g1), repr_), _enc(g2_to_bytes(g2), repr_), struct.pack("<Q", len(ss))] +
                    [_enc(g2_to_bytes(p), repr_) for p in ss])


def read_multilinear_kzg_verifier_params(data, repr_="raw"):
    r = _Reader(data)
    g1 = g1_from_bytes(_dec(r.take(64), repr_))
    g2 = g2_from_bytes(_dec(r.take(128), repr_))
    ss = [g2_from_bytes(_dec(r.take(128), repr_)) for _ in range(r.u64())]
    r.done()
    return dict(g1=g1, g2=g2, ss=ss)


# ------------------------------------------------------------------ UnivariateKzgParam / Zeromorph verifier param
def write_univariate_kzg_param(powers_g1_bytes, powers_g2, repr_="raw"):
    """UnivariateKzgParam { powers_of_s_g1, powers_of_s_g2 } (univariate/kzg.rs:33-41); powers_g1_bytes: the library's
    device layout (64 bytes per point), powers_g2: integer tuples"""
    n1 = len(powers_g1_bytes) // 64
    return b"".join([struct.pack("<Q", n1), _enc(powers_g1_bytes, repr_), struct.pack("<Q", len(powers_g2))] +
                    [_enc(g2_to_bytes(p), repr_) for p in powers_g2])


def read_univariate_kzg_param(data, repr_="raw"):
    r = _Reader(data)
    g1 = _dec(r.take(64 * r.u64()), repr_)
    g2 = [g2_from_bytes(_dec(r.take(128), repr_)) for _ in range(r.u64())]
    r.done()
    return dict(powers_g1_bytes=g1, powers_g2=g2)


def write_zeromorph_verifier_param(g1, g2, s_g2, s_offset_g2, repr_="raw"):
    """ZeromorphKzgVerifierParam { vp: UnivariateKzgVerifierParam { g1, g2, s_g2 }, s_offset_g2 } (zeromorph.rs:42-50,
    univariate/kzg.rs:89-97)"""
    return b"".join([_enc(g1_to_bytes(g1), repr_)] + [_enc(g2_to_bytes(p), repr_) for p in (g2, s_g2, s_offset_g2)])


def read_zeromorph_verifier_param(data, repr_="raw"):
    r = _Reader(data)
    g1 = g1_from_bytes(_dec(r.take(64), repr_))
    g2, s_g2, s_off = (g2_from_bytes(_dec(r.take(128), repr_)) for _ in range(3))
    r.done()
    return dict(g1=g1, g2=g2, s_g2=s_g2, s_offset_g2=s_off)
