"""BN254 field constants and helpers (Python big-int).  TEST INFRASTRUCTURE ONLY.

Oracle for the hot path of DoHoonKim8/halo2-lasso; the reference takes all of this
from the external crate halo2_curves 0.3.3 (plonkish_backend/Cargo.toml:7), used via
plonkish_backend/src/util/arithmetic.rs:15-22.  Nothing in the product may import this.

PARITY UNPINNED at byte level: the reference holds no golden vectors;
the constants below are pinned by public KATs (tests/test_oracle_kat.py).
"""

# scalar field Fr (checked to be 254 bits by reference arithmetic.rs:202-205)
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
# base field Fq ; curve y^2 = x^3 + 3 ; generator (1, 2)
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
CURVE_B = 3
MONT_R = 1 << 256  # Montgomery radix of halo2curves' [u64;4] representation


def fr(x):
    return x % R_MOD


def fr_inv(x):
    x %= R_MOD
    if x == 0:
        raise ZeroDivisionError("Fr inverse of zero")
    return pow(x, -1, R_MOD)


def fq_inv(x):
    x %= Q_MOD
    if x == 0:
        raise ZeroDivisionError("Fq inverse of zero")
    return pow(x, -1, Q_MOD)


def batch_invert(xs, mod=R_MOD):
    """Montgomery's trick; zeros stay zero (ff::BatchInvert semantics)."""
    prods, acc = [], 1
    for x in xs:
        prods.append(acc)
        if x % mod:
            acc = acc * x % mod
    inv = pow(acc, -1, mod)
    out = [0] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        x = xs[i] % mod
        if x:
            out[i] = inv * prods[i] % mod
            inv = inv * x % mod
    return out


# ---- byte encodings (reference util/transcript.rs:157-165, util/hash.rs:19-21) ----
def to_repr_le(x):
    return int(x).to_bytes(32, "little")


def from_repr_le(b, mod=R_MOD):
    x = int.from_bytes(b, "little")
    if x >= mod:
        raise ValueError("non-canonical field element")
    return x


# ---- Montgomery in-memory form, the bytes of a Rust `&[Fr]` / the C-ABI ----
def to_mont_bytes(x, mod=R_MOD):
    return ((x % mod) * MONT_R % mod).to_bytes(32, "little")


def from_mont_bytes(b, mod=R_MOD):
    x = int.from_bytes(b, "little")
    return x * pow(MONT_R, -1, mod) % mod


def fe_mod_from_le_bytes(b):
    """reference util/arithmetic.rs:150-152"""
    return int.from_bytes(b, "little") % R_MOD
