"""Keccak-256 with the legacy 0x01 padding (NOT hashlib.sha3_256).  TEST INFRASTRUCTURE ONLY.

The reference hashes with sha3 0.10.6 `Keccak256` (plonkish_backend/src/util/hash.rs:5-8,
plonkish_backend/Cargo.toml:17); this restates the published Keccak-f[1600] permutation.
Pinned by KATs: Keccak256("") and Keccak256("abc") in tests/test_oracle_kat.py.
"""

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]
_M = (1 << 64) - 1
RATE = 136


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def keccak_f(a):
    """a: 25 lanes, index x + 5*y."""
    for rc in _RC:
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = _rol(a[x + 5 * y], _ROT[x][y])
        a = [b[i] ^ ((~b[(i % 5 + 1) % 5 + 5 * (i // 5)]) & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) & _M
             for i in range(25)]
        a[0] ^= rc
    return a


class Keccak256:
    """Streaming hasher with `finalize_reset`, as `sha3::Keccak256`.  `pad`: the domain byte - 0x01 is legacy Keccak
    (what the reference hashes with); 0x06 turns the same sponge into FIPS-202 SHA3-256, which tests/test_oracle.py
    compares with hashlib on messages of every length around the block boundaries (pins permutation, rate, padding)."""

    def __init__(self, pad=0x01):
        self.buf = bytearray()
        self.state = [0] * 25
        self.pad = pad

    def _absorb_block(self, block):
        for i in range(RATE // 8):
            self.state[i] ^= int.from_bytes(block[8 * i:8 * i + 8], "little")
        self.state = keccak_f(self.state)

    def update(self, data):
        self.buf += data
        while len(self.buf) >= RATE:
            self._absorb_block(self.buf[:RATE])
            del self.buf[:RATE]

    def finalize_reset(self):
        pad = bytearray(self.buf)
        pad.append(self.pad)
        pad += b"\x00" * (RATE - len(pad))
        pad[-1] |= 0x80
        self._absorb_block(pad)
        out = b"".join(self.state[i].to_bytes(8, "little") for i in range(4))
        self.buf = bytearray()
        self.state = [0] * 25
        return out


def keccak256(data):
    h = Keccak256()
    h.update(data)
    return h.finalize_reset()
