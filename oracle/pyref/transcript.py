"""Keccak-256 Fiat-Shamir transcript.  TEST INFRASTRUCTURE ONLY.

Follows reference plonkish_backend/src/util/transcript.rs:101-238 and util/hash.rs:19-21:
  * hash absorbs the LITTLE-endian canonical repr of Fr (for G1: x then y, each LE);
  * the proof stream stores the byte-REVERSED (big-endian) repr (transcript.rs:157-165,
    213-226);
  * squeeze: h = keccak.finalize_reset(); keccak.update(h); challenge = LE(h) mod r
    (transcript.rs:126-131, arithmetic.rs:150-152);
  * the identity point cannot be absorbed (transcript.rs:172-179).
"""
from .field import R_MOD, Q_MOD, to_repr_le, fe_mod_from_le_bytes
from .keccak import Keccak256
from .curve import is_on_curve


class TranscriptError(Exception):
    pass


class Keccak256Transcript:
    def __init__(self, proof=None):
        self.state = Keccak256()
        self.stream = bytearray() if proof is None else bytes(proof)
        self.pos = 0
        self.reading = proof is not None

    # FieldTranscript
    def squeeze_challenge(self):
        h = self.state.finalize_reset()
        self.state.update(h)
        return fe_mod_from_le_bytes(h)

    def squeeze_challenges(self, n):
        return [self.squeeze_challenge() for _ in range(n)]

    def common_field_element(self, fe):
        self.state.update(to_repr_le(fe % R_MOD))

    def common_field_elements(self, fes):
        for fe in fes:
            self.common_field_element(fe)

    # FieldTranscriptWrite
    def write_field_element(self, fe):
        self.common_field_element(fe)
        self.stream += to_repr_le(fe % R_MOD)[::-1]

    def write_field_elements(self, fes):
        for fe in fes:
            self.write_field_element(fe)

    # FieldTranscriptRead
    def _read(self, n):
        if self.pos + n > len(self.stream):
            raise TranscriptError("failed to fill whole buffer")
        b = self.stream[self.pos:self.pos + n]
        self.pos += n
        return b

    def read_field_element(self):
        x = int.from_bytes(self._read(32), "big")
        if x >= R_MOD:
            raise TranscriptError("Invalid field element encoding in proof")
        self.common_field_element(x)
        return x

    def read_field_elements(self, n):
        return [self.read_field_element() for _ in range(n)]

    # Transcript<G1Affine, Fr>
    def common_commitment(self, pt):
        if pt is None:
            raise TranscriptError("Invalid elliptic curve point encoding")
        self.state.update(to_repr_le(pt[0]))
        self.state.update(to_repr_le(pt[1]))

    def write_commitment(self, pt):
        self.common_commitment(pt)
        self.stream += to_repr_le(pt[0])[::-1] + to_repr_le(pt[1])[::-1]

    def write_commitments(self, pts):
        for pt in pts:
            self.write_commitment(pt)

    def read_commitment(self):
        x = int.from_bytes(self._read(32), "big")
        y = int.from_bytes(self._read(32), "big")
        if x >= Q_MOD or y >= Q_MOD or not is_on_curve((x, y)):
            raise TranscriptError("Invalid elliptic curve point encoding in proof")
        self.common_commitment((x, y))
        return (x, y)

    def read_commitments(self, n):
        return [self.read_commitment() for _ in range(n)]

    def into_proof(self):
        return bytes(self.stream)
