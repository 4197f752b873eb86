// Keccak256 Fiat-Shamir transcript on the host, behind the lh_transcript callback table.
// Mirrors reference plonkish_backend/src/util/transcript.rs:99-238 (Keccak256Transcript) with
// sha3 0.10.6 `Keccak256` semantics (legacy 0x01 padding, rate 136).  The transcript is a strictly
// serial dependency between rounds and hashes a few hundred bytes per round,
// so it stays on the CPU; only field elements cross to / from the GPU.
#include "host.hpp"

namespace lh {

static const uint64_t RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
    0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int ROT[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                            25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};  // index x + 5y

static inline uint64_t rol(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

static void keccak_f(uint64_t a[25]) {
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], ROT[x + 5 * y]);
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[rnd];
  }
}

void Keccak256::absorb_block(const uint8_t* block) {
  for (int i = 0; i < RATE / 8; i++) {
    uint64_t w;
    memcpy(&w, block + 8 * i, 8);
    state[i] ^= w;
  }
  keccak_f(state);
}

void Keccak256::update(const uint8_t* data, size_t len) {
  while (len) {
    size_t take = RATE - buf_len < len ? RATE - buf_len : len;
    memcpy(buf + buf_len, data, take);
    buf_len += take;
    data += take;
    len -= take;
    if (buf_len == RATE) {
      absorb_block(buf);
      buf_len = 0;
    }
  }
}

void Keccak256::finalize_reset(uint8_t out[32]) {
  memset(buf + buf_len, 0, RATE - buf_len);
  buf[buf_len] ^= 0x01;
  buf[RATE - 1] ^= 0x80;
  absorb_block(buf);
  memcpy(out, state, 32);
  memset(state, 0, sizeof(state));
  buf_len = 0;
}

// ------------------------------------------------------------------ KeccakTranscript callbacks
static int kt_common_fe(void* u, const lh_fr* fe) {
  auto* t = (KeccakTranscript*)u;
  host::Fr f;
  memcpy(&f, fe, 32);
  uint8_t repr[32];
  f.to_repr(repr);  // hash absorbs the little-endian canonical repr (hash.rs:19-21)
  t->hash.update(repr, 32);
  return LH_OK;
}
static int kt_write_fe(void* u, const lh_fr* fe) {
  auto* t = (KeccakTranscript*)u;
  host::Fr f;
  memcpy(&f, fe, 32);
  uint8_t repr[32];
  f.to_repr(repr);
  t->hash.update(repr, 32);
  for (int i = 31; i >= 0; i--) t->stream.push_back(repr[i]);  // stream stores the reversed repr
  return LH_OK;
}
static int kt_squeeze(void* u, lh_fr* out) {
  auto* t = (KeccakTranscript*)u;
  uint8_t h[32];
  t->hash.finalize_reset(h);
  t->hash.update(h, 32);
  host::Fr c = host::fr_mod_from_le_bytes(h);
  memcpy(out, &c, 32);
  return LH_OK;
}
static int kt_common_comm(void* u, const lh_g1* pt) {
  auto* t = (KeccakTranscript*)u;
  host::G1Affine p;
  memcpy(&p, pt, 64);
  if (p.is_identity()) {
    set_last_error("Invalid elliptic curve point encoding");  // transcript.rs:172-179
    return LH_ERR_TRANSCRIPT;
  }
  uint8_t repr[32];
  p.x.to_repr(repr);
  t->hash.update(repr, 32);
  p.y.to_repr(repr);
  t->hash.update(repr, 32);
  return LH_OK;
}
static int kt_write_comm(void* u, const lh_g1* pt) {
  auto* t = (KeccakTranscript*)u;
  int rc = kt_common_comm(u, pt);
  if (rc) return rc;
  host::G1Affine p;
  memcpy(&p, pt, 64);
  uint8_t repr[32];
  p.x.to_repr(repr);
  for (int i = 31; i >= 0; i--) t->stream.push_back(repr[i]);
  p.y.to_repr(repr);
  for (int i = 31; i >= 0; i--) t->stream.push_back(repr[i]);
  return LH_OK;
}

// TranscriptRead: the stream holds byte-reversed reprs; a value >= the modulus or a point off the curve is
// rejected (from_repr_vartime / CurveAffine::from_xy, transcript.rs:138-154,185-210)
enum ReadStatus { READ_OK, READ_EOF, READ_BAD };
template <class F>
static ReadStatus read_repr(KeccakTranscript* t, F* out) {
  if (t->stream.size() - t->pos < 32) return READ_EOF;
  uint64_t c[4];
  uint8_t repr[32];
  for (int i = 0; i < 32; i++) repr[i] = t->stream[t->pos + 31 - i];
  t->pos += 32;
  memcpy(c, repr, 32);
  if (F::geq_mod(c)) return READ_BAD;
  *out = F::from_canonical(c);
  return READ_OK;
}
static int read_failed(ReadStatus st, const char* what) {
  set_last_error(st == READ_EOF ? "failed to fill whole buffer" : what);  // read_exact's UnexpectedEof text
  return LH_ERR_TRANSCRIPT;
}
static int kt_read_fe(void* u, lh_fr* out) {
  auto* t = (KeccakTranscript*)u;
  host::Fr f;
  ReadStatus st = read_repr(t, &f);
  if (st != READ_OK) return read_failed(st, "Invalid field element encoding in proof");
  memcpy(out, &f, 32);
  return kt_common_fe(u, out);
}
static int kt_read_comm(void* u, lh_g1* out) {
  auto* t = (KeccakTranscript*)u;
  host::G1Affine p;
  ReadStatus sx = read_repr(t, &p.x);
  if (sx == READ_EOF) return read_failed(sx, "");
  ReadStatus sy = read_repr(t, &p.y);
  if (sy == READ_EOF) return read_failed(sy, "");
  // on the curve y^2 = x^3 + 3 (the identity has no affine coordinates and cannot be encoded)
  if (sx != READ_OK || sy != READ_OK || !(p.y.sqr() == p.x.sqr() * p.x + host::Fq::from_u64(3)))
    return read_failed(READ_BAD, "Invalid elliptic curve point encoding in proof");
  memcpy(out, &p, 64);
  return kt_common_comm(u, out);
}

KeccakTranscript::KeccakTranscript() {
  vt.user = this;
  vt.write_field_element = kt_write_fe;
  vt.common_field_element = kt_common_fe;
  vt.squeeze_challenge = kt_squeeze;
  vt.write_commitment = kt_write_comm;
  vt.common_commitment = kt_common_comm;
  vt.read_field_element = kt_read_fe;
  vt.read_commitment = kt_read_comm;
}

}  // namespace lh
