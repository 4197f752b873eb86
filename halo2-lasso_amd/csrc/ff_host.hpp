// Host-side BN254 arithmetic on 4 x u64 limbs (unsigned __int128 products).
// Used by the host half of the prover: Fiat-Shamir challenges, round-message interpolation,
// claim bookkeeping and the final window combine / normalisation of an MSM.
// Same Montgomery representation as the device code (ff.cuh) and as halo2curves.
#pragma once
#include <vector>
#include <stdint.h>
#include <string.h>

namespace lh {
namespace host {

typedef unsigned __int128 u128;

struct FrTag {
  static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull,
                                      0x30644e72e131a029ull};
  static constexpr uint64_t R1[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull,
                                     0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull,
                                     0x0216d0b17f4e44a5ull};
  static constexpr uint64_t INV = 0xc2e1f593efffffffull;
};
struct FqTag {
  static constexpr uint64_t MOD[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull,
                                      0x30644e72e131a029ull};
  static constexpr uint64_t R1[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull,
                                     0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull,
                                     0x06d89f71cab8351full};
  static constexpr uint64_t INV = 0x87d20782e4866389ull;
};

template <class T>
struct F {
  uint64_t l[4];

  static F zero() { return F{{0, 0, 0, 0}}; }
  static F one() { return F{{T::R1[0], T::R1[1], T::R1[2], T::R1[3]}}; }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const F& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
  bool operator!=(const F& o) const { return !(*this == o); }

  static bool geq_mod(const uint64_t* a) {
    for (int i = 3; i >= 0; i--) {
      if (a[i] > T::MOD[i]) return true;
      if (a[i] < T::MOD[i]) return false;
    }
    return true;
  }
  static void sub_mod(uint64_t* a) {
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a[i] - T::MOD[i] - bw;
      a[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
  }
  F operator+(const F& o) const {
    F r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)l[i] + o.l[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  F operator-(const F& o) const {
    F r;
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)l[i] - o.l[i] - bw;
      r.l[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
    if (bw) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)r.l[i] + T::MOD[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  F operator-() const { return zero() - *this; }
  F operator*(const F& o) const {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)l[j] * o.l[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * T::INV;
      c = (u128)m * T::MOD[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * T::MOD[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    F r{{t[0], t[1], t[2], t[3]}};
    if (geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  F& operator+=(const F& o) { return *this = *this + o; }
  F& operator-=(const F& o) { return *this = *this - o; }
  F& operator*=(const F& o) { return *this = *this * o; }
  F dbl() const { return *this + *this; }
  F sqr() const { return *this * *this; }

  static F from_u64(uint64_t v) {
    F c{{v, 0, 0, 0}};
    return c * F{{T::R2[0], T::R2[1], T::R2[2], T::R2[3]}};
  }
  // canonical little-endian limbs -> Montgomery (input must be < modulus)
  static F from_canonical(const uint64_t* c) {
    F x{{c[0], c[1], c[2], c[3]}};
    return x * F{{T::R2[0], T::R2[1], T::R2[2], T::R2[3]}};
  }
  void to_canonical(uint64_t* out) const {
    F o = *this * F{{1, 0, 0, 0}};
    memcpy(out, o.l, 32);
  }
  // to_repr(): 32 bytes little-endian canonical
  void to_repr(uint8_t* out) const {
    uint64_t c[4];
    to_canonical(c);
    memcpy(out, c, 32);  // little-endian host
  }
  F pow(const uint64_t* e) const {
    F acc = one();
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        acc = acc.sqr();
        if ((e[i] >> b) & 1) acc = acc * *this;
      }
    return acc;
  }
  // Field::invert(); zero -> zero (callers check)
  F inv() const {
    uint64_t e[4] = {T::MOD[0] - 2, T::MOD[1], T::MOD[2], T::MOD[3]};
    return pow(e);
  }
};

typedef F<FrTag> Fr;
typedef F<FqTag> Fq;

// 32 little-endian bytes (any 256-bit value) reduced mod r: fe_mod_from_le_bytes
// (reference util/arithmetic.rs:150-152)
inline Fr fr_mod_from_le_bytes(const uint8_t* b) {
  uint64_t c[4];
  memcpy(c, b, 32);
  // value < 2^256 < 6r: subtract r while >= r
  while (Fr::geq_mod(c)) Fr::sub_mod(c);
  return Fr::from_canonical(c);
}

// ---------------------------------------------------------------- G1, XYZZ coordinates (see ec.cuh)
struct G1Affine {
  Fq x, y;
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
};
struct G1Xyzz {
  Fq x, y, zz, zzz;
  static G1Xyzz identity() { return G1Xyzz{Fq::zero(), Fq::zero(), Fq::zero(), Fq::zero()}; }
  bool is_identity() const { return zz.is_zero(); }
};

inline G1Xyzz g1_dbl(const G1Xyzz& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
  Fq u = p.y.dbl(), v = u.sqr(), w = u * v, s = p.x * v, xx = p.x.sqr();
  Fq m = xx.dbl() + xx;
  G1Xyzz r;
  r.x = m.sqr() - s.dbl();
  r.y = m * (s - r.x) - w * p.y;
  r.zz = v * p.zz;
  r.zzz = w * p.zzz;
  return r;
}
inline G1Xyzz g1_add(const G1Xyzz& p, const G1Xyzz& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fq u1 = p.x * q.zz, u2 = q.x * p.zz, s1 = p.y * q.zzz, s2 = q.y * p.zzz;
  Fq pp_ = u2 - u1, r_ = s2 - s1;
  if (pp_.is_zero()) return r_.is_zero() ? g1_dbl(p) : G1Xyzz::identity();
  Fq pp = pp_.sqr(), ppp = pp_ * pp, qq = u1 * pp;
  G1Xyzz r;
  r.x = r_.sqr() - ppp - qq.dbl();
  r.y = r_ * (qq - r.x) - s1 * ppp;
  r.zz = p.zz * q.zz * pp;
  r.zzz = p.zzz * q.zzz * ppp;
  return r;
}
inline G1Affine g1_to_affine(const G1Xyzz& p) {
  if (p.is_identity()) return G1Affine{Fq::zero(), Fq::zero()};
  // x = X/ZZ, y = Y/ZZZ ; one inversion: i = 1/(ZZ*ZZZ)
  Fq i = (p.zz * p.zzz).inv();
  return G1Affine{p.x * (i * p.zzz), p.y * (i * p.zz)};
}
// out[i] = affine form of p[i], i < n, with ONE field inversion (Montgomery's trick over the ZZ * ZZZ of the finite points)
inline void g1_batch_to_affine(const G1Xyzz* p, size_t n, G1Affine* out) {
  std::vector<Fq> d(n), pre(n + 1);
  pre[0] = Fq::one();
  for (size_t i = 0; i < n; i++) {
    d[i] = p[i].is_identity() ? Fq::one() : p[i].zz * p[i].zzz;
    pre[i + 1] = pre[i] * d[i];
  }
  Fq inv = pre[n].inv();
  for (size_t i = n; i-- > 0;) {
    const Fq di = inv * pre[i];  // 1 / d[i]
    inv = inv * d[i];
    out[i] = p[i].is_identity() ? G1Affine{Fq::zero(), Fq::zero()} : G1Affine{p[i].x * (di * p[i].zzz), p[i].y * (di * p[i].zz)};
  }
}
inline G1Xyzz g1_from_affine(const G1Affine& a) {
  if (a.is_identity()) return G1Xyzz::identity();
  return G1Xyzz{a.x, a.y, Fq::one(), Fq::one()};
}
inline G1Xyzz g1_mul(const G1Xyzz& p, const Fr& k) {
  uint64_t c[4];
  k.to_canonical(c);
  G1Xyzz acc = G1Xyzz::identity();
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      acc = g1_dbl(acc);
      if ((c[i] >> b) & 1) acc = g1_add(acc, p);
    }
  return acc;
}

}  // namespace host
}  // namespace lh
