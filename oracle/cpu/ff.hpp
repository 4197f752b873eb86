// TEST INFRASTRUCTURE ONLY - CPU oracle arithmetic (never linked into the product).
//
// BN254 Fr / Fq on 4 x u64 limbs and G1 in Jacobian coordinates: a restatement of what the
// reference takes from halo2_curves 0.3.3 (plonkish_backend/Cargo.toml:7, used through
// plonkish_backend/src/util/arithmetic.rs:15-22).  Written independently of the product's
// arithmetic on purpose: separated-operand-scanning Montgomery (full 512-bit product, then
// reduction) instead of CIOS, Jacobian instead of XYZZ, so that agreement on the golden vectors
// means something.  Parity with the reference itself is UNPINNED (see oracle/README.md).
#pragma once
#include <stdint.h>
#include <string.h>

namespace orc {

typedef unsigned __int128 u128;

struct FrP {
  static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull,
                                      0x30644e72e131a029ull};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull,
                                     0x0216d0b17f4e44a5ull};
  static constexpr uint64_t INV = 0xc2e1f593efffffffull;
};
struct FqP {
  static constexpr uint64_t MOD[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull,
                                      0x30644e72e131a029ull};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull,
                                     0x06d89f71cab8351full};
  static constexpr uint64_t INV = 0x87d20782e4866389ull;
};

template <class P>
struct Fe {
  uint64_t v[4];

  static Fe zero() { return Fe{{0, 0, 0, 0}}; }
  static Fe from_raw(const uint64_t* c) {  // canonical -> Montgomery
    Fe x{{c[0], c[1], c[2], c[3]}}, r2{{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}};
    return x * r2;
  }
  static Fe from_u64(uint64_t x) {
    uint64_t c[4] = {x, 0, 0, 0};
    return from_raw(c);
  }
  static Fe one() { return from_u64(1); }
  bool is_zero() const { return !(v[0] | v[1] | v[2] | v[3]); }
  bool operator==(const Fe& o) const { return !memcmp(v, o.v, 32); }
  bool operator!=(const Fe& o) const { return !(*this == o); }

  static bool ge_mod(const uint64_t* a) {
    for (int i = 3; i >= 0; i--)
      if (a[i] != P::MOD[i]) return a[i] > P::MOD[i];
    return true;
  }
  static void sub_mod_inplace(uint64_t* a) {
    uint64_t bw = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a[i] - P::MOD[i] - bw;
      a[i] = (uint64_t)d;
      bw = (uint64_t)(d >> 127);
    }
  }
  Fe operator+(const Fe& o) const {
    Fe r;
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) {
      u128 s = (u128)v[i] + o.v[i] + c;
      r.v[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    if (ge_mod(r.v)) sub_mod_inplace(r.v);
    return r;
  }
  Fe operator-(const Fe& o) const {
    Fe r;
    uint64_t bw = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)v[i] - o.v[i] - bw;
      r.v[i] = (uint64_t)d;
      bw = (uint64_t)(d >> 127);
    }
    if (bw) {
      uint64_t c = 0;
      for (int i = 0; i < 4; i++) {
        u128 s = (u128)r.v[i] + P::MOD[i] + c;
        r.v[i] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
      }
    }
    return r;
  }
  Fe neg() const { return zero() - *this; }
  Fe dbl() const { return *this + *this; }
  // SOS Montgomery: t = a*b (512 bit), then 4 reduction steps
  Fe operator*(const Fe& o) const {
    uint64_t t[9] = {0};
    for (int i = 0; i < 4; i++) {
      uint64_t c = 0;
      for (int j = 0; j < 4; j++) {
        u128 s = (u128)v[j] * o.v[i] + t[i + j] + c;
        t[i + j] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
      }
      t[i + 4] = c;
    }
    for (int i = 0; i < 4; i++) {
      uint64_t m = t[i] * P::INV, c = 0;
      for (int j = 0; j < 4; j++) {
        u128 s = (u128)m * P::MOD[j] + t[i + j] + c;
        t[i + j] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
      }
      for (int k = i + 4; c && k < 9; k++) {
        u128 s = (u128)t[k] + c;
        t[k] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
      }
    }
    Fe r{{t[4], t[5], t[6], t[7]}};
    if (t[8] || ge_mod(r.v)) sub_mod_inplace(r.v);
    return r;
  }
  Fe sqr() const { return *this * *this; }
  Fe pow(const uint64_t* e) const {
    Fe acc = one();
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        acc = acc.sqr();
        if ((e[i] >> b) & 1) acc = acc * *this;
      }
    return acc;
  }
  Fe inv() const {  // zero -> zero
    uint64_t e[4] = {P::MOD[0] - 2, P::MOD[1], P::MOD[2], P::MOD[3]};
    return pow(e);
  }
  void to_raw(uint64_t* out) const {
    Fe one_raw{{1, 0, 0, 0}};
    Fe c = *this * one_raw;
    memcpy(out, c.v, 32);
  }
};

typedef Fe<FrP> Fr;
typedef Fe<FqP> Fq;

// ---------------------------------------------------------------- G1: y^2 = x^3 + 3
struct Affine {
  Fq x, y;
  bool is_identity() const { return x.is_zero() && y.is_zero(); }  // halo2curves: identity = (0, 0)
};
struct Jac {
  Fq x, y, z;
  static Jac identity() { return Jac{Fq::zero(), Fq::one(), Fq::zero()}; }
  bool is_identity() const { return z.is_zero(); }
};

inline Jac jac_from_affine(const Affine& a) {
  if (a.is_identity()) return Jac::identity();
  return Jac{a.x, a.y, Fq::one()};
}
inline Jac jac_dbl(const Jac& p) {  // dbl-2009-l
  if (p.is_identity()) return p;
  Fq a = p.x.sqr(), b = p.y.sqr(), c = b.sqr();
  Fq d = ((p.x + b).sqr() - a - c).dbl();
  Fq e = a.dbl() + a, f = e.sqr();
  Jac r;
  r.x = f - d.dbl();
  r.y = e * (d - r.x) - c.dbl().dbl().dbl();
  r.z = (p.y * p.z).dbl();
  return r;
}
inline Jac jac_add(const Jac& p, const Jac& q) {  // add-2007-bl
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fq z1z1 = p.z.sqr(), z2z2 = q.z.sqr();
  Fq u1 = p.x * z2z2, u2 = q.x * z1z1;
  Fq s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
  if (u1 == u2) return s1 == s2 ? jac_dbl(p) : Jac::identity();
  Fq h = u2 - u1, i = h.dbl().sqr(), j = h * i, rr = (s2 - s1).dbl(), v = u1 * i;
  Jac r;
  r.x = rr.sqr() - j - v.dbl();
  r.y = rr * (v - r.x) - (s1 * j).dbl();
  r.z = ((p.z + q.z).sqr() - z1z1 - z2z2) * h;
  return r;
}
inline Jac jac_add_affine(const Jac& p, const Affine& q) {  // madd-2007-bl
  if (q.is_identity()) return p;
  if (p.is_identity()) return jac_from_affine(q);
  Fq z1z1 = p.z.sqr();
  Fq u2 = q.x * z1z1, s2 = q.y * p.z * z1z1;
  if (p.x == u2) return p.y == s2 ? jac_dbl(p) : Jac::identity();
  Fq h = u2 - p.x, hh = h.sqr(), i = hh.dbl().dbl(), j = h * i, rr = (s2 - p.y).dbl(), v = p.x * i;
  Jac r;
  r.x = rr.sqr() - j - v.dbl();
  r.y = rr * (v - r.x) - (p.y * j).dbl();
  r.z = (p.z + h).sqr() - z1z1 - hh;
  return r;
}
inline Affine jac_to_affine(const Jac& p) {
  if (p.is_identity()) return Affine{Fq::zero(), Fq::zero()};
  Fq zi = p.z.inv(), zi2 = zi.sqr();
  return Affine{p.x * zi2, p.y * zi2 * zi};
}

}  // namespace orc
