// BN254 G2 and the optimal ate pairing on the host, for the verifier half of the PCS
// (reference util/arithmetic.rs:24-33 `pairings_product_is_identity` over halo2_curves bn256, call site
// pcs/multilinear/kzg.rs:330-361).  Verification is a few dozen pairings per proof and never touches the GPU.
//
// Tower: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9 + u.
// Twist E': y^2 = x^3 + 3/xi, untwist (x, y) -> (x w^2, y w^3).  Only "product of pairings == 1" is
// observable, so lines are affine (denominators of all pairs inverted together per step), subfield
// factors are dropped, and the final exponentiation is conj/inv + one Frobenius + one fixed power.
#pragma once
#include <utility>
#include <vector>
#include "ff_host.hpp"

namespace lh {
namespace host {

// ------------------------------------------------------------------ Fq2
struct Fq2 {
  Fq c0, c1;
  static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool operator!=(const Fq2& o) const { return !(*this == o); }
  Fq2 operator+(const Fq2& o) const { return Fq2{c0 + o.c0, c1 + o.c1}; }
  Fq2 operator-(const Fq2& o) const { return Fq2{c0 - o.c0, c1 - o.c1}; }
  Fq2 operator-() const { return Fq2{-c0, -c1}; }
  Fq2 operator*(const Fq2& o) const {
    Fq a = c0 * o.c0, b = c1 * o.c1;
    return Fq2{a - b, (c0 + c1) * (o.c0 + o.c1) - a - b};
  }
  Fq2 scale(const Fq& k) const { return Fq2{c0 * k, c1 * k}; }
  Fq2 sqr() const { return Fq2{(c0 + c1) * (c0 - c1), (c0 * c1).dbl()}; }
  Fq2 dbl() const { return Fq2{c0.dbl(), c1.dbl()}; }
  Fq2 conj() const { return Fq2{c0, -c1}; }
  Fq2 inv() const {
    Fq n = (c0.sqr() + c1.sqr()).inv();
    return Fq2{c0 * n, -(c1 * n)};
  }
  Fq2 mul_xi() const {  // * (9 + u)
    Fq a8 = c0.dbl().dbl().dbl(), b8 = c1.dbl().dbl().dbl();
    return Fq2{a8 + c0 - c1, b8 + c1 + c0};
  }
  Fq2 pow(const uint64_t* e, int limbs) const {
    Fq2 acc = one();
    for (int i = limbs - 1; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        acc = acc.sqr();
        if ((e[i] >> b) & 1) acc = acc * *this;
      }
    return acc;
  }
};

// ------------------------------------------------------------------ Fq6, Fq12
struct Fq6 {
  Fq2 c0, c1, c2;
  static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
  static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
  bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
  Fq6 operator+(const Fq6& o) const { return Fq6{c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
  Fq6 operator-(const Fq6& o) const { return Fq6{c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
  Fq6 operator-() const { return Fq6{-c0, -c1, -c2}; }
  Fq6 operator*(const Fq6& o) const {
    Fq2 t0 = c0 * o.c0, t1 = c1 * o.c1, t2 = c2 * o.c2;
    Fq2 r0 = ((c1 + c2) * (o.c1 + o.c2) - t1 - t2).mul_xi() + t0;
    Fq2 r1 = (c0 + c1) * (o.c0 + o.c1) - t0 - t1 + t2.mul_xi();
    Fq2 r2 = (c0 + c2) * (o.c0 + o.c2) - t0 - t2 + t1;
    return Fq6{r0, r1, r2};
  }
  Fq6 mul_v() const { return Fq6{c2.mul_xi(), c0, c1}; }
  Fq6 inv() const {
    Fq2 t0 = c0.sqr() - (c1 * c2).mul_xi();
    Fq2 t1 = c2.sqr().mul_xi() - c0 * c1;
    Fq2 t2 = c1.sqr() - c0 * c2;
    Fq2 n = (c0 * t0 + (c2 * t1 + c1 * t2).mul_xi()).inv();
    return Fq6{t0 * n, t1 * n, t2 * n};
  }
};

struct Fq12 {
  Fq6 c0, c1;  // c0 + c1 w
  static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
  bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
  Fq12 operator*(const Fq12& o) const {
    Fq6 a = c0 * o.c0, b = c1 * o.c1;
    return Fq12{a + b.mul_v(), (c0 + c1) * (o.c0 + o.c1) - a - b};
  }
  Fq12 sqr() const { return *this * *this; }
  Fq12 conj() const { return Fq12{c0, -c1}; }  // the q^6 Frobenius
  Fq12 inv() const {
    Fq6 n = (c0 * c0 - (c1 * c1).mul_v()).inv();
    return Fq12{c0 * n, -(c1 * n)};
  }
  // coefficient of w^k, k = 0..5 (w^0: c0.c0, w^1: c1.c0, w^2: c0.c1, w^3: c1.c1, w^4: c0.c2, w^5: c1.c2)
  Fq2& coeff(int k) {
    Fq6& h = (k & 1) ? c1 : c0;
    return (k >> 1) == 0 ? h.c0 : (k >> 1) == 1 ? h.c1 : h.c2;
  }
  Fq12 pow(const uint64_t* e, int limbs) const {
    Fq12 acc = one();
    bool started = false;
    for (int i = limbs - 1; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        if (started) acc = acc.sqr();
        if ((e[i] >> b) & 1) {
          acc = started ? acc * *this : *this;
          started = true;
        }
      }
    return acc;
  }
};

// ------------------------------------------------------------------ constants derived at start-up
struct PairingConsts {
  Fq gamma[6];     // gamma^k, gamma = xi^((q^2-1)/6) = 82^((q-1)/6) in Fq
  Fq2 frob_x, frob_y;  // xi^((q-1)/3), xi^((q-1)/2)
  Fq2 b2;          // 3 / xi
  PairingConsts() {
    // (q - 1) / d on 4 limbs
    auto div_small = [](const uint64_t* a, uint64_t d, uint64_t* out) {
      u128 rem = 0;
      for (int i = 3; i >= 0; i--) {
        u128 cur = (rem << 64) | a[i];
        out[i] = (uint64_t)(cur / d);
        rem = cur % d;
      }
    };
    uint64_t qm1[4] = {FqTag::MOD[0] - 1, FqTag::MOD[1], FqTag::MOD[2], FqTag::MOD[3]};
    uint64_t e6[4], e3[4], e2[4];
    div_small(qm1, 6, e6);
    div_small(qm1, 3, e3);
    div_small(qm1, 2, e2);
    Fq g = Fq::from_u64(82).pow(e6);
    gamma[0] = Fq::one();
    for (int k = 1; k < 6; k++) gamma[k] = gamma[k - 1] * g;
    Fq2 xi{Fq::from_u64(9), Fq::one()};
    frob_x = xi.pow(e3, 4);
    frob_y = xi.pow(e2, 4);
    b2 = Fq2{Fq::from_u64(3), Fq::zero()} * xi.inv();
  }
};
inline const PairingConsts& pairing_consts() {
  static const PairingConsts c;
  return c;
}

inline Fq12 frobenius_q2(Fq12 a) {  // a^(q^2): the coefficient of w^k picks up gamma^k
  const PairingConsts& pc = pairing_consts();
  for (int k = 1; k < 6; k++) a.coeff(k) = a.coeff(k).scale(pc.gamma[k]);
  return a;
}

// ------------------------------------------------------------------ G2
struct G2Affine {
  Fq2 x, y;
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
  bool operator==(const G2Affine& o) const { return x == o.x && y == o.y; }
};
inline G2Affine g2_generator() {
  // the bn256 / EIP-197 generator, canonical little-endian limbs
  static const uint64_t X0[4] = {0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull};
  static const uint64_t X1[4] = {0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull};
  static const uint64_t Y0[4] = {0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull};
  static const uint64_t Y1[4] = {0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull};
  return G2Affine{Fq2{Fq::from_canonical(X0), Fq::from_canonical(X1)}, Fq2{Fq::from_canonical(Y0), Fq::from_canonical(Y1)}};
}
inline bool g2_is_on_curve(const G2Affine& p) {
  if (p.is_identity()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + pairing_consts().b2;
}
inline G2Affine g2_neg(const G2Affine& p) { return p.is_identity() ? p : G2Affine{p.x, -p.y}; }

struct G2Xyzz {  // same coordinates as G1Xyzz, over Fq2
  Fq2 x, y, zz, zzz;
  static G2Xyzz identity() { return G2Xyzz{Fq2::zero(), Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
  bool is_identity() const { return zz.is_zero(); }
};
inline G2Xyzz g2_dbl(const G2Xyzz& p) {
  if (p.is_identity() || p.y.is_zero()) return G2Xyzz::identity();
  Fq2 u = p.y.dbl(), v = u.sqr(), w = u * v, s = p.x * v, xx = p.x.sqr();
  Fq2 m = xx.dbl() + xx;
  G2Xyzz r;
  r.x = m.sqr() - s.dbl();
  r.y = m * (s - r.x) - w * p.y;
  r.zz = v * p.zz;
  r.zzz = w * p.zzz;
  return r;
}
inline G2Xyzz g2_add(const G2Xyzz& p, const G2Xyzz& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fq2 u1 = p.x * q.zz, u2 = q.x * p.zz, s1 = p.y * q.zzz, s2 = q.y * p.zzz;
  Fq2 pp_ = u2 - u1, r_ = s2 - s1;
  if (pp_.is_zero()) return r_.is_zero() ? g2_dbl(p) : G2Xyzz::identity();
  Fq2 pp = pp_.sqr(), ppp = pp_ * pp, qq = u1 * pp;
  G2Xyzz r;
  r.x = r_.sqr() - ppp - qq.dbl();
  r.y = r_ * (qq - r.x) - s1 * ppp;
  r.zz = p.zz * q.zz * pp;
  r.zzz = p.zzz * q.zzz * ppp;
  return r;
}
inline G2Xyzz g2_from_affine(const G2Affine& a) {
  if (a.is_identity()) return G2Xyzz::identity();
  return G2Xyzz{a.x, a.y, Fq2::one(), Fq2::one()};
}
inline G2Affine g2_to_affine(const G2Xyzz& p) {
  if (p.is_identity()) return G2Affine{Fq2::zero(), Fq2::zero()};
  Fq2 i = (p.zz * p.zzz).inv();
  return G2Affine{p.x * (i * p.zzz), p.y * (i * p.zz)};
}
inline G2Xyzz g2_mul(const G2Xyzz& p, const Fr& k) {
  uint64_t c[4];
  k.to_canonical(c);
  G2Xyzz acc = G2Xyzz::identity();
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      acc = g2_dbl(acc);
      if ((c[i] >> b) & 1) acc = g2_add(acc, p);
    }
  return acc;
}

// ------------------------------------------------------------------ Miller loop, final exponentiation
// line through the untwisted T (and U, or tangent) at P = (xp, yp), up to a factor in Fq2:
//   yp - lambda xp w + (lambda x_T - y_T) w^3
inline Fq12 line_eval(const Fq2& lambda, const G2Affine& t, const G1Affine& p) {
  Fq12 l{Fq6::zero(), Fq6::zero()};
  l.c0.c0 = Fq2{p.y, Fq::zero()};
  l.c1.c0 = -(lambda.scale(p.x));
  l.c1.c1 = lambda * t.x - t.y;
  return l;
}

// prod_i f_{6x+2,Q_i}(P_i) l(P_i) l'(P_i); identity members contribute 1
inline Fq12 multi_miller_loop(const std::vector<std::pair<G1Affine, G2Affine>>& pairs_in) {
  static const uint64_t ATE = 0x9d797039be763ba8ull;  // 6x + 2 = 29793968203157093288 needs 65 bits: see below
  // 29793968203157093288 = 0x1_9d797039be763ba8: bit 64 is the leading one
  const PairingConsts& pc = pairing_consts();
  std::vector<std::pair<G1Affine, G2Affine>> pairs;
  for (auto& pr : pairs_in)
    if (!pr.first.is_identity() && !pr.second.is_identity()) pairs.push_back(pr);
  const size_t m = pairs.size();
  Fq12 f = Fq12::one();
  if (!m) return f;
  std::vector<G2Affine> T(m);
  for (size_t i = 0; i < m; i++) T[i] = pairs[i].second;
  std::vector<Fq2> den(m), lam(m), pref(m);
  // batch inversion of den[] into lam[] = num[] / den[]
  auto slopes = [&](std::vector<Fq2>& num) {
    Fq2 acc = Fq2::one();
    for (size_t i = 0; i < m; i++) {
      pref[i] = acc;
      acc = acc * den[i];
    }
    Fq2 inv = acc.inv();
    for (size_t i = m; i-- > 0;) {
      lam[i] = num[i] * (inv * pref[i]);
      inv = inv * den[i];
    }
  };
  std::vector<Fq2> num(m);
  auto dbl_step = [&]() {
    for (size_t i = 0; i < m; i++) {
      Fq2 xx = T[i].x.sqr();
      num[i] = xx.dbl() + xx;
      den[i] = T[i].y.dbl();
    }
    slopes(num);
    for (size_t i = 0; i < m; i++) {
      f = f * line_eval(lam[i], T[i], pairs[i].first);
      Fq2 x3 = lam[i].sqr() - T[i].x.dbl();
      T[i] = G2Affine{x3, lam[i] * (T[i].x - x3) - T[i].y};
    }
  };
  auto add_step = [&](const std::vector<G2Affine>& Q) {
    for (size_t i = 0; i < m; i++) {
      num[i] = Q[i].y - T[i].y;
      den[i] = Q[i].x - T[i].x;
    }
    slopes(num);
    for (size_t i = 0; i < m; i++) {
      f = f * line_eval(lam[i], T[i], pairs[i].first);
      Fq2 x3 = lam[i].sqr() - T[i].x - Q[i].x;
      T[i] = G2Affine{x3, lam[i] * (T[i].x - x3) - T[i].y};
    }
  };
  std::vector<G2Affine> Q(m);
  for (size_t i = 0; i < m; i++) Q[i] = pairs[i].second;
  for (int b = 63; b >= 0; b--) {  // bits below the leading one (bit 64)
    f = f.sqr();
    dbl_step();
    if ((ATE >> b) & 1) add_step(Q);
  }
  std::vector<G2Affine> Q1(m), nQ2(m);
  const Fq g2 = pc.gamma[2], g3 = pc.gamma[3];
  for (size_t i = 0; i < m; i++) {
    Q1[i] = G2Affine{Q[i].x.conj() * pc.frob_x, Q[i].y.conj() * pc.frob_y};
    nQ2[i] = G2Affine{Q[i].x.scale(g2), -(Q[i].y.scale(g3))};
  }
  add_step(Q1);
  add_step(nQ2);
  return f;
}

inline Fq12 final_exponentiation(const Fq12& f) {
  // (q^4 - q^2 + 1) / r, little-endian limbs
  static const uint64_t HARD[12] = {
      0xe81bb482ccdf42b1ull, 0x5abf5cc4f49c36d4ull, 0xf1154e7e1da014fdull, 0xdcc7b44c87cdbacfull,
      0xaaa441e3954bcf8aull, 0x6b887d56d5095f23ull, 0x79581e16f3fd90c6ull, 0x3b1b1355d189227dull,
      0x4e529a5861876f6bull, 0x6c0eb522d5b12278ull, 0x331ec15183177fafull, 0x01baaa710b0759adull};
  Fq12 f1 = f.conj() * f.inv();        // f^(q^6 - 1)
  Fq12 f2 = frobenius_q2(f1) * f1;     // ^(q^2 + 1)
  return f2.pow(HARD, 12);
}

inline Fq12 pairing(const G1Affine& p, const G2Affine& q) {
  return final_exponentiation(multi_miller_loop({{p, q}}));
}

// util/arithmetic.rs:24-33
inline bool pairings_product_is_identity(const std::vector<std::pair<G1Affine, G2Affine>>& pairs) {
  return final_exponentiation(multi_miller_loop(pairs)) == Fq12::one();
}

}  // namespace host
}  // namespace lh
