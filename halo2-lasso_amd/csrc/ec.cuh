// BN254 G1 (y^2 = x^3 + 3) point arithmetic in extended-Jacobian XYZZ coordinates.
//
// Replaces halo2_curves 0.3.3 bn256::{G1Affine,G1} mixed add / double as used by the reference
// MSM (plonkish_backend/src/util/arithmetic/msm.rs:129-179).  Only the affine value of a sum is
// observable, so the coordinate system is free: XYZZ needs 8M+2S for a mixed add
// against 7M+4S+more adds for Jacobian and has no inversion.
//   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 ; identity <=> ZZ == 0 (all-zero bytes are the identity,
//   so hipMemset(0) initialises bucket arrays).  Affine identity is (0,0) as in halo2curves.
#pragma once
#include "ff.cuh"

namespace lh {

struct alignas(16) G1Affine {
  Fq x, y;
  LH_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

struct alignas(16) G1Xyzz {
  Fq x, y, zz, zzz;
  static LH_HD G1Xyzz identity() {
    G1Xyzz r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.zz = Fq::zero();
    r.zzz = Fq::zero();
    return r;
  }
  static LH_HD G1Xyzz from_affine(const G1Affine& p) {
    G1Xyzz r;
    if (p.is_identity()) return identity();
    r.x = p.x;
    r.y = p.y;
    r.zz = Fq::one();
    r.zzz = Fq::one();
    return r;
  }
  LH_HD bool is_identity() const { return zz.is_zero(); }
};

// 2*P for affine P (mdbl-2008-s-1, a = 0)
LH_HD G1Xyzz dbl_affine(const G1Affine& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
  Fq u = dbl(p.y);
  Fq v = sqr(u);
  Fq w = mul(u, v);
  Fq s = mul(p.x, v);
  Fq xx = sqr(p.x);
  Fq m = add(dbl(xx), xx);
  G1Xyzz r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// 2*P (dbl-2008-s-1, a = 0)
LH_HD G1Xyzz dbl(const G1Xyzz& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
  Fq u = dbl(p.y);
  Fq v = sqr(u);
  Fq w = mul(u, v);
  Fq s = mul(p.x, v);
  Fq xx = sqr(p.x);
  Fq m = add(dbl(xx), xx);
  G1Xyzz r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
  r.zz = mul(v, p.zz);
  r.zzz = mul(w, p.zzz);
  return r;
}

// P + Q, Q affine (madd-2008-s); `negate` adds -Q.
LH_HD G1Xyzz add_mixed(const G1Xyzz& p, const G1Affine& q_in, bool negate = false) {
  if (q_in.is_identity()) return p;
  G1Affine q = q_in;
  if (negate) q.y = neg(q.y);
  if (p.is_identity()) return G1Xyzz::from_affine(q);
  Fq u2 = mul(q.x, p.zz);
  Fq s2 = mul(q.y, p.zzz);
  Fq pp_ = sub(u2, p.x);
  Fq r_ = sub(s2, p.y);
  if (pp_.is_zero()) {
    if (r_.is_zero()) return dbl_affine(q);
    return G1Xyzz::identity();
  }
  Fq pp = sqr(pp_);
  Fq ppp = mul(pp_, pp);
  Fq qq = mul(p.x, pp);
  G1Xyzz r;
  r.x = sub(sub(sqr(r_), ppp), dbl(qq));
  r.y = sub(mul(r_, sub(qq, r.x)), mul(p.y, ppp));
  r.zz = mul(p.zz, pp);
  r.zzz = mul(p.zzz, ppp);
  return r;
}

// P + Q (add-2008-s)
LH_HD G1Xyzz add(const G1Xyzz& p, const G1Xyzz& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fq u1 = mul(p.x, q.zz);
  Fq u2 = mul(q.x, p.zz);
  Fq s1 = mul(p.y, q.zzz);
  Fq s2 = mul(q.y, p.zzz);
  Fq pp_ = sub(u2, u1);
  Fq r_ = sub(s2, s1);
  if (pp_.is_zero()) {
    if (r_.is_zero()) return dbl(p);
    return G1Xyzz::identity();
  }
  Fq pp = sqr(pp_);
  Fq ppp = mul(pp_, pp);
  Fq qq = mul(u1, pp);
  G1Xyzz r;
  r.x = sub(sub(sqr(r_), ppp), dbl(qq));
  r.y = sub(mul(r_, sub(qq, r.x)), mul(s1, ppp));
  r.zz = mul(mul(p.zz, q.zz), pp);
  r.zzz = mul(mul(p.zzz, q.zzz), ppp);
  return r;
}

}  // namespace lh
