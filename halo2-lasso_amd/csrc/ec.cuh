// BN254 G1 (y^2 = x^3 + 3) point arithmetic in extended-Jacobian XYZZ coordinates.
//
// Replaces halo2_curves 0.3.3 bn256::{G1Affine,G1} mixed add / double as used by the reference
// MSM (plonkish_backend/src/util/arithmetic/msm.rs:129-179).  Only the affine value of a sum is
// observable, so the coordinate system is free: XYZZ needs 8M+2S for a mixed add
// against 7M+4S+more adds for Jacobian and has no inversion.
//   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 ; identity <=> ZZ == 0 (all-zero bytes are the identity,
//   so hipMemset(0) initialises bucket arrays).  Affine identity is (0,0) as in halo2curves.
#pragma once
#if defined(__HIPCC__) && !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>  // threadIdx in the quad-cooperative routines
#endif
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

// a c - b d: on the device the two products share ONE Montgomery reduction (ff.cuh dot_scan: 193 multiply-adds instead
// of 258; every Y3 of the XYZZ formulas has this shape)
LH_HD Fq mul_sub_mul(const Fq& a, const Fq& c, const Fq& b, const Fq& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  const Fq x[2] = {a, neg(b)}, y[2] = {c, d};
  return dot<FqParams, 2>(x, y);
#else
  return sub(mul(a, c), mul(b, d));
#endif
}

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
  r.y = mul_sub_mul(m, sub(s, r.x), w, p.y);
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
  r.y = mul_sub_mul(m, sub(s, r.x), w, p.y);
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
  r.y = mul_sub_mul(r_, sub(qq, r.x), p.y, ppp);
  r.zz = mul(p.zz, pp);
  r.zzz = mul(p.zzz, ppp);
  return r;
}

#if defined(__HIPCC__)
// P + Q, Q affine and canonical, P with LAZY coordinates (every coordinate in [0, 2 q), ff.cuh): the accumulator of the
// MSM's bucket accumulation never leaves the lazy range between two flushes, and 8 of the 10 products of the formulas
// run without their final conditional subtraction (-4 % of the addition's instructions).  The result is lazy too;
// canon_xyzz makes it canonical (before it is stored: everything else in the library expects canonical coordinates).
__device__ __forceinline__ G1Xyzz add_mixed_lazy(const G1Xyzz& p, const G1Affine& q_in, bool negate) {
  if (q_in.is_identity()) return p;
  G1Affine q = q_in;
  if (negate) q.y = neg(q.y);
  if (p.is_identity()) return G1Xyzz::from_affine(q);  // (ZZ = 0 exactly: only the identity is ever given a zero ZZ)
  const Fq u2 = mul_lazy(q.x, p.zz);
  const Fq s2 = mul_lazy(q.y, p.zzz);
  const Fq pp_ = sub_lazy(u2, p.x);
  const Fq r_ = sub_lazy(s2, p.y);
  if (is_zero_lazy(pp_)) {
    if (is_zero_lazy(r_)) return dbl_affine(q);
    return G1Xyzz::identity();
  }
  const Fq pp = mul_lazy(pp_, pp_);
  const Fq ppp = mul_lazy(pp_, pp);
  const Fq qq = mul_lazy(p.x, pp);
  G1Xyzz r;
  // (2 Q as two subtractions: the subtraction takes 2 q as literals, the addition would hold its limbs in registers - and the
  // kernel is one wave of occupancy from the edge)
  r.x = sub_lazy(sub_lazy(sub_lazy(mul_lazy(r_, r_), ppp), qq), qq);
  {
    // R (Q - X3) - Y PPP with one Montgomery reduction; the negated factor 2 q - Y lies in (0, 2 q]: the dot product stays
    // below q (8 q / R + 1) < 2.51 q, one conditional subtraction of q leaves it below 1.51 q
    const Fq x[2] = {r_, sub_lazy(Fq::zero(), p.y)}, y[2] = {sub_lazy(qq, r.x), ppp};
    r.y = dot<FqParams, 2>(x, y);
  }
  r.zz = mul_lazy(p.zz, pp);
  r.zzz = mul_lazy(p.zzz, ppp);
  return r;
}
__device__ __forceinline__ G1Xyzz canon_xyzz(const G1Xyzz& p) {
  G1Xyzz r;
  r.x = canon(p.x), r.y = canon(p.y), r.zz = canon(p.zz), r.zzz = canon(p.zzz);
  return r;
}
#endif

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
  r.y = mul_sub_mul(r_, sub(qq, r.x), s1, ppp);
  r.zz = mul(mul(p.zz, q.zz), pp);
  r.zzz = mul(mul(p.zzz, q.zzz), ppp);
  return r;
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------ quad-cooperative arithmetic (device only)
// The tails of an MSM (late continuation levels, small bucket reductions) are chains of DEPENDENT additions on
// launches far below one wave per SIMD: a lone lane multiplies at ~1 us per Montgomery product, so one addition
// (14 products) costs ~14 us however idle the machine is.  Here the 4 lanes of a quad (lanes 4k..4k+3) hold the
// SAME operands and share the work: the products of one formula step run in different lanes and are exchanged with
// DPP quad_perm moves, 4 product-steps per addition and 3 per doubling instead of 14 and 9.  Every lane of the quad
// returns the full result, so a kernel maps one logical thread to a quad and keeps its control flow (which depends
// only on the replicated data) unchanged.
template <int K>
__device__ __forceinline__ Fq quad_bcast(const Fq& v) {
  Fq o;
#pragma unroll
  for (int i = 0; i < 8; i++)
    o.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], K * 0x55, 0xf, 0xf, false);
  return o;
}

// lane-dependent operand choice as mask arithmetic: (x & m) | (y & ~m) is one v_bfi_b32 (the ternary form is turned
// into a branch per limb)
__device__ __forceinline__ Fq quad_sel(int ql, const Fq& a, const Fq& b, const Fq& c, const Fq& d) {
  const uint32_t lo = 0u - (uint32_t)(ql & 1), hi = 0u - (uint32_t)((ql >> 1) & 1);
  Fq o;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t t0 = (b.l[i] & lo) | (a.l[i] & ~lo), t1 = (d.l[i] & lo) | (c.l[i] & ~lo);
    o.l[i] = (t1 & hi) | (t0 & ~hi);
  }
  return o;
}

__device__ __forceinline__ G1Xyzz dbl_quad(const G1Xyzz& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
  const int ql = (int)(threadIdx.x & 3u);
  const Fq u = dbl(p.y);
  Fq t = quad_sel(ql, u, p.x, u, u);
  Fq m = mul(t, t);                                              // 0: V = U^2   1: XX = X^2
  const Fq v = quad_bcast<0>(m), xx = quad_bcast<1>(m);
  const Fq mm = add(dbl(xx), xx);
  m = mul(quad_sel(ql, u, p.x, mm, v), quad_sel(ql, v, v, mm, p.zz));   // 0: W = U V  1: S = X V  2: M^2  3: V ZZ
  const Fq w = quad_bcast<0>(m), s_ = quad_bcast<1>(m), m2 = quad_bcast<2>(m);
  G1Xyzz r;
  r.zz = quad_bcast<3>(m);
  r.x = sub(m2, dbl(s_));
  m = mul(quad_sel(ql, mm, w, w, w), quad_sel(ql, sub(s_, r.x), p.y, p.zzz, p.zzz));  // 0: M (S - X3)  1: W Y  2: W ZZZ
  r.y = sub(quad_bcast<0>(m), quad_bcast<1>(m));
  r.zzz = quad_bcast<2>(m);
  return r;
}

__device__ __forceinline__ G1Xyzz add_quad(const G1Xyzz& p, const G1Xyzz& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  const int ql = (int)(threadIdx.x & 3u);
  Fq m = mul(quad_sel(ql, p.x, q.x, p.y, q.y), quad_sel(ql, q.zz, p.zz, q.zzz, p.zzz));
  const Fq u1 = quad_bcast<0>(m), u2 = quad_bcast<1>(m), s1 = quad_bcast<2>(m), s2 = quad_bcast<3>(m);
  const Fq pp_ = sub(u2, u1), r_ = sub(s2, s1);
  if (pp_.is_zero()) {
    if (r_.is_zero()) return dbl_quad(p);
    return G1Xyzz::identity();
  }
  m = mul(quad_sel(ql, pp_, r_, p.zz, p.zzz), quad_sel(ql, pp_, r_, q.zz, q.zzz));   // PP, RR, ZZ1 ZZ2, ZZZ1 ZZZ2
  const Fq pp = quad_bcast<0>(m), rr = quad_bcast<1>(m), zz12 = quad_bcast<2>(m), zzz12 = quad_bcast<3>(m);
  m = mul(quad_sel(ql, pp_, u1, zz12, zz12), pp);                                        // PPP, Q, ZZ3
  const Fq ppp = quad_bcast<0>(m), qq = quad_bcast<1>(m);
  G1Xyzz r;
  r.zz = quad_bcast<2>(m);
  r.x = sub(sub(rr, ppp), dbl(qq));
  m = mul(quad_sel(ql, r_, s1, zzz12, zzz12), quad_sel(ql, sub(qq, r.x), ppp, ppp, ppp));
  r.y = sub(quad_bcast<0>(m), quad_bcast<1>(m));
  r.zzz = quad_bcast<2>(m);
  return r;
}
#endif

}  // namespace lh
