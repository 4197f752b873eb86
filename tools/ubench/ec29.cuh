// BN254 G1 in XYZZ coordinates over the 9 x 29-bit lazy-carry field form (ff29.cuh): the mixed addition of the MSM's bucket
// accumulation (ec.cuh add_mixed, msm.rs:129-179) with ~205-instruction products, limb-wise additions and no comparison
// against the modulus inside the formulas.  Values are Montgomery with radix 2^261; bounds (p = the base-field modulus):
//   bases canonical (< p); accumulator X, Y < 16 p, ZZ, ZZZ < 3 p; every product < 3 p.
// The accumulated point leaves through to_xyzz (one product per coordinate back to radix 2^256, canonical).
#pragma once
#include "ec.cuh"
#include "ff29.cuh"

namespace lh {

// a base point in radix 2^261, canonical: kept in memory PACKED (64 bytes, the layout of G1Affine - a 72-byte point
// straddles cache lines and made the gathers of the accumulation slower than the arithmetic got faster) and re-sliced into
// limbs when loaded (~40 instructions per point against ~2300 of the addition)
struct alignas(16) G1Affine29 {
  Fq x, y;  // identity (0, 0)
};
struct G1Xyzz29 {
  Fq29 x, y, zz, zzz;
};

LH_HD bool is_zero_limbs29(const Fq29& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= a.l[i];
  return o == 0;
}
// a value known to be < 3 p is 0 mod p iff it is 0, p or 2 p: the low limb decides almost always
LH_HD bool is_zero_mod29(const Fq29& a) {
  const uint32_t l0 = a.l[0];
  if (l0 != 0 && l0 != kmod29<FqParams, 1>(0) && l0 != kmod29<FqParams, 2>(0)) return false;
  bool z0 = true, z1 = true, z2 = true;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    z0 = z0 && a.l[i] == 0;
    z1 = z1 && a.l[i] == kmod29<FqParams, 1>(i);
    z2 = z2 && a.l[i] == kmod29<FqParams, 2>(i);
  }
  return z0 || z1 || z2;
}

// radix 2^256 canonical (ff.cuh Fq) -> radix 2^261 canonical limbs: five modular doublings, then re-slicing
LH_HD Fq29 to29(const Fq& x) {
  Fq v = x;
#pragma unroll
  for (int k = 0; k < 5; k++) v = add(v, v);
  return slice29(v);
}
// the packed radix-2^261 form of a canonical radix-2^256 value: five modular doublings
LH_HD Fq to261(const Fq& x) {
  Fq v = x;
#pragma unroll
  for (int k = 0; k < 5; k++) v = add(v, v);
  return v;
}
#if defined(__HIPCC__)
// radix 2^261 (< 16 p) -> radix 2^256 canonical: one product with the plain integer 2^256 mod p divides by 2^5
__device__ __forceinline__ Fq from29(const Fq29& v) {
  const Fq29 r = mul29(v, slice29(Fq::one()));  // < 1.1 p
  return reduce_once(unslice29(r));
}
__device__ __forceinline__ G1Xyzz to_xyzz(const G1Xyzz29& p) {
  G1Xyzz r;
  if (is_zero_limbs29(p.zz)) return G1Xyzz::identity();
  r.x = from29(p.x), r.y = from29(p.y), r.zz = from29(p.zz), r.zzz = from29(p.zzz);
  return r;
}
__device__ __forceinline__ G1Xyzz29 from_xyzz(const G1Xyzz& p) {
  G1Xyzz29 r;
  r.x = to29(p.x), r.y = to29(p.y), r.zz = to29(p.zz), r.zzz = to29(p.zzz);
  return r;
}
__device__ __forceinline__ G1Xyzz29 identity29() {
  G1Xyzz29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
  return r;
}

// P + Q, Q affine (madd-2008-s); `negate` adds -Q.  P: X, Y < 16 p, ZZ, ZZZ < 3 p (identity: ZZ all-zero limbs).
__device__ __forceinline__ G1Xyzz29 add_mixed29(const G1Xyzz29& p, const G1Affine29& q_in, bool negate) {
  typedef FqParams P;
  if (q_in.x.is_zero() && q_in.y.is_zero()) return p;
  const Fq29 qx = slice29(q_in.x);
  Fq29 qy = slice29(q_in.y);
  if (negate) {  // p - y (y canonical)
    Fq29 z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.l[i] = 0;
    qy = sub29<P, 1>(z, qy);
  }
  if (is_zero_limbs29(p.zz)) {
    G1Xyzz29 r;
    r.x = qx, r.y = qy;
    r.zz = r.zzz = to29(Fq::one());
    return r;
  }
  const Fq29 u2 = mul29(qx, p.zz), s2 = mul29(qy, p.zzz);              // < 1.1 p
  const Fq29 pp_ = sub29<P, 16>(u2, p.x), r_ = sub29<P, 16>(s2, p.y);  // < 17.1 p
  const Fq29 pp = mul29(pp_, pp_);                                     // < 2.8 p
  if (is_zero_mod29(pp)) {  // the same x: a doubling or the identity (rare: through the standard form)
    const Fq29 rr0 = mul29(r_, r_);
    if (!is_zero_mod29(rr0)) return identity29();
    G1Affine q;
    q.x = from29(qx), q.y = from29(qy);
    return from_xyzz(dbl_affine(q));
  }
  const Fq29 ppp = mul29(pp_, pp), qq = mul29(p.x, pp), rr = mul29(r_, r_);  // < 1.3 p, < 1.3 p, < 2.8 p
  G1Xyzz29 r;
  r.x = sub29<P, 4>(sub29<P, 2>(rr, ppp), add29(qq, qq));                      // < 8.8 p
  r.y = sub29<P, 2>(mul29(r_, sub29<P, 16>(qq, r.x)), mul29(p.y, ppp));        // < 4.8 p
  r.zz = mul29(p.zz, pp);
  r.zzz = mul29(p.zzz, ppp);
  return r;
}
#endif

}  // namespace lh
