// BN254 prime-field arithmetic for gfx950 (and the host): 8 x u32 limbs, Montgomery R = 2^256.
//
// Replaces halo2_curves 0.3.3 bn256::{Fr,Fq} as used by the reference through
// plonkish_backend/src/util/arithmetic.rs:15-22.  The in-memory form is
// byte-identical to halo2curves' `[u64; 4]` Montgomery representation, so a Rust `&[Fr]` can be
// handed to the C-ABI unchanged.
//
// CDNA4 has no 64-bit integer multiplier: a limb product is one v_mad_u64_u32 (32x32+64->64).
// CIOS keeps the running value in 32-bit limbs so that every mad's addend is (limb + carry).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LH_HD __host__ __device__ __forceinline__
#else
#define LH_HD inline
#endif

namespace lh {

struct FrParams {
  // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
  static LH_HD constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return m[i];
  }
  static LH_HD constexpr uint32_t r1(int i) {  // R mod r
    constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                               0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return m[i];
  }
  static LH_HD constexpr uint32_t r2(int i) {  // R^2 mod r
    constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                               0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    return m[i];
  }
  static constexpr uint32_t INV = 0xefffffffu;  // -r^{-1} mod 2^32
};

struct FqParams {
  // q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
  static LH_HD constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return m[i];
  }
  static LH_HD constexpr uint32_t r1(int i) {
    constexpr uint32_t m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                               0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return m[i];
  }
  static LH_HD constexpr uint32_t r2(int i) {
    constexpr uint32_t m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                               0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    return m[i];
  }
  static constexpr uint32_t INV = 0xe4866389u;  // -q^{-1} mod 2^32
};

// 2 p as a modulus provider (p < 2^254: 2 p fits 8 limbs) - the lazy forms below keep values in [0, 2 p)
template <class P>
struct Twice {
  static LH_HD constexpr uint32_t mod(int i) { return (P::mod(i) << 1) | (i ? P::mod(i - 1) >> 31 : 0u); }
};

template <class P>
struct alignas(16) Fp {
  typedef P params;
  uint32_t l[8];

  static LH_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
  }
  static LH_HD Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::r1(i);
    return r;
  }
  static LH_HD Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::r2(i);
    return r;
  }
  LH_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i];
    return o == 0;
  }
  LH_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
    return o == 0;
  }
  LH_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a + b (no reduction); returns carry
template <class P>
LH_HD uint32_t add_raw(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t s = (uint64_t)a.l[i] + b.l[i] + c;
    r.l[i] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  return c;
}

// r = a - p ; returns borrow (1 if a < p)
template <class P>
LH_HD uint32_t sub_mod_raw(Fp<P>& r, const Fp<P>& a) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t s = (uint64_t)a.l[i] - P::mod(i) - bw;
    r.l[i] = (uint32_t)s;
    bw = (uint32_t)(s >> 63);
  }
  return bw;
}

template <class P>
LH_HD Fp<P> reduce_once_generic(const Fp<P>& a) {
  Fp<P> t;
  uint32_t bw = sub_mod_raw(t, a);
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = bw ? a.l[i] : t.l[i];
  return r;
}

template <class P>
LH_HD Fp<P> add_generic(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> s;
  add_raw(s, a, b);  // a,b < p < 2^254 : no carry out of 256 bits
  return reduce_once_generic(s);
}

template <class P>
LH_HD Fp<P> sub_generic(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> d;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t s = (uint64_t)a.l[i] - b.l[i] - bw;
    d.l[i] = (uint32_t)s;
    bw = (uint32_t)(s >> 63);
  }
  uint32_t mask = 0u - bw;
  uint32_t c = 0;
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t s = (uint64_t)d.l[i] + (P::mod(i) & mask) + c;
    r.l[i] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Device forms: the 8-limb carry chains written as the 8 instructions they are (from the C++ above the compiler makes
// ~100 instructions per modular addition: 64-bit adds on zero-extended register pairs and the moves to build them;
// these are ~25).  Checked against the generic forms by tools/ubench/mul_forms.hip and by every parity test.
template <class P>
__device__ __forceinline__ void add_chain(Fp<P>& s, const Fp<P>& a, const Fp<P>& b) {
  asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
      "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(s.l[0]), "=&v"(s.l[1]), "=&v"(s.l[2]), "=&v"(s.l[3]), "=&v"(s.l[4]), "=&v"(s.l[5]), "=&v"(s.l[6]),
        "=&v"(s.l[7])
      : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
        "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
      : "vcc");
}
// t = s - p, mask = all ones if that borrowed (s < p)
template <class P>
__device__ __forceinline__ uint32_t sub_p_chain(Fp<P>& t, const Fp<P>& s) {
  uint32_t mask;
  asm("v_subrev_co_u32 %0, vcc, %17, %9\n\t"
      "v_subbrev_co_u32 %1, vcc, %18, %10, vcc\n\t"
      "v_subbrev_co_u32 %2, vcc, %19, %11, vcc\n\t"
      "v_subbrev_co_u32 %3, vcc, %20, %12, vcc\n\t"
      "v_subbrev_co_u32 %4, vcc, %21, %13, vcc\n\t"
      "v_subbrev_co_u32 %5, vcc, %22, %14, vcc\n\t"
      "v_subbrev_co_u32 %6, vcc, %23, %15, vcc\n\t"
      "v_subbrev_co_u32 %7, vcc, %24, %16, vcc\n\t"
      "v_cndmask_b32_e64 %8, 0, -1, vcc"
      : "=&v"(t.l[0]), "=&v"(t.l[1]), "=&v"(t.l[2]), "=&v"(t.l[3]), "=&v"(t.l[4]), "=&v"(t.l[5]), "=&v"(t.l[6]),
        "=&v"(t.l[7]), "=&v"(mask)
      : "v"(s.l[0]), "v"(s.l[1]), "v"(s.l[2]), "v"(s.l[3]), "v"(s.l[4]), "v"(s.l[5]), "v"(s.l[6]), "v"(s.l[7]),
        // the modulus in VGPRs: an SGPR operand next to the carry-in would be two constant-bus reads
        "v"(P::mod(0)), "v"(P::mod(1)), "v"(P::mod(2)), "v"(P::mod(3)), "v"(P::mod(4)), "v"(P::mod(5)), "v"(P::mod(6)),
        "v"(P::mod(7))
      : "vcc");
  return mask;
}
// d = a - b, mask = all ones if that borrowed (a < b)
template <class P>
__device__ __forceinline__ uint32_t sub_chain(Fp<P>& d, const Fp<P>& a, const Fp<P>& b) {
  uint32_t mask;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_cndmask_b32_e64 %8, 0, -1, vcc"
      : "=&v"(d.l[0]), "=&v"(d.l[1]), "=&v"(d.l[2]), "=&v"(d.l[3]), "=&v"(d.l[4]), "=&v"(d.l[5]), "=&v"(d.l[6]),
        "=&v"(d.l[7]), "=&v"(mask)
      : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
        "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
      : "vcc");
  return mask;
}
// The same chains fused with what follows them, one asm block each (fewer instructions - the select is a v_cndmask by the
// borrow instead of a mask and three logic operations per limb - and one hazard nop instead of three, see ff_cols.inc):
// r = a < p ? a : a - p                                  (16 instructions)
template <class P, class M = P>
__device__ __forceinline__ void reduce_sel(Fp<P>& r, const Fp<P>& a) {
  asm("v_subrev_co_u32 %0, vcc, %16, %8\n\t"
      "v_subbrev_co_u32 %1, vcc, %17, %9, vcc\n\t"
      "v_subbrev_co_u32 %2, vcc, %18, %10, vcc\n\t"
      "v_subbrev_co_u32 %3, vcc, %19, %11, vcc\n\t"
      "v_subbrev_co_u32 %4, vcc, %20, %12, vcc\n\t"
      "v_subbrev_co_u32 %5, vcc, %21, %13, vcc\n\t"
      "v_subbrev_co_u32 %6, vcc, %22, %14, vcc\n\t"
      "v_subbrev_co_u32 %7, vcc, %23, %15, vcc\n\t"
      "v_cndmask_b32 %0, %0, %8, vcc\n\t"
      "v_cndmask_b32 %1, %1, %9, vcc\n\t"
      "v_cndmask_b32 %2, %2, %10, vcc\n\t"
      "v_cndmask_b32 %3, %3, %11, vcc\n\t"
      "v_cndmask_b32 %4, %4, %12, vcc\n\t"
      "v_cndmask_b32 %5, %5, %13, vcc\n\t"
      "v_cndmask_b32 %6, %6, %14, vcc\n\t"
      "v_cndmask_b32 %7, %7, %15, vcc"
      : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7])
      : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
        "v"(M::mod(0)), "v"(M::mod(1)), "v"(M::mod(2)), "v"(M::mod(3)), "v"(M::mod(4)), "v"(M::mod(5)), "v"(M::mod(6)), "v"(M::mod(7))
      : "vcc");
}
// r = a + b mod p for a, b < p                          (24 instructions)
template <class P, class M = P>
__device__ __forceinline__ void add_sel(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
  Fp<P> t;
  asm("v_add_co_u32 %8, vcc, %16, %24\n\t"
      "v_addc_co_u32 %9, vcc, %17, %25, vcc\n\t"
      "v_addc_co_u32 %10, vcc, %18, %26, vcc\n\t"
      "v_addc_co_u32 %11, vcc, %19, %27, vcc\n\t"
      "v_addc_co_u32 %12, vcc, %20, %28, vcc\n\t"
      "v_addc_co_u32 %13, vcc, %21, %29, vcc\n\t"
      "v_addc_co_u32 %14, vcc, %22, %30, vcc\n\t"
      "v_addc_co_u32 %15, vcc, %23, %31, vcc\n\t"
      "v_subrev_co_u32 %0, vcc, %32, %8\n\t"
      "v_subbrev_co_u32 %1, vcc, %33, %9, vcc\n\t"
      "v_subbrev_co_u32 %2, vcc, %34, %10, vcc\n\t"
      "v_subbrev_co_u32 %3, vcc, %35, %11, vcc\n\t"
      "v_subbrev_co_u32 %4, vcc, %36, %12, vcc\n\t"
      "v_subbrev_co_u32 %5, vcc, %37, %13, vcc\n\t"
      "v_subbrev_co_u32 %6, vcc, %38, %14, vcc\n\t"
      "v_subbrev_co_u32 %7, vcc, %39, %15, vcc\n\t"
      "v_cndmask_b32 %0, %0, %8, vcc\n\t"
      "v_cndmask_b32 %1, %1, %9, vcc\n\t"
      "v_cndmask_b32 %2, %2, %10, vcc\n\t"
      "v_cndmask_b32 %3, %3, %11, vcc\n\t"
      "v_cndmask_b32 %4, %4, %12, vcc\n\t"
      "v_cndmask_b32 %5, %5, %13, vcc\n\t"
      "v_cndmask_b32 %6, %6, %14, vcc\n\t"
      "v_cndmask_b32 %7, %7, %15, vcc"
      : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]),
        "=&v"(t.l[0]), "=&v"(t.l[1]), "=&v"(t.l[2]), "=&v"(t.l[3]), "=&v"(t.l[4]), "=&v"(t.l[5]), "=&v"(t.l[6]), "=&v"(t.l[7])
      : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
        "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]),
        "v"(M::mod(0)), "v"(M::mod(1)), "v"(M::mod(2)), "v"(M::mod(3)), "v"(M::mod(4)), "v"(M::mod(5)), "v"(M::mod(6)), "v"(M::mod(7))
      : "vcc");
}
// r = a - b mod p for a, b < p                          (25 instructions; the modulus as literals)
template <class P, class M = P>
__device__ __forceinline__ void sub_sel(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
  Fp<P> d;
  uint32_t mask;
  asm("v_sub_co_u32 %8, vcc, %17, %25\n\t"
      "v_subb_co_u32 %9, vcc, %18, %26, vcc\n\t"
      "v_subb_co_u32 %10, vcc, %19, %27, vcc\n\t"
      "v_subb_co_u32 %11, vcc, %20, %28, vcc\n\t"
      "v_subb_co_u32 %12, vcc, %21, %29, vcc\n\t"
      "v_subb_co_u32 %13, vcc, %22, %30, vcc\n\t"
      "v_subb_co_u32 %14, vcc, %23, %31, vcc\n\t"
      "v_subb_co_u32 %15, vcc, %24, %32, vcc\n\t"
      "v_cndmask_b32_e64 %16, 0, -1, vcc\n\t"
      "v_and_b32 %0, %33, %16\n\t"
      "v_and_b32 %1, %34, %16\n\t"
      "v_and_b32 %2, %35, %16\n\t"
      "v_and_b32 %3, %36, %16\n\t"
      "v_and_b32 %4, %37, %16\n\t"
      "v_and_b32 %5, %38, %16\n\t"
      "v_and_b32 %6, %39, %16\n\t"
      "v_and_b32 %7, %40, %16\n\t"
      "v_add_co_u32 %0, vcc, %8, %0\n\t"
      "v_addc_co_u32 %1, vcc, %9, %1, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %2, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %3, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %4, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %5, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %6, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %7, vcc"
      : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]),
        "=&v"(d.l[0]), "=&v"(d.l[1]), "=&v"(d.l[2]), "=&v"(d.l[3]), "=&v"(d.l[4]), "=&v"(d.l[5]), "=&v"(d.l[6]), "=&v"(d.l[7]), "=&v"(mask)
      : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
        "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]),
        "n"(M::mod(0)), "n"(M::mod(1)), "n"(M::mod(2)), "n"(M::mod(3)), "n"(M::mod(4)), "n"(M::mod(5)), "n"(M::mod(6)), "n"(M::mod(7))
      : "vcc");
}
#endif

template <class P>
LH_HD Fp<P> reduce_once(const Fp<P>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  Fp<P> r;
  reduce_sel(r, a);
  return r;
#else
  return reduce_once_generic(a);
#endif
}

template <class P>
LH_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  Fp<P> r;
  add_sel(r, a, b);  // a, b < p < 2^254: no carry out of 256 bits
  return r;
#else
  return add_generic(a, b);
#endif
}

template <class P>
LH_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  Fp<P> r;
  sub_sel(r, a, b);  // a - b, plus p when that borrowed
  return r;
#else
  return sub_generic(a, b);
#endif
}

template <class P>
LH_HD Fp<P> neg(const Fp<P>& a) {
  return sub(Fp<P>::zero(), a);
}

template <class P>
LH_HD Fp<P> dbl(const Fp<P>& a) {
  return add(a, a);
}

// Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs: the host form, and what the device form is checked against
// (tools/ubench/mul_forms.hip).
template <class P>
LH_HD Fp<P> mul_cios(const Fp<P>& a, const Fp<P>& b) {
  uint32_t t[8];
#pragma unroll
  for (int j = 0; j < 8; j++) t[j] = 0;
  uint32_t t8 = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.l[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t s = (uint64_t)a.l[j] * bi + t[j] + c;
      t[j] = (uint32_t)s;
      c = s >> 32;
    }
    uint64_t s8 = (uint64_t)t8 + c;
    t8 = (uint32_t)s8;
    uint32_t t9 = (uint32_t)(s8 >> 32);
    const uint32_t m = t[0] * P::INV;
    uint64_t s = (uint64_t)m * P::mod(0) + t[0];
    c = s >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      s = (uint64_t)m * P::mod(j) + t[j] + c;
      t[j - 1] = (uint32_t)s;
      c = s >> 32;
    }
    s8 = (uint64_t)t8 + c;
    t[7] = (uint32_t)s8;
    t8 = t9 + (uint32_t)(s8 >> 32);
  }
  Fp<P> r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = t[j];
  return reduce_once(r);  // result < 2p < 2^255, t8 == 0
}

#if defined(__HIP_DEVICE_COMPILE__)
// Device form: product scanning (column by column, a 96-bit accumulator, the Montgomery quotient digit of a column as
// soon as its low word is known).  The accumulation step is spelled out as the two instructions it should be -
// v_mad_u64_u32 with carry-out and an add-with-carry on the third word: from the C++ of mul_cios the compiler makes
// 128 multiply-adds, 133 64-bit additions and 290 register moves per product (zero-extended operand pairs), this is
// 129 + 128 + ~50.  Measured (tools/ubench/mul_forms.hip, MI355X): 87.7 -> 119.2 G products/s, identical results.
#define LH_MAC(x, y)                                                                          \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"      \
               : "+v"(acc), "+v"(top)                                                         \
               : "v"(x), "v"(y)                                                               \
               : "vcc")
#define LH_MACS(x, sc)                                                                        \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"      \
               : "+v"(acc), "+v"(top)                                                         \
               : "v"(x), "s"(sc)                                                              \
               : "vcc")
template <class P>
__device__ __forceinline__ Fp<P> mul_scan(const Fp<P>& a, const Fp<P>& b) {
  uint64_t acc = 0;
  uint32_t top = 0;
  uint32_t m[8], r[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) LH_MAC(a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) LH_MACS(m[i], P::mod(k - i));
    m[k] = (uint32_t)acc * P::INV;
    LH_MACS(m[k], P::mod(0));
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) LH_MAC(a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) LH_MACS(m[i], P::mod(k - i));
    r[k - 8] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
  Fp<P> out;
#pragma unroll
  for (int j = 0; j < 8; j++) out.l[j] = r[j];
  return reduce_once(out);  // a, b < p < 2^254: the result is < 2p < 2^255, no ninth word
}
// sum_j a[j] * b[j] * R^-1 mod p with ONE Montgomery reduction: the K operand products of a column go into the same
// accumulator, the reduction digits are those of the sum (K * 64 + 65 multiply-adds instead of K * 129: 0.62x for K = 4,
// 0.56x for K = 8).  The sum is < K p^2, so the unreduced result is < (K p / R + 1) p = (0.19 K + 1) p < 2^256 for
// K <= 16 (BN254: p / R = 0.189 for both fields): it takes ceil(0.19 K) conditional subtractions.  The layer
// expressions of the grand products (sum_k l_k r_k with the batching coefficients folded into l, kernels_gkr.hip and
// the streaming rounds) and the curve formulas' two-term sums are the users.
template <class P, int K>
__device__ __forceinline__ Fp<P> dot_scan(const Fp<P>* a, const Fp<P>* b) {
  static_assert(K >= 1 && K <= 16, "dot_scan: the result must stay below 2^256");
  uint64_t acc = 0;
  uint32_t top = 0;
  uint32_t m[8], r[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int j = 0; j < K; j++) {
#pragma unroll
      for (int i = 0; i <= k; i++) LH_MAC(a[j].l[i], b[j].l[k - i]);
    }
#pragma unroll
    for (int i = 0; i < k; i++) LH_MACS(m[i], P::mod(k - i));
    m[k] = (uint32_t)acc * P::INV;
    LH_MACS(m[k], P::mod(0));
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int j = 0; j < K; j++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) LH_MAC(a[j].l[i], b[j].l[k - i]);
    }
#pragma unroll
    for (int i = k - 7; i < 8; i++) LH_MACS(m[i], P::mod(k - i));
    r[k - 8] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
  Fp<P> out;
#pragma unroll
  for (int j = 0; j < 8; j++) out.l[j] = r[j];
  constexpr int NRED = K <= 5 ? 1 : K <= 10 ? 2 : K <= 15 ? 3 : 4;  // (0.189 K + 1) p < (NRED + 1) p
#pragma unroll
  for (int t = 0; t < NRED; t++) out = reduce_once(out);
  return out;
}
// The same two routines with ONE asm block per column (ff_cols.inc, generated by tools/gen_ff_cols.py): after every asm
// block whose result the next instruction reads the compiler's hazard recogniser puts an `s_nop 0` (it must assume a
// dst_sel write), which with a block per multiply-add was 136 + ~16 extra issue slots next to the 307 instructions of a
// product.  Same instructions otherwise, same results (tools/ubench/mul_cols.hip).
#ifndef LH_FF_COLS
#define LH_FF_COLS 1
#endif
#include "ff_cols.inc"
// (the next column's first block writes `top` afresh - the carry of its first multiply-add - so it is not zeroed here)
#define LH_COL_STEP_LO(k)                       \
  m[k] = (uint32_t)acc * P::INV;                \
  LH_MACS(m[k], P::mod(0));                     \
  acc = (acc >> 32) | ((uint64_t)top << 32)
#define LH_COL_STEP_HI(k)                       \
  r[k - 8] = (uint32_t)acc;                     \
  acc = (acc >> 32) | ((uint64_t)top << 32)
template <class P>
__device__ __forceinline__ Fp<P> mul_scan_cols(const Fp<P>& a, const Fp<P>& b) {
  uint64_t acc;
  uint32_t top;
  uint32_t m[8], r[8];
  LH_COL_MUL_0(a, b, m, P); LH_COL_STEP_LO(0);
  LH_COL_MUL_1(a, b, m, P); LH_COL_STEP_LO(1);
  LH_COL_MUL_2(a, b, m, P); LH_COL_STEP_LO(2);
  LH_COL_MUL_3(a, b, m, P); LH_COL_STEP_LO(3);
  LH_COL_MUL_4(a, b, m, P); LH_COL_STEP_LO(4);
  LH_COL_MUL_5(a, b, m, P); LH_COL_STEP_LO(5);
  LH_COL_MUL_6(a, b, m, P); LH_COL_STEP_LO(6);
  LH_COL_MUL_7(a, b, m, P); LH_COL_STEP_LO(7);
  LH_COL_MUL_8(a, b, m, P); LH_COL_STEP_HI(8);
  LH_COL_MUL_9(a, b, m, P); LH_COL_STEP_HI(9);
  LH_COL_MUL_10(a, b, m, P); LH_COL_STEP_HI(10);
  LH_COL_MUL_11(a, b, m, P); LH_COL_STEP_HI(11);
  LH_COL_MUL_12(a, b, m, P); LH_COL_STEP_HI(12);
  LH_COL_MUL_13(a, b, m, P); LH_COL_STEP_HI(13);
  LH_COL_MUL_14(a, b, m, P); LH_COL_STEP_HI(14);
  r[7] = (uint32_t)acc;  // (column 15 has no products: the last carry word)
  Fp<P> out;
#pragma unroll
  for (int j = 0; j < 8; j++) out.l[j] = r[j];
  return reduce_once(out);
}
#define LH_COL_DOT(k)                                       \
  LH_COL_VV0_##k(a[0], b[0]);                               \
  _Pragma("unroll") for (int j = 1; j < K; j++) {           \
    LH_COL_VV_##k(a[j], b[j]);                              \
  }                                                         \
  LH_COL_VS_##k(m, P)
template <class P, int K>
__device__ __forceinline__ Fp<P> dot_scan_cols(const Fp<P>* a, const Fp<P>* b) {
  static_assert(K >= 1 && K <= 16, "dot_scan: the result must stay below 2^256");
  uint64_t acc;
  uint32_t top;
  uint32_t m[8], r[8];
  LH_COL_DOT(0); LH_COL_STEP_LO(0);
  LH_COL_DOT(1); LH_COL_STEP_LO(1);
  LH_COL_DOT(2); LH_COL_STEP_LO(2);
  LH_COL_DOT(3); LH_COL_STEP_LO(3);
  LH_COL_DOT(4); LH_COL_STEP_LO(4);
  LH_COL_DOT(5); LH_COL_STEP_LO(5);
  LH_COL_DOT(6); LH_COL_STEP_LO(6);
  LH_COL_DOT(7); LH_COL_STEP_LO(7);
  LH_COL_DOT(8); LH_COL_STEP_HI(8);
  LH_COL_DOT(9); LH_COL_STEP_HI(9);
  LH_COL_DOT(10); LH_COL_STEP_HI(10);
  LH_COL_DOT(11); LH_COL_STEP_HI(11);
  LH_COL_DOT(12); LH_COL_STEP_HI(12);
  LH_COL_DOT(13); LH_COL_STEP_HI(13);
  LH_COL_DOT(14); LH_COL_STEP_HI(14);
  r[7] = (uint32_t)acc;
  Fp<P> out;
#pragma unroll
  for (int j = 0; j < 8; j++) out.l[j] = r[j];
  constexpr int NRED = K <= 5 ? 1 : K <= 10 ? 2 : K <= 15 ? 3 : 4;
#pragma unroll
  for (int t = 0; t < NRED; t++) out = reduce_once(out);
  return out;
}
#undef LH_COL_DOT
#undef LH_COL_STEP_LO
#undef LH_COL_STEP_HI
#endif

template <class P>
LH_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return LH_FF_COLS ? mul_scan_cols(a, b) : mul_scan(a, b);
#else
  return mul_cios(a, b);
#endif
}

// sum_j a[j] * b[j] (device: one reduction for the K products, dot_scan above)
template <class P, int K>
LH_HD Fp<P> dot(const Fp<P>* a, const Fp<P>* b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return LH_FF_COLS ? dot_scan_cols<P, K>(a, b) : dot_scan<P, K>(a, b);
#else
  Fp<P> s = mul_cios(a[0], b[0]);
  for (int j = 1; j < K; j++) s = add_generic(s, mul_cios(a[j], b[j]));
  return s;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// ------------------------------------------------------------------ lazy forms: values in [0, 2 p) (device only)
// With R = 2^256 > 4 p (both BN254 moduli are below 2^254) the Montgomery product of two values below 2 p is
// (a b + m p) / R < p (4 p / R + 1) < 2 p WITHOUT the final conditional subtraction (16 of a product's ~296 instructions);
// additions and subtractions stay in the range with a conditional subtraction / addition of 2 p - what they cost against
// p.  A dot product of two pairs lands below p (8 p / R + 1) < 2.51 p: one conditional subtraction of p brings it below
// 1.51 p.  Canonical values (< p) are lazy values; `canon` makes a lazy value canonical again.  Used by the MSM's bucket
// accumulation (ec.cuh add_mixed_lazy): 8 of the 10 products of a mixed addition lose their subtraction.
template <class P>
__device__ __forceinline__ Fp<P> mul_lazy(const Fp<P>& a, const Fp<P>& b) {
  uint64_t acc;
  uint32_t top;
  uint32_t m[8], r[8];
#define LH_COL_STEP_LO(k)                       \
  m[k] = (uint32_t)acc * P::INV;                \
  LH_MACS(m[k], P::mod(0));                     \
  acc = (acc >> 32) | ((uint64_t)top << 32)
#define LH_COL_STEP_HI(k)                       \
  r[k - 8] = (uint32_t)acc;                     \
  acc = (acc >> 32) | ((uint64_t)top << 32)
  LH_COL_MUL_0(a, b, m, P); LH_COL_STEP_LO(0);
  LH_COL_MUL_1(a, b, m, P); LH_COL_STEP_LO(1);
  LH_COL_MUL_2(a, b, m, P); LH_COL_STEP_LO(2);
  LH_COL_MUL_3(a, b, m, P); LH_COL_STEP_LO(3);
  LH_COL_MUL_4(a, b, m, P); LH_COL_STEP_LO(4);
  LH_COL_MUL_5(a, b, m, P); LH_COL_STEP_LO(5);
  LH_COL_MUL_6(a, b, m, P); LH_COL_STEP_LO(6);
  LH_COL_MUL_7(a, b, m, P); LH_COL_STEP_LO(7);
  LH_COL_MUL_8(a, b, m, P); LH_COL_STEP_HI(8);
  LH_COL_MUL_9(a, b, m, P); LH_COL_STEP_HI(9);
  LH_COL_MUL_10(a, b, m, P); LH_COL_STEP_HI(10);
  LH_COL_MUL_11(a, b, m, P); LH_COL_STEP_HI(11);
  LH_COL_MUL_12(a, b, m, P); LH_COL_STEP_HI(12);
  LH_COL_MUL_13(a, b, m, P); LH_COL_STEP_HI(13);
  LH_COL_MUL_14(a, b, m, P); LH_COL_STEP_HI(14);
#undef LH_COL_STEP_LO
#undef LH_COL_STEP_HI
  r[7] = (uint32_t)acc;
  Fp<P> out;
#pragma unroll
  for (int j = 0; j < 8; j++) out.l[j] = r[j];
  return out;
}
template <class P>
__device__ __forceinline__ Fp<P> add_lazy(const Fp<P>& a, const Fp<P>& b) {  // a + b < 4 p < 2^256
  Fp<P> r;
  add_sel<P, Twice<P>>(r, a, b);
  return r;
}
template <class P>
__device__ __forceinline__ Fp<P> sub_lazy(const Fp<P>& a, const Fp<P>& b) {  // a - b, plus 2 p when that borrowed
  Fp<P> r;
  sub_sel<P, Twice<P>>(r, a, b);
  return r;
}
template <class P>
__device__ __forceinline__ Fp<P> canon(const Fp<P>& a) {  // [0, 2 p) -> [0, p)
  return reduce_once(a);
}
template <class P>
__device__ __forceinline__ bool is_zero_lazy(const Fp<P>& a) {  // a = 0 mod p for a in [0, 2 p): 0 or p
  bool z = true, m = true;
#pragma unroll
  for (int i = 0; i < 8; i++) z = z && a.l[i] == 0, m = m && a.l[i] == P::mod(i);
  return z || m;
}
#undef LH_MAC
#undef LH_MACS
#elif defined(__HIPCC__)
// (the host pass over device code only needs the names: canonical arithmetic is a valid lazy arithmetic)
template <class P>
LH_HD Fp<P> mul_lazy(const Fp<P>& a, const Fp<P>& b) { return mul(a, b); }
template <class P>
LH_HD Fp<P> add_lazy(const Fp<P>& a, const Fp<P>& b) { return add(a, b); }
template <class P>
LH_HD Fp<P> sub_lazy(const Fp<P>& a, const Fp<P>& b) { return sub(a, b); }
template <class P>
LH_HD Fp<P> canon(const Fp<P>& a) { return a; }
template <class P>
LH_HD bool is_zero_lazy(const Fp<P>& a) { return a.is_zero(); }
#endif

template <class P>
LH_HD Fp<P> sqr(const Fp<P>& a) {
  return mul(a, a);
}

// Montgomery form <-> canonical integer
template <class P>
LH_HD Fp<P> to_mont(const Fp<P>& canon) {
  return mul(canon, Fp<P>::r2());
}
template <class P>
LH_HD Fp<P> from_mont(const Fp<P>& a) {
  Fp<P> one;
#pragma unroll
  for (int i = 0; i < 8; i++) one.l[i] = (i == 0);
  return mul(a, one);
}
template <class P>
LH_HD Fp<P> from_u64(uint64_t v) {
  Fp<P> c = Fp<P>::zero();
  c.l[0] = (uint32_t)v;
  c.l[1] = (uint32_t)(v >> 32);
  return to_mont(c);
}

// a^e for a 254-bit exponent given as canonical limbs (square-and-multiply, MSB first)
template <class P>
LH_HD Fp<P> pow_limbs(const Fp<P>& a, const uint32_t* e) {
  Fp<P> acc = Fp<P>::one();
  for (int i = 7; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      acc = sqr(acc);
      if ((e[i] >> b) & 1u) acc = mul(acc, a);
    }
  }
  return acc;
}

// Fermat inverse a^(p-2); zero maps to zero.
template <class P>
LH_HD Fp<P> inv(const Fp<P>& a) {
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = P::mod(i);
  e[0] -= 2u;  // low limb of both moduli is > 2
  return pow_limbs(a, e);
}

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

}  // namespace lh
