// 9 x 29-bit limbs ("lazy-carry" form) of the BN254 fields for the multiplier-bound inner loop of the MSM.
//
// Why: every VALU instruction of a wave64 costs ~4.4-4.9 SIMD cycles here, v_mad_u64_u32 and v_addc_co_u32 alike
// (tools/ubench/mul_fp64.hip, profiles/r04_ubench_mul_fp64.txt), so a Montgomery product is priced by its instruction
// COUNT.  With 8 x 32-bit limbs a column of 32 x 32-bit products overflows 64 bits: every multiply-add drags an
// add-with-carry into a third accumulator word (ff.cuh mul_scan: 129 + 128 + ~50 instructions).  With 29-bit limbs the 18
// products of a column (< 2^58 each) fit ONE 64-bit accumulator: 81 + 81 multiply-adds, 9 quotient digits (multiply-low +
// mask), 17 shifts, 9 masks - ~205 instructions, and no carry chain.  Values are kept in [0, 16 p) with normalised limbs
// (the top limb has 7 spare bits), so additions and subtractions are limb-wise plus ONE carry sweep and never compare
// against the modulus; the Montgomery radix is 2^261: mul29(a, b) = a b 2^-261 mod p, < 2.5 p for a, b < 16 p.
#pragma once
#include "ff.cuh"

namespace lh {

constexpr uint32_t M29 = (1u << 29) - 1u;

template <class P>
struct Fp29 {
  uint32_t l[9];
};

// limb i (29 bits) of the modulus, and of k * modulus (k <= 16), from the 32-bit limbs
template <class P>
LH_HD constexpr uint32_t mod29(int i) {
  const int bit = 29 * i, w = bit >> 5, s = bit & 31;
  uint64_t v = w < 8 ? (uint64_t)P::mod(w) >> s : 0;
  if (s && w + 1 < 8) v |= (uint64_t)P::mod(w + 1) << (32 - s);
  return (uint32_t)v & M29;
}
// limb i of K p with normalised limbs (the top limb takes what is left)
template <class P, int K>
LH_HD constexpr uint32_t kmod29(int i) {
  uint64_t carry = 0, limb = 0;
  for (int j = 0; j <= i; j++) {
    const uint64_t v = (uint64_t)K * mod29<P>(j) + carry;
    limb = j < 8 ? (v & M29) : v;
    carry = v >> 29;
  }
  return (uint32_t)limb;
}
template <class P>
LH_HD constexpr uint32_t ninv29() {  // -p^-1 mod 2^29 (Newton on the low limb)
  const uint32_t n0 = P::mod(0);
  uint32_t inv = 1;
  for (int it = 0; it < 6; it++) inv *= 2u - n0 * inv;
  return (0u - inv) & M29;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Multiply-adds are emitted THREE (or two) to an assembly statement: between two single-instruction statements that
// write vcc the compiler's hazard recognizer puts an s_nop, which costs an issue slot per product (measured: x1.21
// instead of the x1.4 the instruction count promises); the hardware needs none (bit check below, tools/ubench/mul29.hip).
#define LH_MAD29(x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")
#define LH_MAD29S(x, sc) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(sc) : "vcc")
#define LH_MAD29_2(x0, y0, x1, y1) \
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc")
#define LH_MAD29_3(x0, y0, x1, y1, x2, y2)                                                                                  \
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0" \
      : "+v"(acc)                                                                                                          \
      : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2)                                                               \
      : "vcc")
// sum_{i = lo}^{hi} x[i] * y[k - i] into acc (y in registers: no more than one scalar operand per instruction anyway)
template <int LO, int HI, int K>
__device__ __forceinline__ void mad29_run(uint64_t& acc, const uint32_t* x, const uint32_t* y) {
  if constexpr (HI - LO + 1 >= 3) {
    LH_MAD29_3(x[LO], y[K - LO], x[LO + 1], y[K - LO - 1], x[LO + 2], y[K - LO - 2]);
    mad29_run<LO + 3, HI, K>(acc, x, y);
  } else if constexpr (HI - LO + 1 == 2) {
    LH_MAD29_2(x[LO], y[K - LO], x[LO + 1], y[K - LO - 1]);
  } else if constexpr (HI - LO + 1 == 1) {
    LH_MAD29(x[LO], y[K - LO]);
  }
}
// a b 2^-261 mod p for normalised limbs and a, b < 16 p; the result has normalised limbs and is < a b / 2^261 + p
#define LH_MAD29S_2(x0, s0, x1, s1) \
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0" : "+v"(acc) : "v"(x0), "s"(s0), "v"(x1), "s"(s1) : "vcc")
#define LH_MAD29S_3(x0, s0, x1, s1, x2, s2)                                                                                 \
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0" \
      : "+v"(acc)                                                                                                          \
      : "v"(x0), "s"(s0), "v"(x1), "s"(s1), "v"(x2), "s"(s2)                                                               \
      : "vcc")
// sum_{i = lo}^{hi} m[i] * n_{k - i}: the modulus limbs are compile-time constants in scalar registers (nine vector
// registers fewer than a copy of the modulus per lane)
template <class P, int LO, int HI, int K>
__device__ __forceinline__ void mad29_run_mod(uint64_t& acc, const uint32_t* m) {
  if constexpr (HI - LO + 1 >= 3) {
    LH_MAD29S_3(m[LO], mod29<P>(K - LO), m[LO + 1], mod29<P>(K - LO - 1), m[LO + 2], mod29<P>(K - LO - 2));
    mad29_run_mod<P, LO + 3, HI, K>(acc, m);
  } else if constexpr (HI - LO + 1 == 2) {
    LH_MAD29S_2(m[LO], mod29<P>(K - LO), m[LO + 1], mod29<P>(K - LO - 1));
  } else if constexpr (HI - LO + 1 == 1) {
    LH_MAD29S(m[LO], mod29<P>(K - LO));
  }
}
template <class P, int K>
__device__ __forceinline__ void mul29_column(uint64_t& acc, const uint32_t* a, const uint32_t* b, uint32_t* m, uint32_t* r) {
  constexpr int LO = K < 9 ? 0 : K - 8, HI = K < 9 ? K : 8;
  mad29_run<LO, HI, K>(acc, a, b);
  if constexpr (K < 9) {
    if constexpr (K >= 1) mad29_run_mod<P, 0, K - 1, K>(acc, m);
    m[K] = ((uint32_t)acc * ninv29<P>()) & M29;
    LH_MAD29S(m[K], mod29<P>(0));
  } else {
    mad29_run_mod<P, LO, 8, K>(acc, m);
    r[K - 9] = (uint32_t)acc & M29;
  }
  acc >>= 29;
  if constexpr (K < 16) mul29_column<P, K + 1>(acc, a, b, m, r);
}
template <class P>
__device__ __forceinline__ Fp29<P> mul29(const Fp29<P>& a, const Fp29<P>& b) {
  uint64_t acc = 0;
  uint32_t m[9];
  Fp29<P> r;
  mul29_column<P, 0>(acc, a.l, b.l, m, r.l);
  r.l[8] = (uint32_t)acc;
  return r;
}
#undef LH_MAD29
#undef LH_MAD29S
#undef LH_MAD29_2
#undef LH_MAD29_3
#undef LH_MAD29S_2
#undef LH_MAD29S_3
#else
// host form (checks, conversions at setup): the same columns in plain C++
template <class P>
LH_HD Fp29<P> mul29(const Fp29<P>& a, const Fp29<P>& b) {
  uint64_t acc = 0;
  uint32_t m[9];
  Fp29<P> r;
  for (int k = 0; k < 17; k++) {
    for (int i = (k < 9 ? 0 : k - 8); i <= (k < 9 ? k : 8); i++) acc += (uint64_t)a.l[i] * b.l[k - i];
    for (int i = (k < 9 ? 0 : k - 8); i <= (k < 9 ? k - 1 : 8); i++) acc += (uint64_t)m[i] * mod29<P>(k - i);
    if (k < 9) {
      m[k] = ((uint32_t)acc * ninv29<P>()) & M29;
      acc += (uint64_t)m[k] * mod29<P>(0);
    } else {
      r.l[k - 9] = (uint32_t)acc & M29;
    }
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
#endif

// one carry sweep: limbs may hold up to 32 bits (as signed 32-bit values when `t` comes from a subtraction that stays
// non-negative as a whole) -> normalised limbs
template <class P>
LH_HD Fp29<P> carry29(const int32_t* t) {
  Fp29<P> r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int32_t v = t[i] + c;
    r.l[i] = (uint32_t)v & M29;
    c = v >> 29;  // arithmetic: a negative limb borrows from the next
  }
  r.l[8] = (uint32_t)(t[8] + c);
  return r;
}
// a + b (both < 16 p in total: the caller's bound)
template <class P>
LH_HD Fp29<P> add29(const Fp29<P>& a, const Fp29<P>& b) {
  int32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = (int32_t)(a.l[i] + b.l[i]);
  return carry29<P>(t);
}
// a - b + K p  (K p >= b: the caller's bound on b; the result is < a + K p)
template <class P, int K>
LH_HD Fp29<P> sub29(const Fp29<P>& a, const Fp29<P>& b) {
  int32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    t[i] = (int32_t)(a.l[i] + kmod29<P, K>(i)) - (int32_t)b.l[i];  // in (-2^29, 2^30)
  }
  return carry29<P>(t);
}

// standard 8 x 32-bit integer <-> 9 x 29-bit limbs (plain re-slicing of the bits, no change of value)
template <class P>
LH_HD Fp29<P> slice29(const Fp<P>& x) {
  Fp29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, w = bit >> 5, s = bit & 31;
    uint64_t v = w < 8 ? (uint64_t)x.l[w] >> s : 0;
    if (s && w + 1 < 8) v |= (uint64_t)x.l[w + 1] << (32 - s);
    r.l[i] = (uint32_t)v & M29;
  }
  return r;
}
// (the value must be < 2^256)
template <class P>
LH_HD Fp<P> unslice29(const Fp29<P>& x) {
  Fp<P> r;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int shift = 29 * i - 32 * w;
      if (shift > -29 && shift < 32) v |= shift >= 0 ? (uint64_t)x.l[i] << shift : (uint64_t)x.l[i] >> (-shift);
    }
    r.l[w] = (uint32_t)v;
  }
  return r;
}

using Fq29 = Fp29<FqParams>;
using Fr29 = Fp29<FrParams>;

}  // namespace lh
