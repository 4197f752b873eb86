// 5 x 52-bit limbs held as doubles; Montgomery product a b 2^-260 mod n with the 52 x 52 -> 104-bit limb products made by
// two FMAs each (round toward zero): hi = fma(x, y, 2^104) has floor(x y / 2^52) in its mantissa, lo = fma(x, y, (2^104 +
// 2^52) - hi) = 2^52 + (x y mod 2^52); the raw bit patterns of both are added into 64-bit integer columns (the exponent
// fields add up to known constants that are taken off when a column is read).
#include <stdint.h>
#include <string.h>
#include <math.h>
#ifndef FP64_HD
#define FP64_HD static inline
#endif
struct F52 { double l[5]; };
struct F52Mod { double n[5]; double np; uint64_t ni[5]; };  // modulus limbs (as doubles and as integers), -n^-1 mod 2^52
// (no double -> integer conversion below: the compiler brackets its expansion with writes of the DEFAULT rounding mode to
// the MODE register, which would switch round-toward-zero off for everything after it)
FP64_HD uint64_t f52_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
FP64_HD double f52_dbl(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
#define F52_KH 0x4670000000000000ull  /* exponent field of 2^104 */
#define F52_KL 0x4330000000000000ull  /* exponent field of 2^52 */
#define F52_M  0x000fffffffffffffull
#ifndef F52_FMA
#define F52_FMA(a, b, c) fma(a, b, c)
#endif
// integer in [0, 2^52) -> double
FP64_HD double f52_from_int(uint64_t v) { return f52_dbl(v | F52_KL) - 0x1p52; }
FP64_HD void f52_prod(double x, double y, uint64_t* hi, uint64_t* lo) {
  const double ph = F52_FMA(x, y, 0x1p104);
  const double sub = (0x1p104 + 0x1p52) - ph;
  const double pl = F52_FMA(x, y, sub);
  *hi = f52_bits(ph), *lo = f52_bits(pl);
}
FP64_HD F52 f52_mont_mul(const F52& a, const F52& b, const F52Mod& m) {
  uint64_t c[11];
  for (int k = 0; k < 11; k++) c[k] = 0;
  for (int i = 0; i < 5; i++) {
    for (int j = 0; j < 5; j++) {
      uint64_t hi, lo;
      f52_prod(a.l[i], b.l[j], &hi, &lo);
      c[i + j] += lo, c[i + j + 1] += hi;
    }
    // column i now holds (2 i + 1) lo patterns and 2 i hi patterns (+ the carry of column i - 1)
    const uint64_t ci = (uint64_t)(2 * i + 1) * F52_KL + (uint64_t)(2 * i) * F52_KH;
    const uint64_t t = c[i] - ci;
    uint64_t qh, ql;
    f52_prod(f52_from_int(t & F52_M), m.np, &qh, &ql);
    const double q = f52_from_int(ql & F52_M);
    for (int j = 0; j < 5; j++) {
      uint64_t hi, lo;
      f52_prod(q, m.n[j], &hi, &lo);
      c[i + j] += lo, c[i + j + 1] += hi;
    }
    c[i + 1] += (c[i] - ci - F52_KL) >> 52;  // the column is 0 mod 2^52 now
  }
  // columns 5..9: 2 (9 - m) lo patterns, 2 (10 - m) hi patterns; carries, then one conditional subtraction
  uint64_t r[5], carry = 0;
  for (int k = 0; k < 5; k++) {
    const int mcol = 5 + k;
    const uint64_t v = c[mcol] - (uint64_t)(2 * (9 - mcol)) * F52_KL - (uint64_t)(2 * (10 - mcol)) * F52_KH + carry;
    r[k] = v & F52_M;
    carry = v >> 52;
  }
  uint64_t d[5], borrow = 0;
  for (int k = 0; k < 5; k++) {
    const uint64_t v = r[k] - m.ni[k] - borrow;
    d[k] = v & F52_M;
    borrow = (v >> 63) & 1;
  }
  F52 out;
  for (int k = 0; k < 5; k++) out.l[k] = f52_from_int(borrow ? r[k] : d[k]);
  return out;
}
