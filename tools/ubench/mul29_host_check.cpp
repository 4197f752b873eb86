// Host check of tools/ubench/ff29.cuh (9 x 29-bit lazy-carry limbs): the generic forms of mul29 / add29 / sub29 /
// slice29 against ff.cuh's CIOS product on random inputs (the device form - the same columns as v_mad_u64_u32 groups - is
// checked on the GPU by tools/ubench/mul29.hip).  build: g++ -O1 -std=c++17 -I halo2-lasso_amd/csrc tools/ubench/mul29_host_check.cpp
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "ff29.cuh"
using namespace lh;
int main() {
  typedef FrParams P;
  Fr x, y;
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  int bad = 0;
  for (int t = 0; t < 1000; t++) {
    for (int k = 0; k < 8; k++) x.l[k] = rnd(), y.l[k] = rnd();
    x.l[7] &= 0x3fffffffu; y.l[7] &= 0x3fffffffu;
    x = reduce_once_generic(x); y = reduce_once_generic(y);
    Fr ref = mul_cios(x, y);
    Fp29<P> r = mul29(slice29(x), slice29(y));
    Fr got = unslice29(r);
    for (int k = 0; k < 3; k++) got = reduce_once_generic(got);
    for (int k = 0; k < 5; k++) got = add_generic(got, got);
    if (!(got == ref)) { bad++; if (bad < 3) { printf("x0 %08x slice %08x %08x un %08x\n", x.l[0], slice29(x).l[0], slice29(x).l[1], unslice29(slice29(x)).l[0]); } }
    { Fp29<P> rt2 = sub29<P, 16>(add29(r, slice29(x)), slice29(x)); Fr one261 = Fr::one(); for (int k = 0; k < 5; k++) one261 = add_generic(one261, one261);
      Fr g2 = unslice29(mul29(rt2, slice29(one261))); for (int k = 0; k < 3; k++) g2 = reduce_once_generic(g2); for (int k = 0; k < 5; k++) g2 = add_generic(g2, g2); if (!(g2 == ref)) bad += 1000; }
    Fr rt = unslice29(slice29(x));
    if (!(rt == x)) { printf("slice roundtrip bad\n"); return 1; }
  }
  printf("bad %d ninv %08x mod29_0 %08x\n", bad, ninv29<P>(), mod29<P>(0));
  printf("check n*ninv: %08x\n", (mod29<P>(0) * ninv29<P>()) & M29);
  return 0;
}
