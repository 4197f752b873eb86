#include <fenv.h>
#include <stdio.h>
#include <stdlib.h>
#include "mul_fp64_core.h"
// usage: test <modulus hex limbs 5> <np> then lines of a b (5 limbs each, decimal) on stdin -> result limbs
int main() {
  fesetround(FE_TOWARDZERO);
  F52Mod m;
  unsigned long long v;
  for (int k = 0; k < 5; k++) { if (scanf("%llu", &v) != 1) return 1; m.n[k] = (double)v; m.ni[k] = v; }
  if (scanf("%llu", &v) != 1) return 1;
  m.np = (double)v;
  int n;
  if (scanf("%d", &n) != 1) return 1;
  for (int t = 0; t < n; t++) {
    F52 a, b;
    for (int k = 0; k < 5; k++) { scanf("%llu", &v); a.l[k] = (double)v; }
    for (int k = 0; k < 5; k++) { scanf("%llu", &v); b.l[k] = (double)v; }
    F52 r = f52_mont_mul(a, b, m);
    for (int k = 0; k < 5; k++) printf("%llu ", (unsigned long long)r.l[k]);
    printf("\n");
  }
  return 0;
}
