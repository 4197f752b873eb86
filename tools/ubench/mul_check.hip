// Development check: the device Montgomery multiplication (ff.cuh mul: product scanning, inline asm) against the C++ CIOS
// form on 4 M random and edge inputs, for both fields.  build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/mul_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "ff.cuh"
using namespace lh;
template <class F> __global__ void both(const F* a, const F* b, F* o0, F* o1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const F x = reduce_once_generic(a[i]), y = reduce_once_generic(b[i]);
  o0[i] = mul_cios(x, y);
  o1[i] = mul(x, y);
}
template <class F> int run(const char* name) {
  const size_t n = (size_t)1 << 22;
  std::vector<F> ha(n), hb(n), h0(n), h1(n);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (size_t i = 0; i < n; i++) { for (int k = 0; k < 8; k++) ha[i].l[k] = rnd(), hb[i].l[k] = rnd(); ha[i].l[7] &= 0x3fffffffu; hb[i].l[7] &= 0x3fffffffu; }
  for (int k = 0; k < 8; k++) ha[0].l[k] = 0, ha[1].l[k] = k == 0, ha[2].l[k] = 0xffffffffu, hb[2].l[k] = 0xffffffffu;
  ha[2].l[7] = 0x30000000u; hb[2].l[7] = 0x30000000u;
  F *da, *db, *d0, *d1;
  hipMalloc(&da, n * sizeof(F)); hipMalloc(&db, n * sizeof(F)); hipMalloc(&d0, n * sizeof(F)); hipMalloc(&d1, n * sizeof(F));
  hipMemcpy(da, ha.data(), n * sizeof(F), hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * sizeof(F), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(both<F>, n / 256, 256, 0, 0, da, db, d0, d1);
  hipMemcpy(h0.data(), d0, n * sizeof(F), hipMemcpyDeviceToHost); hipMemcpy(h1.data(), d1, n * sizeof(F), hipMemcpyDeviceToHost);
  bool same = memcmp(h0.data(), h1.data(), n * sizeof(F)) == 0;
  printf("%s: device product-scanning mul vs CIOS on %zu inputs: %s\n", name, n, same ? "identical" : "DIFFER");
  return same ? 0 : 1;
}
int main() { return run<Fr>("Fr") | run<Fq>("Fq"); }
