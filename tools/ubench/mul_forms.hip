// Development micro-benchmark: the library's CIOS Montgomery multiplication (ff.cuh mul, compiled from C++:
// 128 v_mad_u64_u32 + 133 64-bit adds + 290 moves per product) against a product-scanning form whose accumulation
// step is written as two instructions (v_mad_u64_u32 with carry-out, v_addc on the third accumulator word).
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/mul_forms.hip -o /tmp/mul_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "ff.cuh"
using namespace lh;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// acc3 (acc: 64 bits, top: 32 bits) += x * y
#define MAC(x, y)                                                                                      \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"              \
               : "+v"(acc), "+v"(top)                                                                  \
               : "v"(x), "v"(y)                                                                        \
               : "vcc")
#define MACS(x, s)                                                                                     \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"              \
               : "+v"(acc), "+v"(top)                                                                  \
               : "v"(x), "s"(s)                                                                        \
               : "vcc")

template <class P>
__device__ __forceinline__ Fp<P> mul_ps(const Fp<P>& a, const Fp<P>& b) {
  uint64_t acc = 0;
  uint32_t top = 0;
  uint32_t m[8], r[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) MAC(a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) MACS(m[i], P::mod(k - i));
    m[k] = (uint32_t)acc * P::INV;
    MACS(m[k], P::mod(0));
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) MAC(a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) MACS(m[i], P::mod(k - i));
    r[k - 8] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
  Fp<P> out;
#pragma unroll
  for (int j = 0; j < 8; j++) out.l[j] = r[j];
  return reduce_once(out);  // a, b < p < 2^254: the result is < 2p < 2^255, no ninth word
}

// additions / subtractions: device forms (ff.cuh add, sub) against the generic C++ forms
__global__ __launch_bounds__(256) void addsub_check(const Fr* a, const Fr* b, Fr* o0, Fr* o1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const Fr x = reduce_once_generic(a[i]), y = reduce_once_generic(b[i]);
  o0[i] = add_generic(sub_generic(add_generic(x, y), sub_generic(y, x)), reduce_once_generic(add_generic(x, x)));
  o1[i] = add(sub(add(x, y), sub(y, x)), reduce_once(dbl(x)));
}
template <int FORM>
__global__ __launch_bounds__(256) void addsub_chain(const Fr* a, const Fr* b, Fr* o, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = reduce_once_generic(a[i]), y = reduce_once_generic(b[i]);
  for (int k = 0; k < iters; k++) {
    x = FORM ? add(x, y) : add_generic(x, y);
    y = FORM ? sub(y, x) : sub_generic(y, x);
  }
  o[i] = FORM ? add(x, y) : add_generic(x, y);
}

template <int FORM>
__global__ __launch_bounds__(256) void chain(const Fr* a, const Fr* b, Fr* o, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = a[i], y = b[i];
  for (int k = 0; k < iters; k++) x = FORM ? mul_ps(x, y) : mul_cios(x, y);
  o[i] = x;
}

int main() {
  const size_t n = (size_t)1 << 22;
  std::vector<Fr> ha(n), hb(n);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < 8; k++) ha[i].l[k] = rnd(), hb[i].l[k] = rnd();
    ha[i].l[7] &= 0x0fffffffu, hb[i].l[7] &= 0x0fffffffu;  // < 2^252 < r
  }
  // edge values: 0, 1, r - 1 in raw form
  memset(&ha[0], 0, sizeof(Fr));
  memset(&ha[1], 0, sizeof(Fr)), ha[1].l[0] = 1;
  for (int k = 0; k < 8; k++) ha[2].l[k] = FrParams::mod(k), hb[2].l[k] = FrParams::mod(k);
  ha[2].l[0] -= 1, hb[2].l[0] -= 1;
  Fr *da, *db, *d0, *d1;
  CK(hipMalloc(&da, n * sizeof(Fr)));
  CK(hipMalloc(&db, n * sizeof(Fr)));
  CK(hipMalloc(&d0, n * sizeof(Fr)));
  CK(hipMalloc(&d1, n * sizeof(Fr)));
  CK(hipMemcpy(da, ha.data(), n * sizeof(Fr), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), n * sizeof(Fr), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 64;
  float ms[2] = {0, 0};
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain<0>, n / 256, 256, 0, 0, da, db, d0, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[0], e0, e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain<1>, n / 256, 256, 0, 0, da, db, d1, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[1], e0, e1));
  }
  std::vector<Fr> h0(n), h1(n);
  CK(hipMemcpy(h0.data(), d0, n * sizeof(Fr), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), d1, n * sizeof(Fr), hipMemcpyDeviceToHost));
  const bool same = memcmp(h0.data(), h1.data(), n * sizeof(Fr)) == 0;
  printf("CIOS (C++): %.1f G mul/s   product scanning (mad + addc): %.1f G mul/s   results %s\n",
         n * (double)iters / ms[0] / 1e6, n * (double)iters / ms[1] / 1e6, same ? "identical" : "DIFFER");
  if (!same) return 2;
  hipLaunchKernelGGL(addsub_check, n / 256, 256, 0, 0, da, db, d0, d1);
  CK(hipMemcpy(h0.data(), d0, n * sizeof(Fr), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), d1, n * sizeof(Fr), hipMemcpyDeviceToHost));
  const bool same_as = memcmp(h0.data(), h1.data(), n * sizeof(Fr)) == 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(addsub_chain<0>, n / 256, 256, 0, 0, da, db, d0, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[0], e0, e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(addsub_chain<1>, n / 256, 256, 0, 0, da, db, d1, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[1], e0, e1));
  }
  CK(hipMemcpy(h0.data(), d0, n * sizeof(Fr), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), d1, n * sizeof(Fr), hipMemcpyDeviceToHost));
  const bool same_chain = memcmp(h0.data(), h1.data(), n * sizeof(Fr)) == 0;
  printf("modular add + sub pairs: generic C++ %.1f G/s   carry chains %.1f G/s   results %s\n",
         n * (double)iters / ms[0] / 1e6, n * (double)iters / ms[1] / 1e6, same_as && same_chain ? "identical" : "DIFFER");
  return same_as && same_chain ? 0 : 3;
}
