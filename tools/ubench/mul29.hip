// The 9 x 29-bit lazy-carry Montgomery product (tools/ubench/ff29.cuh) against the 8 x 32-bit product-scanning
// form: bit check through mul_cios (the radices differ: x 2^5) on random and extreme inputs, both fields, inputs up to
// 16 p; throughput of dependent product chains (two per thread) and of a chain that mixes additions and subtractions in
// the proportion of an XYZZ mixed addition.
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc -I tools/ubench tools/ubench/mul29.hip -o tools/ubench/mul29.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "ff29.cuh"
using namespace lh;

template <class F> __global__ void check_kernel(const F* a, const F* b, uint32_t* bad, size_t n, int ka, int kb) {
  typedef typename F::params P;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F x = reduce_once_generic(a[i]), y = reduce_once_generic(b[i]);
  const F ref = mul_cios(x, y);  // x y 2^-256
  // operands x + ka p, y + kb p (the lazy range) as 29-bit limbs
  Fp29<P> xa = slice29(x), yb = slice29(y);
  F pm;
  for (int w = 0; w < 8; w++) pm.l[w] = P::mod(w);
  const Fp29<P> p29 = slice29(pm);
  for (int k = 0; k < ka; k++) xa = add29(xa, p29);
  for (int k = 0; k < kb; k++) yb = add29(yb, p29);
  const Fp29<P> r = mul29(xa, yb);  // x y 2^-261 + (0..2) p
  F got = unslice29(r);
  for (int k = 0; k < 3; k++) got = reduce_once_generic(got);
  for (int k = 0; k < 5; k++) got = add_generic(got, got);
  // sub29 / add29 round trip on the way: (r + xa) - xa + 16 p  ==  r mod p
  // (the value r + 16 p does not fit 256 bits: one product with 2^261 mod p - the radix's "one" - brings it back below 2.5 p)
  Fp29<P> rt = sub29<P, 16>(add29(r, xa), xa);
  F one261 = F::one();  // 2^256 mod p
  for (int k = 0; k < 5; k++) one261 = add_generic(one261, one261);
  F got2 = unslice29(mul29(rt, slice29(one261)));
  for (int k = 0; k < 3; k++) got2 = reduce_once_generic(got2);
  for (int k = 0; k < 5; k++) got2 = add_generic(got2, got2);
  if (!(got == ref) || !(got2 == ref)) atomicAdd(bad, 1u);
}
template <class P> __global__ __launch_bounds__(256) void chain29(const Fp29<P>* in, Fp29<P>* out, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fp29<P> x = in[i], y = in[i ^ 1], u = in[i ^ 2], v = in[i ^ 3];
  for (int k = 0; k < iters; k += 2) {
    x = mul29(x, y);
    u = mul29(u, v);
  }
  out[i] = add29(x, u);
}
template <class F> __global__ __launch_bounds__(256) void chain32(const F* in, F* out, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x = in[i], y = in[i ^ 1], u = in[i ^ 2], v = in[i ^ 3];
  for (int k = 0; k < iters; k += 2) {
    x = mul(x, y);
    u = mul(u, v);
  }
  out[i] = add(x, u);
}
// the arithmetic mix of one XYZZ mixed addition: 10 products, 6 additions / subtractions (values stay bounded: the
// products bring everything back below 2.5 p)
template <class P> __global__ __launch_bounds__(256) void mix29(const Fp29<P>* in, Fp29<P>* out, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fp29<P> x = in[i], y = in[i ^ 1], zz = in[i ^ 2], zzz = in[i ^ 3];
  const Fp29<P> qx = in[i ^ 4], qy = in[i ^ 5];
  for (int k = 0; k < iters; k++) {
    const Fp29<P> u2 = mul29(qx, zz), s2 = mul29(qy, zzz);
    const Fp29<P> pp_ = sub29<P, 4>(u2, x), r_ = sub29<P, 4>(s2, y);
    const Fp29<P> pp = mul29(pp_, pp_), ppp = mul29(pp_, pp), qq = mul29(x, pp), rr = mul29(r_, r_);
    x = sub29<P, 8>(sub29<P, 4>(rr, ppp), add29(qq, qq));
    y = sub29<P, 4>(mul29(r_, sub29<P, 16>(qq, x)), mul29(y, ppp));
    zz = mul29(zz, pp);
    zzz = mul29(zzz, ppp);
  }
  out[i] = add29(add29(x, y), add29(zz, zzz));
}
template <class F> __global__ __launch_bounds__(256) void mix32(const F* in, F* out, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x = in[i], y = in[i ^ 1], zz = in[i ^ 2], zzz = in[i ^ 3];
  const F qx = in[i ^ 4], qy = in[i ^ 5];
  for (int k = 0; k < iters; k++) {
    const F u2 = mul(qx, zz), s2 = mul(qy, zzz);
    const F pp_ = sub(u2, x), r_ = sub(s2, y);
    const F pp = mul(pp_, pp_), ppp = mul(pp_, pp), qq = mul(x, pp), rr = mul(r_, r_);
    x = sub(sub(rr, ppp), dbl(qq));
    const F ab[2] = {r_, neg(y)}, cd[2] = {sub(qq, x), ppp};
    y = dot<typename F::params, 2>(ab, cd);
    zz = mul(zz, pp);
    zzz = mul(zzz, ppp);
  }
  out[i] = add(add(x, y), add(zz, zzz));
}
template <class K, class... A> float time_kernel(K kern, dim3 g, dim3 b, A... args) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}
template <class F> int check(const char* name) {
  const size_t n = (size_t)1 << 20;
  std::vector<F> ha(n), hb(n);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (size_t i = 0; i < n; i++) { for (int k = 0; k < 8; k++) ha[i].l[k] = rnd(), hb[i].l[k] = rnd(); ha[i].l[7] &= 0x3fffffffu; hb[i].l[7] &= 0x3fffffffu; }
  for (int k = 0; k < 8; k++) ha[0].l[k] = hb[0].l[k] = F::params::mod(k) - (k == 0), ha[1].l[k] = 0, ha[2].l[k] = k == 0;
  F *da, *db; uint32_t* dbad;
  (void)hipMalloc(&da, n * sizeof(F)); (void)hipMalloc(&db, n * sizeof(F)); (void)hipMalloc(&dbad, 4);
  (void)hipMemcpy(da, ha.data(), n * sizeof(F), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), n * sizeof(F), hipMemcpyHostToDevice);
  int fail = 0;
  const int ranges[4][2] = {{0, 0}, {1, 2}, {15, 15}, {15, 0}};
  for (auto& rg : ranges) {
    (void)hipMemset(dbad, 0, 4);
    hipLaunchKernelGGL(check_kernel<F>, n / 256, 256, 0, 0, da, db, dbad, n, rg[0], rg[1]);
    uint32_t bad = 1;
    (void)hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    printf("%s: 29-bit-limb product (operands + %d p, + %d p) vs CIOS on %zu inputs: %s (%u differ)\n", name, rg[0], rg[1], n,
           bad ? "DIFFER" : "identical", bad);
    fail |= bad ? 1 : 0;
  }
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dbad);
  return fail;
}
int main() {
  int bad = check<Fr>("Fr") | check<Fq>("Fq");
  const int iters = 256;
  const size_t n = (size_t)256 * 16 * 256;
  std::vector<Fq29> h29(n);
  std::vector<Fq> h32(n);
  for (size_t i = 0; i < n; i++) { for (int k = 0; k < 9; k++) h29[i].l[k] = (uint32_t)(i * 2654435761ull + k * 40503ull) & M29; h29[i].l[8] &= 0x1fffffu;
    for (int k = 0; k < 8; k++) h32[i].l[k] = (uint32_t)(i * 2654435761ull + k); h32[i].l[7] &= 0x1fffffffu; }
  Fq29 *d29, *o29; Fq *d32, *o32;
  (void)hipMalloc(&d29, n * sizeof(Fq29)); (void)hipMalloc(&o29, n * sizeof(Fq29)); (void)hipMalloc(&d32, n * sizeof(Fq)); (void)hipMalloc(&o32, n * sizeof(Fq));
  (void)hipMemcpy(d29, h29.data(), n * sizeof(Fq29), hipMemcpyHostToDevice); (void)hipMemcpy(d32, h32.data(), n * sizeof(Fq), hipMemcpyHostToDevice);
  const double prods = (double)n * iters;
  const float t29 = time_kernel(chain29<FqParams>, dim3(n / 256), dim3(256), (const Fq29*)d29, o29, iters);
  const float t32 = time_kernel(chain32<Fq>, dim3(n / 256), dim3(256), (const Fq*)d32, o32, iters);
  printf("Fq product chains (two per thread): 9 x 29-bit lazy-carry form %.1f G products/s, 8 x 32-bit product scanning %.1f G products/s (x%.2f)\n",
         prods / t29 / 1e6, prods / t32 / 1e6, t32 / t29);
  const int mi = 64;
  const float m29 = time_kernel(mix29<FqParams>, dim3(n / 256), dim3(256), (const Fq29*)d29, o29, mi);
  const float m32 = time_kernel(mix32<Fq>, dim3(n / 256), dim3(256), (const Fq*)d32, o32, mi);
  const double adds = (double)n * mi;
  printf("XYZZ mixed-addition arithmetic (10 products + 6 additions / subtractions per step): 29-bit form %.2f G steps/s, 32-bit form (with the shared reduction of Y3) %.2f G steps/s (x%.2f)\n",
         adds / m29 / 1e6, adds / m32 / 1e6, m32 / m29);
  return bad;
}
