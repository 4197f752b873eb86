// One asm block per column (ff.cuh mul_scan_cols / dot_scan_cols, ff_cols.inc) against one block per multiply-add
// (mul_scan / dot_scan): bit check on random and extreme inputs, both fields, products and 2- / 4-term dot products;
// throughput of dependent product chains (two per thread) and of the XYZZ mixed addition's arithmetic mix.
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/mul_cols.hip -o tools/ubench/mul_cols.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "ff.cuh"
using namespace lh;

template <class F> __global__ void check_kernel(const F* a, const F* b, uint32_t* bad, size_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename F::params P;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F x[4], y[4];
  for (int j = 0; j < 4; j++) x[j] = reduce_once_generic(a[(i + j) % n]), y[j] = reduce_once_generic(b[(i + 3 * j) % n]);
  bool ok = mul_scan_cols(x[0], y[0]) == mul_cios(x[0], y[0]) && mul_scan_cols(x[0], y[0]) == mul_scan(x[0], y[0]);
  ok = ok && mul_scan_cols(x[1], x[1]) == mul_cios(x[1], x[1]);
  ok = ok && dot_scan_cols<P, 2>(x, y) == dot_scan<P, 2>(x, y) && dot_scan_cols<P, 4>(x, y) == dot_scan<P, 4>(x, y);
  F s = mul_cios(x[0], y[0]);
  for (int j = 1; j < 4; j++) s = add_generic(s, mul_cios(x[j], y[j]));
  ok = ok && dot_scan_cols<P, 4>(x, y) == s;
  // operands with compile-time zero limbs (the to_mont of a 64-bit value): an input equal to top's zero on entry must not
  // share its register (the blocks' acc / top are early-clobber operands)
  F c = F::zero();
  c.l[0] = a[i].l[0], c.l[1] = a[i].l[1] | 0xd0000000u;
  ok = ok && mul_scan_cols(c, F::r2()) == mul_cios(c, F::r2());
  const F cs[2] = {c, F::one()}, ds[2] = {F::r2(), c};
  ok = ok && dot_scan_cols<P, 2>(cs, ds) == add_generic(mul_cios(c, F::r2()), mul_cios(F::one(), c));
  // the fused addition / subtraction / reduction blocks against the generic forms
  ok = ok && add(x[0], y[0]) == add_generic(x[0], y[0]) && add(x[1], x[1]) == add_generic(x[1], x[1]);
  ok = ok && sub(x[0], y[0]) == sub_generic(x[0], y[0]) && sub(y[0], x[0]) == sub_generic(y[0], x[0]) && sub(x[2], x[2]).is_zero();
  ok = ok && neg(x[3]) == sub_generic(F::zero(), x[3]) && reduce_once(a[i]) == reduce_once_generic(a[i]);
  if (!ok) atomicAdd(bad, 1u);
#endif
}
template <class F, bool COLS> __global__ __launch_bounds__(256) void chain(const F* in, F* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x = in[i], y = in[i ^ 1], u = in[i ^ 2], v = in[i ^ 3];
  for (int k = 0; k < iters; k += 2) {
    x = COLS ? mul_scan_cols(x, y) : mul_scan(x, y);
    u = COLS ? mul_scan_cols(u, v) : mul_scan(u, v);
  }
  out[i] = add(x, u);
#endif
}
template <class F, bool COLS> __global__ __launch_bounds__(256) void mix(const F* in, F* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename F::params P;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x = in[i], y = in[i ^ 1], zz = in[i ^ 2], zzz = in[i ^ 3];
  const F qx = in[i ^ 4], qy = in[i ^ 5];
  auto M = [](const F& p, const F& q) { return COLS ? mul_scan_cols(p, q) : mul_scan(p, q); };
  for (int k = 0; k < iters; k++) {
    const F u2 = M(qx, zz), s2 = M(qy, zzz);
    const F pp_ = sub(u2, x), r_ = sub(s2, y);
    const F pp = M(pp_, pp_), ppp = M(pp_, pp), qq = M(x, pp), rr = M(r_, r_);
    x = sub(sub(rr, ppp), dbl(qq));
    const F ab[2] = {r_, neg(y)}, cd[2] = {sub(qq, x), ppp};
    y = COLS ? dot_scan_cols<P, 2>(ab, cd) : dot_scan<P, 2>(ab, cd);
    zz = M(zz, pp);
    zzz = M(zzz, ppp);
  }
  out[i] = add(add(x, y), add(zz, zzz));
#endif
}
template <class K, class... A> float time_kernel(K kern, dim3 g, dim3 b, A... args) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
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
  for (size_t i = 0; i < n; i++) for (int k = 0; k < 8; k++) ha[i].l[k] = rnd(), hb[i].l[k] = rnd();
  for (size_t i = 0; i < n; i++) ha[i].l[7] &= 0x3fffffffu, hb[i].l[7] &= 0x3fffffffu;
  for (int k = 0; k < 8; k++) ha[0].l[k] = hb[0].l[k] = F::params::mod(k) - (k == 0), ha[1].l[k] = 0, ha[2].l[k] = k == 0, hb[3].l[k] = 0xffffffffu >> (k == 7 ? 2 : 0);
  F *da, *db; uint32_t* dbad;
  (void)hipMalloc(&da, n * sizeof(F)); (void)hipMalloc(&db, n * sizeof(F)); (void)hipMalloc(&dbad, 4);
  (void)hipMemcpy(da, ha.data(), n * sizeof(F), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), n * sizeof(F), hipMemcpyHostToDevice);
  (void)hipMemset(dbad, 0, 4);
  hipLaunchKernelGGL(check_kernel<F>, n / 256, 256, 0, 0, da, db, dbad, n);
  uint32_t bad = 1;
  (void)hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("%s: column-block products / dot products vs CIOS and the per-multiply-add form on %zu inputs: %s (%u differ)\n", name, n,
         bad ? "DIFFER" : "identical", bad);
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dbad);
  return bad ? 1 : 0;
}
template <class F> void bench(const char* name) {
  const int iters = 256;
  const size_t n = (size_t)256 * 16 * 256;
  std::vector<F> h(n);
  for (size_t i = 0; i < n; i++) { for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761ull + k); h[i].l[7] &= 0x1fffffffu; }
  F *d, *o;
  (void)hipMalloc(&d, n * sizeof(F)); (void)hipMalloc(&o, n * sizeof(F));
  (void)hipMemcpy(d, h.data(), n * sizeof(F), hipMemcpyHostToDevice);
  const float a0 = time_kernel(chain<F, false>, dim3(n / 256), dim3(256), d, o, iters);
  const float a1 = time_kernel(chain<F, true>, dim3(n / 256), dim3(256), d, o, iters);
  printf("%s product chains: per multiply-add %.3f ms = %.1f G products/s | per column %.3f ms = %.1f G products/s (x%.3f)\n", name,
         a0, n * (double)iters / a0 / 1e6, a1, n * (double)iters / a1 / 1e6, a0 / a1);
  const int it2 = 64;
  const float b0 = time_kernel(mix<F, false>, dim3(n / 256), dim3(256), d, o, it2);
  const float b1 = time_kernel(mix<F, true>, dim3(n / 256), dim3(256), d, o, it2);
  printf("%s mixed-addition mix: per multiply-add %.3f ms = %.2f G additions/s | per column %.3f ms = %.2f G additions/s (x%.3f)\n",
         name, b0, n * (double)it2 / b0 / 1e6, b1, n * (double)it2 / b1 / 1e6, b0 / b1);
  (void)hipFree(d); (void)hipFree(o);
}
int main() {
  int bad = check<Fr>("Fr") | check<Fq>("Fq");
  bench<Fr>("Fr");
  bench<Fq>("Fq");
  return bad;
}
