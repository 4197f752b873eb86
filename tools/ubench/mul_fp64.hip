// the FP64 route to a 256-bit Montgomery product, MEASURED.  5 x 52-bit limbs held as doubles, the
// 52 x 52 -> 104-bit limb products by two v_fma_f64 each under round-toward-zero (hi = fma(x, y, 2^104), lo = fma(x, y,
// 2^104 + 2^52 - hi)), their raw bit patterns accumulated straight into 64-bit integer columns (no per-product
// extraction), R = 2^260 (mul_fp64_core.h; the same source is checked on the host against big-integer arithmetic by
// tools/ubench/mul_fp64_host_check.py).  Here: (1) bit check on the device against ff.cuh mul_cios for both fields
// (x 2^4: the two Montgomery radices differ), (2) throughput of dependent product chains (two per thread, as
// lh_fr_mul_chain measures the integer form) next to the product-scanning v_mad_u64_u32 form, (3) the issue rates of the
// instructions the two forms are made of.
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/mul_fp64.hip -o tools/ubench/mul_fp64.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "ff.cuh"
#define FP64_HD __host__ __device__ __forceinline__
#define F52_FMA(a, b, c) __builtin_fma(a, b, c)
#include "mul_fp64_core.h"
using namespace lh;

__device__ __forceinline__ void round_toward_zero_f64() {
  // MODE register, FP_ROUND for f64 / f16 = bits [3:2] <- 3 (toward zero).  As inline assembly: the compiler TRACKS writes
  // to MODE it can see (SIModeRegister) and, its double-precision operations being defined under round-to-nearest, puts
  // the default mode back in front of the first of them (measured: with __builtin_amdgcn_s_setreg the kernel ran in
  // round-to-nearest after all).  Nothing in these kernels needs the default mode.
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}
template <class F> __host__ __device__ F52 to_f52(const F& x) {  // canonical 8 x u32 integer -> 5 x 52-bit limbs
  uint64_t w[5] = {0, 0, 0, 0, 0};
  for (int bit = 0; bit < 256; bit++)
    if ((x.l[bit >> 5] >> (bit & 31)) & 1u) w[bit / 52] |= 1ull << (bit % 52);
  F52 o;
  for (int k = 0; k < 5; k++) o.l[k] = f52_from_int(w[k]);
  return o;
}
template <class F> __host__ __device__ F from_f52(const F52& x) {
  F o = F::zero();
  for (int k = 0; k < 5; k++) {
    const uint64_t v = f52_bits(x.l[k] + 0x1p52) & F52_M;  // (no double -> integer conversion on the device: see mul_fp64_core.h)
    for (int b = 0; b < 52; b++)
      if ((v >> b) & 1ull) {
        const int bit = 52 * k + b;
        if (bit < 256) o.l[bit >> 5] |= 1u << (bit & 31);
      }
  }
  return o;
}
template <class F> F52Mod make_mod() {
  F n;
  for (int i = 0; i < 8; i++) n.l[i] = F::params::mod(i);
  F52 nl = to_f52(n);
  F52Mod m;
  for (int k = 0; k < 5; k++) m.n[k] = nl.l[k], m.ni[k] = (uint64_t)nl.l[k];
  // -n^-1 mod 2^52 by Newton iteration on the low limb
  const uint64_t n0 = (uint64_t)nl.l[0];
  uint64_t inv = 1;
  for (int it = 0; it < 7; it++) inv *= 2 - n0 * inv;
  m.np = (double)((0 - inv) & F52_M);
  return m;
}
template <class F> __global__ void check_kernel(const F* a, const F* b, F52Mod m, uint32_t* bad, size_t n) {
  round_toward_zero_f64();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F x = reduce_once_generic(a[i]), y = reduce_once_generic(b[i]);
  const F ref = mul_cios(x, y);                                // x y 2^-256
  F got = from_f52<F>(f52_mont_mul(to_f52(x), to_f52(y), m));  // x y 2^-260
  for (int k = 0; k < 4; k++) got = add_generic(got, got);
  if (!(got == ref)) atomicAdd(bad, 1u);
}
__global__ __launch_bounds__(256) void chain_fp64(const F52* in, F52* out, F52Mod m, int iters) {
  round_toward_zero_f64();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F52 x = in[i], y = in[i ^ 1], u = in[i ^ 2], v = in[i ^ 3];
  for (int k = 0; k < iters; k += 2) {
    x = f52_mont_mul(x, y, m);
    u = f52_mont_mul(u, v, m);
  }
  F52 o;
  for (int k = 0; k < 5; k++) o.l[k] = x.l[k] + u.l[k];
  out[i] = o;
}
template <class F> __global__ __launch_bounds__(256) void chain_int(const F* in, F* out, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x = in[i], y = in[i ^ 1], u = in[i ^ 2], v = in[i ^ 3];
  for (int k = 0; k < iters; k += 2) {
    x = mul(x, y);
    u = mul(u, v);
  }
  out[i] = add(x, u);
}
// issue rates: N independent-ish instructions per lane, everything in registers
__global__ void rate_mad(uint64_t* out, int iters) {
  uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t x = threadIdx.x * 2654435761u + 1, y = x ^ 0x9e3779b9u;
  for (int k = 0; k < iters; k++) {
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void rate_fma64(double* out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  const double x = 1.0000001, y = 0.9999999;
  for (int k = 0; k < iters; k++) {
    asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void rate_add64(uint64_t* out, int iters) {
  uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  const uint64_t x = 0x123456789abcdefull + threadIdx.x;
  for (int k = 0; k < iters; k++) {
    asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void rate_addc(uint32_t* out, int iters) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  const uint32_t x = 0x9e3779b9u + threadIdx.x;
  for (int k = 0; k < iters; k++) {
    asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <class K, class... A> float time_kernel(K kern, dim3 g, dim3 b, A... args) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
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
  hipMalloc(&da, n * sizeof(F)); hipMalloc(&db, n * sizeof(F)); hipMalloc(&dbad, 4); hipMemset(dbad, 0, 4);
  hipMemcpy(da, ha.data(), n * sizeof(F), hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * sizeof(F), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check_kernel<F>, n / 256, 256, 0, 0, da, db, make_mod<F>(), dbad, n);
  uint32_t bad = 1;
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("%s: FP64-FMA Montgomery product vs CIOS on %zu inputs: %s (%u differ)\n", name, n, bad ? "DIFFER" : "identical", bad);
  hipFree(da); hipFree(db); hipFree(dbad);
  return bad ? 1 : 0;
}
int main() {
  int bad = check<Fr>("Fr") | check<Fq>("Fq");
  // throughput: 256 CUs x 16 workgroups x 256 threads, two dependent chains of 128 products each per thread
  const int iters = 256;
  const size_t n = (size_t)256 * 16 * 256;
  std::vector<F52> h52(n);
  std::vector<Fr> hfr(n);
  for (size_t i = 0; i < n; i++) { for (int k = 0; k < 5; k++) h52[i].l[k] = (double)((i * 2654435761ull + k * 40503ull) & F52_M); h52[i].l[4] = (double)(i & 0xffff);
    for (int k = 0; k < 8; k++) hfr[i].l[k] = (uint32_t)(i * 2654435761ull + k); hfr[i].l[7] &= 0x1fffffffu; }
  F52 *d52, *o52; Fr *dfr, *ofr;
  hipMalloc(&d52, n * sizeof(F52)); hipMalloc(&o52, n * sizeof(F52)); hipMalloc(&dfr, n * sizeof(Fr)); hipMalloc(&ofr, n * sizeof(Fr));
  hipMemcpy(d52, h52.data(), n * sizeof(F52), hipMemcpyHostToDevice); hipMemcpy(dfr, hfr.data(), n * sizeof(Fr), hipMemcpyHostToDevice);
  const float ms52 = time_kernel(chain_fp64, dim3(n / 256), dim3(256), (const F52*)d52, o52, make_mod<Fr>(), iters);
  const float msint = time_kernel(chain_int<Fr>, dim3(n / 256), dim3(256), (const Fr*)dfr, ofr, iters);
  const double prods = (double)n * iters;
  printf("Fr product chains (two per thread): FP64-FMA form %.1f G products/s, v_mad_u64_u32 product scanning %.1f G products/s (x%.2f)\n",
         prods / ms52 / 1e6, prods / msint / 1e6, msint / ms52);
  // instruction issue rates (wave64 instructions per SIMD cycle; 1024 SIMDs at the measured clock are assumed 2.4 GHz)
  const int ri = 4096;
  uint64_t* d64; hipMalloc(&d64, n * 8);
  const double lane_ops = (double)n * ri * 4;
  const float t_mad = time_kernel(rate_mad, dim3(n / 256), dim3(256), d64, ri);
  const float t_fma = time_kernel(rate_fma64, dim3(n / 256), dim3(256), (double*)d64, ri);
  const float t_a64 = time_kernel(rate_add64, dim3(n / 256), dim3(256), d64, ri);
  const float t_adc = time_kernel(rate_addc, dim3(n / 256), dim3(256), (uint32_t*)d64, ri);
  auto cyc = [&](float ms) { return (double)ms * 1e-3 * 2.4e9 * 1024.0 / (lane_ops / 64.0); };
  printf("issue cost per wave64 instruction (cycles of one SIMD at 2.4 GHz): v_mad_u64_u32 %.2f, v_fma_f64 %.2f, v_lshl_add_u64 %.2f, v_add_co/addc_co_u32 %.2f\n",
         cyc(t_mad), cyc(t_fma), cyc(t_a64), cyc(t_adc));
  printf("T lane-ops/s: v_mad_u64_u32 %.1f, v_fma_f64 %.1f, v_lshl_add_u64 %.1f, v_add(c)_co_u32 %.1f\n", lane_ops / t_mad / 1e9,
         lane_ops / t_fma / 1e9, lane_ops / t_a64 / 1e9, lane_ops / t_adc / 1e9);
  return bad;
}
