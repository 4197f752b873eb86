// Development check + micro-benchmark: ff.cuh dot_scan<K> (sum of K products, ONE Montgomery reduction) against the sum of
// K separate CIOS products on random and extreme inputs (every operand p - 1: the largest unreduced result), both fields,
// and its throughput against K mul + K - 1 add.
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/dot_check.hip -o tools/ubench/dot_check.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "ff.cuh"
using namespace lh;
template <class F, int K> __global__ void both(const F* a, const F* b, F* o0, F* o1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x[K], y[K];
  F ref = F::zero();
#pragma unroll
  for (int j = 0; j < K; j++) {
    x[j] = reduce_once_generic(a[i * K + j]), y[j] = reduce_once_generic(b[i * K + j]);
    ref = add_generic(ref, mul_cios(x[j], y[j]));
  }
  o0[i] = ref;
  o1[i] = dot<typename F::params, K>(x, y);
}
template <class F, int K> __global__ void bench_dot(F* io, int reps) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x[K], y[K];
#pragma unroll
  for (int j = 0; j < K; j++) x[j] = io[(i + j) & 0xffff], y[j] = io[(i + 7 * j + 3) & 0xffff];
  for (int r = 0; r < reps; r++) {
    F d = dot<typename F::params, K>(x, y);
    x[r % K] = d;
  }
  io[i & 0xffff] = x[0];
}
template <class F, int K> __global__ void bench_mul(F* io, int reps) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F x[K], y[K];
#pragma unroll
  for (int j = 0; j < K; j++) x[j] = io[(i + j) & 0xffff], y[j] = io[(i + 7 * j + 3) & 0xffff];
  for (int r = 0; r < reps; r++) {
    F d = mul(x[0], y[0]);
#pragma unroll
    for (int j = 1; j < K; j++) d = add(d, mul(x[j], y[j]));
    x[r % K] = d;
  }
  io[i & 0xffff] = x[0];
}
template <class F, int K> int check(const char* name) {
  const size_t n = (size_t)1 << 18;
  std::vector<F> ha(n * K), hb(n * K), h0(n), h1(n);
  unsigned long long s = 88172645463325252ull + K;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (size_t i = 0; i < n * K; i++) { for (int k = 0; k < 8; k++) ha[i].l[k] = rnd(), hb[i].l[k] = rnd(); ha[i].l[7] &= 0x3fffffffu; hb[i].l[7] &= 0x3fffffffu; }
  // element 0: all operands p - 1 (the largest sum); element 1: zeros; element 2: one operand pair p - 1, the rest zero
  for (int j = 0; j < K; j++) for (int k = 0; k < 8; k++) {
    ha[j].l[k] = hb[j].l[k] = F::params::mod(k) - (k == 0);
    ha[K + j].l[k] = hb[K + j].l[k] = 0;
    ha[2 * K + j].l[k] = hb[2 * K + j].l[k] = j == 0 ? F::params::mod(k) - (k == 0) : 0;
  }
  F *da, *db, *d0, *d1;
  hipMalloc(&da, n * K * sizeof(F)); hipMalloc(&db, n * K * sizeof(F)); hipMalloc(&d0, n * sizeof(F)); hipMalloc(&d1, n * sizeof(F));
  hipMemcpy(da, ha.data(), n * K * sizeof(F), hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * K * sizeof(F), hipMemcpyHostToDevice);
  hipLaunchKernelGGL((both<F, K>), n / 256, 256, 0, 0, da, db, d0, d1);
  hipMemcpy(h0.data(), d0, n * sizeof(F), hipMemcpyDeviceToHost); hipMemcpy(h1.data(), d1, n * sizeof(F), hipMemcpyDeviceToHost);
  bool same = memcmp(h0.data(), h1.data(), n * sizeof(F)) == 0;
  printf("%s dot_scan<%d> vs sum of CIOS products on %zu inputs: %s\n", name, K, n, same ? "identical" : "DIFFER");
  hipFree(da); hipFree(db); hipFree(d0); hipFree(d1);
  return same ? 0 : 1;
}
template <class F, int K> void bench(const char* name) {
  F* io;
  hipMalloc(&io, 65536 * sizeof(F));
  hipMemset(io, 1, 65536 * sizeof(F));
  const int reps = 256, grid = 256 * 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms_dot = 0, ms_mul = 0;
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0, 0); hipLaunchKernelGGL((bench_dot<F, K>), grid, 256, 0, 0, io, reps); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_dot, e0, e1);
    hipEventRecord(e0, 0); hipLaunchKernelGGL((bench_mul<F, K>), grid, 256, 0, 0, io, reps); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_mul, e0, e1);
  }
  const double prods = (double)grid * 256 * reps * K;
  printf("%s K=%d: dot_scan %.1f G products/s, mul+add %.1f G products/s (x%.2f)\n", name, K, prods / ms_dot / 1e6, prods / ms_mul / 1e6, ms_mul / ms_dot);
  hipFree(io);
}
int main() {
  int bad = 0;
  bad |= check<Fr, 1>("Fr") | check<Fr, 2>("Fr") | check<Fr, 4>("Fr") | check<Fr, 5>("Fr") | check<Fr, 8>("Fr") | check<Fr, 10>("Fr") | check<Fr, 11>("Fr") | check<Fr, 16>("Fr");
  bad |= check<Fq, 2>("Fq") | check<Fq, 4>("Fq") | check<Fq, 8>("Fq") | check<Fq, 16>("Fq");
  bench<Fr, 2>("Fr"); bench<Fr, 4>("Fr"); bench<Fr, 8>("Fr"); bench<Fq, 2>("Fq");
  return bad;
}
