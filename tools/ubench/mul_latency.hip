// Development micro-benchmark: latency of a dependent chain of Montgomery multiplications on a lone wave vs
// throughput on a full chip, for (A) the CIOS of ff.cuh, (B) two independent chains interleaved in one thread,
// (C) a row-wise formulation whose limb products are independent (carries resolved with 32-bit adds).
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/mul_latency.hip -o /tmp/mul_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include "ff.cuh"
using namespace lh;

template <class P>
__device__ __forceinline__ Fp<P> mul_rows(const Fp<P>& a, const Fp<P>& b) {
  uint32_t t[9];
#pragma unroll
  for (int j = 0; j < 9; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t bi = b.l[i];
    uint64_t p[8];
#pragma unroll
    for (int j = 0; j < 8; j++) p[j] = (uint64_t)a.l[j] * bi + t[j];  // independent
    // t = (p0.lo, p1.lo + p0.hi, ..., p7.hi + t8) with carries
    uint32_t lo0 = (uint32_t)p[0];
    uint64_t c = p[0] >> 32;
    uint32_t u[9];
    u[0] = lo0;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      uint64_t s = (uint64_t)(uint32_t)p[j] + c;
      u[j] = (uint32_t)s;
      c = (s >> 32) + (p[j] >> 32);
    }
    uint64_t s8 = (uint64_t)t[8] + c;
    u[8] = (uint32_t)s8;
    uint32_t u9 = (uint32_t)(s8 >> 32);
    const uint32_t m = u[0] * P::INV;
    uint64_t q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = (uint64_t)m * P::mod(j) + u[j];  // independent
    c = q[0] >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      uint64_t s = (uint64_t)(uint32_t)q[j] + c;
      t[j - 1] = (uint32_t)s;
      c = (s >> 32) + (q[j] >> 32);
    }
    s8 = (uint64_t)u[8] + c;
    t[7] = (uint32_t)s8;
    t[8] = u9 + (uint32_t)(s8 >> 32);
  }
  Fp<P> r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = t[j];
  return reduce_once(r);
}

template <int MODE>
__global__ void chain(const Fr* a, const Fr* b, int iters, Fr* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fr x = a[i], y = b[i];
  if (MODE == 0) {
    for (int k = 0; k < iters; k++) x = mul(x, y);
    out[i] = x;
  } else if (MODE == 1) {  // two independent chains
    Fr x2 = y;
    for (int k = 0; k < iters; k++) {
      x = mul(x, y);
      x2 = mul(x2, y);
    }
    out[i] = add(x, x2);
  } else if (MODE == 2) {
    for (int k = 0; k < iters; k++) x = mul_rows(x, y);
    out[i] = x;
  } else {  // four independent chains
    Fr x2 = y, x3 = add(x, y), x4 = add(x3, y);
    for (int k = 0; k < iters; k++) {
      x = mul(x, y);
      x2 = mul(x2, y);
      x3 = mul(x3, y);
      x4 = mul(x4, y);
    }
    out[i] = add(add(x, x2), add(x3, x4));
  }
}

template <int MODE>
static void run(const char* name, const Fr* a, const Fr* b, Fr* out, int blocks, int threads, int iters, int per) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  chain<MODE><<<blocks, threads>>>(a, b, iters, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  chain<MODE><<<blocks, threads>>>(a, b, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double muls = (double)blocks * threads * iters * per;
  printf("%-28s grid %6d x %3d: %8.3f ms  %7.1f ns per chain step  %7.2f G mul/s\n", name, blocks, threads, ms,
         ms * 1e6 / iters, muls / ms / 1e6);
}

int main() {
  const size_t n = 1 << 22;
  Fr *a, *b, *out;
  hipMalloc(&a, n * sizeof(Fr));
  hipMalloc(&b, n * sizeof(Fr));
  hipMalloc(&out, n * sizeof(Fr));
  hipMemset(a, 0x11, n * sizeof(Fr));
  hipMemset(b, 0x07, n * sizeof(Fr));
  const int iters = 2000;
  for (int blocks : {1, 256, 1024, 4096, 16384}) {
    for (int threads : {64, 256}) {
      run<0>("CIOS dependent", a, b, out, blocks, threads, iters, 1);
      run<1>("CIOS x2 interleaved", a, b, out, blocks, threads, iters, 2);
      run<3>("CIOS x4 interleaved", a, b, out, blocks, threads, iters, 4);
      run<2>("rows (independent mads)", a, b, out, blocks, threads, iters, 1);
    }
  }
  return 0;
}
