// Development micro-benchmark: what does FETCH_SIZE count for the bucket accumulation's access shape?
// msm_accumulate0 gathers one 64-byte base (four 16-byte loads of one lane) per sorted entry from an SRS level far larger
// than any cache.  MI355X_MICROARCH.md calibrates FETCH_SIZE for wide coalesced streams only (it reports HALF their bytes:
// 128-byte requests tallied at 64); "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern".  This is that calibration: N lanes read N x 64 B
//   stream64   record i by lane i                (coalesced: 4 KB per wave)
//   gather64   a random record per lane          (the accumulation's shape; table 1 GB)
//   gather128  a random ALIGNED PAIR of records  (128 B per lane: what a full-line fetch would be)
// under `rocprofv3 --pmc FETCH_SIZE` (tools/gather_calib.sh prints counter / known bytes).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/gather_calib.hip -o tools/ubench/gather_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Rec { uint4 w[4]; };  // 64 bytes, like a G1Affine

__global__ void stream64(const Rec* __restrict__ tab, const unsigned* __restrict__ idx, size_t n, unsigned* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Rec r = tab[i];
  out[i] = r.w[0].x ^ r.w[1].y ^ r.w[2].z ^ r.w[3].w;
}
__global__ void gather64(const Rec* __restrict__ tab, const unsigned* __restrict__ idx, size_t n, unsigned* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Rec r = tab[idx[i]];
  out[i] = r.w[0].x ^ r.w[1].y ^ r.w[2].z ^ r.w[3].w;
}
__global__ void gather128(const Rec* __restrict__ tab, const unsigned* __restrict__ idx, size_t n, unsigned* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t j = (size_t)(idx[i] & ~1u);
  const Rec r = tab[j], s = tab[j + 1];
  out[i] = r.w[0].x ^ r.w[1].y ^ r.w[2].z ^ r.w[3].w ^ s.w[0].x ^ s.w[3].w;
}

int main() {
  const size_t n = (size_t)1 << 24;
  Rec* tab;
  unsigned *idx, *out;
  CK(hipMalloc(&tab, n * sizeof(Rec)));
  CK(hipMalloc(&idx, n * 4));
  CK(hipMalloc(&out, n * 4));
  CK(hipMemset(tab, 1, n * sizeof(Rec)));
  std::vector<unsigned> h(n);
  unsigned long long s = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    h[i] = (unsigned)(s >> 20) & (unsigned)(n - 1);
  }
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const dim3 g((unsigned)(n / 256)), b(256);
  for (int rep = 0; rep < 3; rep++) {
    float ms[3];
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream64, g, b, 0, 0, tab, idx, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[0], e0, e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather64, g, b, 0, 0, tab, idx, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[1], e0, e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather128, g, b, 0, 0, tab, idx, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[2], e0, e1));
    printf("n = 2^24 records of 64 B: stream64 %.3f ms (%.2f TB/s)  gather64 %.3f ms (%.2f TB/s of records)  gather128 %.3f ms (%.2f TB/s)\n",
           ms[0], n * 64e-9 / ms[0], ms[1], n * 64e-9 / ms[1], ms[2], n * 128e-9 / ms[2]);
  }
  return 0;
}
