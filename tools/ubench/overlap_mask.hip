// Development micro-benchmark: can memory-bound work hide behind the ALU-bound bucket accumulation of an
// MSM when each runs on its own CU-masked stream (hipExtStreamCreateWithCUMask)?  An ALU-bound kernel (chains of Montgomery
// products, sized like one msm_accumulate0 launch, ~13 ms on the whole chip) and a memory-bound kernel (a streaming copy,
// sized like the radix-sort passes of one MSM batch, ~3 ms on the whole chip) are timed alone, back to back on one stream,
// concurrently on two unmasked streams, and concurrently with the chip split by CU masks (1/32 .. 1/4 of the CUs for the
// memory-bound stream).
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/overlap_mask.hip -o /tmp/overlap_mask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "ff.cuh"
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void alu_kernel(Fr* out, int iters) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  Fr a = Fr::one(), b = add(a, a);
  for (unsigned k = 0; k < (gid & 7u); k++) b = add(b, a);
  Fr c = b, d = add(b, a);
  for (int i = 0; i < iters; i++) {  // two independent chains per thread
    c = mul(c, b);
    d = mul(d, c);
  }
  out[gid] = add(c, d);
}
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int alu_blocks = cus * 8, alu_iters = 2600;                 // ~13 ms of products on the whole chip
  const size_t copy_n = (size_t)3 << 26;                            // 3 * 2^26 * 16 B = 3.2 GB read + 3.2 GB written per launch
  const int copy_reps = 2;
  Fr* d_out;
  uint4 *d_a, *d_b;
  CK(hipMalloc(&d_out, sizeof(Fr) * alu_blocks * 256));
  CK(hipMalloc(&d_a, copy_n * 16));
  CK(hipMalloc(&d_b, copy_n * 16));
  CK(hipMemset(d_a, 1, copy_n * 16));
  hipEvent_t e0, e1, ea, eb;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  hipStream_t s0;
  CK(hipStreamCreate(&s0));
  auto alu = [&](hipStream_t s) { hipLaunchKernelGGL(alu_kernel, dim3(alu_blocks), dim3(256), 0, s, d_out, alu_iters); };
  auto mem = [&](hipStream_t s) {
    for (int r = 0; r < copy_reps; r++)
      hipLaunchKernelGGL(copy_kernel, dim3(cus * 16), dim3(256), 0, s, (r & 1) ? d_b : d_a, (r & 1) ? d_a : d_b, copy_n);
  };
  auto timed = [&](auto&& fn) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s0));
      fn();
      CK(hipEventRecord(e1, s0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    return best;
  };
  const float t_alu = timed([&] { alu(s0); });
  const float t_mem = timed([&] { mem(s0); });
  const float t_serial = timed([&] { alu(s0); mem(s0); });
  printf("%d CUs; alone: ALU-bound %.2f ms, memory-bound %.2f ms (%.2f TB/s); back to back %.2f ms\n", cus, t_alu, t_mem,
         2.0 * copy_n * 16 * copy_reps / (t_mem * 1e-3) / 1e12, t_serial);
  // two streams, fork / join around s0
  auto concurrent = [&](hipStream_t sa, hipStream_t sb) {
    return timed([&] {
      CK(hipEventRecord(ea, s0));
      CK(hipStreamWaitEvent(sa, ea, 0));
      CK(hipStreamWaitEvent(sb, ea, 0));
      alu(sa);
      mem(sb);
      CK(hipEventRecord(ea, sa));
      CK(hipEventRecord(eb, sb));
      CK(hipStreamWaitEvent(s0, ea, 0));
      CK(hipStreamWaitEvent(s0, eb, 0));
    });
  };
  {
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    printf("two unmasked streams: %.2f ms\n", concurrent(sa, sb));
  }
  for (int frac : {32, 16, 8, 4}) {
    // the memory-bound stream gets every frac-th CU (spread over the XCDs), the ALU-bound stream the rest
    const int words = (cus + 31) / 32;
    std::vector<uint32_t> ma(words, 0), mb(words, 0);
    int nb = 0;
    for (int cu = 0; cu < cus; cu++) {
      const bool to_b = cu % frac == 0;
      (to_b ? mb : ma)[cu / 32] |= 1u << (cu % 32);
      nb += to_b;
    }
    hipStream_t sa, sb;
    if (hipExtStreamCreateWithCUMask(&sa, words, ma.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&sb, words, mb.data()) != hipSuccess) {
      printf("hipExtStreamCreateWithCUMask is not available here\n");
      return 0;
    }
    const float t_a = timed([&] { CK(hipEventRecord(ea, s0)); CK(hipStreamWaitEvent(sa, ea, 0)); alu(sa); CK(hipEventRecord(ea, sa)); CK(hipStreamWaitEvent(s0, ea, 0)); });
    const float t_b = timed([&] { CK(hipEventRecord(ea, s0)); CK(hipStreamWaitEvent(sb, ea, 0)); mem(sb); CK(hipEventRecord(ea, sb)); CK(hipStreamWaitEvent(s0, ea, 0)); });
    printf("masks %3d + %3d CUs: ALU-bound alone on its mask %.2f ms, memory-bound alone on its mask %.2f ms, both concurrently %.2f ms (back to back on the whole chip: %.2f)\n",
           cus - nb, nb, t_a, t_b, concurrent(sa, sb), t_serial);
  }
  return 0;
}
