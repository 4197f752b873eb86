// Development micro-benchmark: where does the time of sc_round_u32<bind2> (kernels_poly.hip) go?  Variants of the kernel on a
// 2^24-entry column: as shipped, without the eq product, without the store, with a grid of one wave-set per CU x {4, 8, 16}.
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/u32_bind.hip -o tools/ubench/u32_bind.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>
#include "ff.cuh"
#include "reduce.cuh"
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Wide { uint32_t l[10]; };
__device__ __forceinline__ Wide wzero() { Wide w; for (int k = 0; k < 10; k++) w.l[k] = 0; return w; }
__device__ __forceinline__ void wide_mac(Wide& acc, const Fr& w, uint32_t v) {
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint64_t t = (uint64_t)w.l[k] * v + acc.l[k] + carry;
    acc.l[k] = (uint32_t)t;
    carry = t >> 32;
  }
  const uint64_t t = (uint64_t)acc.l[8] + carry;
  acc.l[8] = (uint32_t)t;
  acc.l[9] += (uint32_t)(t >> 32);
}
__device__ __forceinline__ Fr wide_redc(const Wide& acc) {
  uint32_t a[18];
#pragma unroll
  for (int k = 0; k < 10; k++) a[k] = acc.l[k];
#pragma unroll
  for (int k = 10; k < 18; k++) a[k] = 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t m = a[i] * FrParams::INV;
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t t = (uint64_t)m * FrParams::mod(j) + a[i + j] + carry;
      a[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
#pragma unroll
    for (int j = i + 8; j < 18; j++) {
      const uint64_t t = (uint64_t)a[j] + carry;
      a[j] = (uint32_t)t;
      carry = t >> 32;
    }
  }
  Fr r;
#pragma unroll
  for (int k = 0; k < 8; k++) r.l[k] = a[8 + k];
  return reduce_once(r);
}
struct W4 { Fr w[4]; };
template <int MODE>  // 0 full, 1 no eq product, 2 no store, 3 neither (loads + reduction only), 4 copy-shaped (no arithmetic)
__global__ __launch_bounds__(256) void bind2(const uint32_t* __restrict__ col, const Fr* __restrict__ eq, W4 k, size_t entries,
                                             Fr* __restrict__ out, Fr* __restrict__ partials, uint32_t* ticket, uint32_t seq, int getenv_batched) {
  Fr acc = Fr::zero();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < entries; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 a = ((const uint4*)col)[i];
    Fr v;
    if (MODE == 4) {
      v = k.w[0];
      v.l[0] = a.x ^ a.y, v.l[1] = a.z ^ a.w;
    } else {
      Wide t = wzero();
      wide_mac(t, k.w[0], a.x), wide_mac(t, k.w[1], a.y), wide_mac(t, k.w[2], a.z), wide_mac(t, k.w[3], a.w);
      v = wide_redc(t);
    }
    if (MODE == 0 || MODE == 1 || MODE == 4 || MODE >= 5) out[i] = v;
    if (MODE == 0 || MODE == 2 || MODE >= 5) {
      if (i & 1) acc = add(acc, mul(v, eq[i >> 1]));
    } else {
      acc = add(acc, v);
    }
  }
  if (MODE < 5) {
    partials[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    return;
  }
  // MODE 5: the shipped kernel's epilogue - block sum, ticket, the last workgroup adds the partials up
  __shared__ Fr lds[4];
  __shared__ int is_last;
  acc = block_reduce_sum(acc, lds);
  if (MODE == 7) {
    // tagged lanes: no fence anywhere.  Every limb of the block's sum leaves as one 8-byte agent-scope store (limb | seq << 32);
    // the workgroup that draws the last ticket polls the lanes until each carries this launch's tag
    uint64_t* lanes = (uint64_t*)(partials + 65536);
    if (threadIdx.x < 8) {
      __hip_atomic_store(&lanes[(size_t)blockIdx.x * 8 + threadIdx.x], (uint64_t)acc.l[threadIdx.x] | ((uint64_t)seq << 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) {
      const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    Fr a2 = Fr::zero();
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x) {
      Fr p;
      if (getenv_batched) {  // all eight loads in flight, then the tags
        uint64_t v[8];
        bool ok;
        do {
          ok = true;
#pragma unroll
          for (int k = 0; k < 8; k++) v[k] = __hip_atomic_load(&lanes[(size_t)i * 8 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int k = 0; k < 8; k++) ok = ok && (uint32_t)(v[k] >> 32) == seq;
        } while (!ok);
#pragma unroll
        for (int k = 0; k < 8; k++) p.l[k] = (uint32_t)v[k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          uint64_t v;
          do v = __hip_atomic_load(&lanes[(size_t)i * 8 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while ((uint32_t)(v >> 32) != seq);
          p.l[k] = (uint32_t)v;
        }
      }
      a2 = add(a2, p);
    }
    a2 = block_reduce_sum(a2, lds);
    if (threadIdx.x == 0) partials[gridDim.x] = a2, *ticket = 0;
    return;
  }
  if (MODE == 5) {
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = acc;
      __threadfence();
      const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      is_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
  } else {  // MODE 6: kernels_sumcheck.hip finish_round - release by every producer, acquire by the one consumer
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = acc;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == gridDim.x - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
  }
  Fr a2 = Fr::zero();
  for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x) a2 = add(a2, partials[i]);
  a2 = block_reduce_sum(a2, lds);
  if (threadIdx.x == 0) partials[gridDim.x] = a2, *ticket = 0;
}
static uint32_t* ticket;
static uint32_t g_seq = 0;
template <int MODE>
static void run(const char* name, const uint32_t* col, const Fr* eq, const W4& k, size_t entries, Fr* out, Fr* partials, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  const char* idle = getenv("IDLE_US");
  for (int rep = 0; rep < 5; rep++) {
    if (idle) {  // the chip idles before the launch, as it does while a proof's host side works
      CK(hipDeviceSynchronize());
      usleep(atoi(idle));
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bind2<MODE>, dim3(grid), dim3(256), 0, 0, col, eq, k, entries, out, partials, ticket, ++g_seq, getenv("BATCHED") ? 1 : 0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-40s grid %5d: %.3f ms (%.0f GB/s of 64 B per entry)\n", name, grid, best, 64.0 * entries / best / 1e6);
}
int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t N = (size_t)1 << (getenv("LOG_N") ? atoi(getenv("LOG_N")) : 24), entries = N / 4;
  uint32_t* col;
  Fr *eq, *out, *partials;
  CK(hipMalloc(&col, N * 4));
  CK(hipMalloc(&eq, entries / 2 * sizeof(Fr)));
  CK(hipMalloc(&out, entries * sizeof(Fr)));
  CK(hipMalloc(&partials, (size_t)cus * 64 * 256 * sizeof(Fr)));
  CK(hipMalloc(&ticket, 4));
  CK(hipMemset(ticket, 0, 4));
  CK(hipMemset(col, 0x5a, N * 4));
  CK(hipMemset(eq, 0x11, entries / 2 * sizeof(Fr)));
  W4 k;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) k.w[i].l[j] = 0x01234567u * (i + 1) + j;
  if (getenv("RANDOM_DATA")) {  // random column and eq entries (< 2^253), random weights
    std::vector<uint32_t> h(N), he(entries / 2 * 8);
    uint64_t x = 88172645463325252ull;
    auto next = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
    for (auto& v : h) v = next();
    for (size_t i = 0; i < he.size(); i++) he[i] = (i & 7) == 7 ? next() >> 3 : next();
    CK(hipMemcpy(col, h.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(eq, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 8; j++) k.w[i].l[j] = j == 7 ? next() >> 3 : next();
    printf("random data\n");
  }
  for (int g : {8}) {
    run<7>("epilogue: tagged lanes, no fence", col, eq, k, entries, out, partials, cus * g);
    run<6>("epilogue: release / relaxed ticket / one acquire", col, eq, k, entries, out, partials, cus * g);
    run<5>("with the shipped epilogue", col, eq, k, entries, out, partials, cus * g);
    run<0>("as shipped", col, eq, k, entries, out, partials, cus * g);
    run<1>("no eq product", col, eq, k, entries, out, partials, cus * g);
    run<2>("no store", col, eq, k, entries, out, partials, cus * g);
    run<3>("loads + reduction only", col, eq, k, entries, out, partials, cus * g);
    run<4>("copy-shaped (no arithmetic)", col, eq, k, entries, out, partials, cus * g);
  }
  return 0;
}
