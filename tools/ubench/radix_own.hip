// Development micro-benchmark ("own the sort"): a hand-written stable LSD radix sort of (u32 key,
// u32 value) pairs against rocPRIM's radix_sort_pairs on the shapes the MSM and the Lasso access counters sort: 2^22..2^24
// pairs, 16-20 significant key bits.
//
// Design: 8-bit passes; per pass three kernels - per-tile digit histogram -> exclusive scan of the (digit, tile) matrix ->
// scatter.  What it tries against the library's onesweep (4 B histogram read + 16 B per pass, tiles of ~3-4 K pairs whose
// output runs are ~12 pairs = 48 B per array): BIG tiles (8192 pairs per workgroup) ranked by wave-level digit matching and
// reordered through LDS, so that a workgroup writes runs of ~32 pairs (128 B per array) per digit - whole cache lines.
// Traffic: 4 B (histogram) + 8 B + 8 B per pass = 20 B per pair and pass.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/radix_own.hip -o /tmp/radix_own ; run: /tmp/radix_own [log_n] [bits]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef OWN_BS
#define OWN_BS 512
#endif
#ifndef OWN_IPT
#define OWN_IPT 16
#endif
constexpr int BS = OWN_BS;         // threads per workgroup
constexpr int IPT = OWN_IPT;       // pairs per thread
constexpr int TILE = BS * IPT;     // 8192 pairs per workgroup
constexpr int NW = BS / 64;        // waves per workgroup
constexpr int CHUNKS = IPT * NW;   // (row, wave) chunks of 64 consecutive pairs: the unit of ranking
constexpr int RADIX = 256;

// ---- kernel A: digit histogram of every tile, written bin-major: hist[bin * ntiles + tile]
__global__ __launch_bounds__(BS) void hist_kernel(const uint32_t* __restrict__ keys, size_t n, unsigned shift, uint32_t ntiles,
                                                  uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[RADIX];
  if (threadIdx.x < RADIX) h[threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * TILE;
#pragma unroll
  for (int i = 0; i < IPT; i++) {
    const size_t k = base + (size_t)i * BS + threadIdx.x;
    if (k < n) atomicAdd(&h[(keys[k] >> shift) & 0xff], 1u);
  }
  __syncthreads();
  if (threadIdx.x < RADIX) hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// ---- kernel B: exclusive scan of the whole bin-major matrix (256 * ntiles counters) = global start of every (bin, tile)
// one workgroup per bin scans its row; the bins' totals are scanned by the last workgroup to finish (a ticket)
__global__ __launch_bounds__(256) void scan_rows_kernel(uint32_t* __restrict__ hist, uint32_t ntiles, uint32_t* __restrict__ bin_total) {
  __shared__ uint32_t part[256];
  uint32_t* row = hist + (size_t)blockIdx.x * ntiles;
  const uint32_t per = (ntiles + 255) / 256, lo = threadIdx.x * per, hi = min(lo + per, ntiles);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += row[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int i = 0; i < 256; i++) {
      const uint32_t v = part[i];
      part[i] = acc;
      acc += v;
    }
    bin_total[blockIdx.x] = acc;
  }
  __syncthreads();
  uint32_t acc = part[threadIdx.x];
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t v = row[i];
    row[i] = acc;
    acc += v;
  }
}
__global__ void scan_bins_kernel(const uint32_t* __restrict__ bin_total, uint32_t* __restrict__ bin_base) {
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int b = 0; b < RADIX; b++) {
      bin_base[b] = acc;
      acc += bin_total[b];
    }
  }
}

// ---- kernel C: stable scatter of one tile
// LDS: cnt[256][CHUNKS] u16 (64 KB) + the reordered tile (64 KB)
__global__ __launch_bounds__(BS) void scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n,
                                                     unsigned shift, uint32_t ntiles, const uint32_t* __restrict__ hist,
                                                     const uint32_t* __restrict__ bin_base) {
  extern __shared__ uint32_t lds[];
  // cnt[digit][chunk] as u16; every run of PER = 64 counters (one thread's share of the scan) is followed by one u32 of
  // padding, so that the threads of a wave walk their runs in different LDS banks
  uint16_t* cnt = (uint16_t*)lds;
  constexpr int PER = RADIX * CHUNKS / BS;                         // 64
  constexpr int CNT_WORDS = BS * (PER / 2 + 1);                    // u32 words incl. padding
  uint32_t* skey = lds + CNT_WORDS;                                // [TILE]
  uint32_t* sval = skey + TILE;                                    // [TILE]
#define CNT_AT(e) cnt[(e) + ((e) / PER) * 2]
  __shared__ uint32_t bin_start[RADIX + 1];                        // start of every bin inside the sorted tile
  __shared__ uint32_t gbase[RADIX];                                // global position of the tile's first pair of every bin
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * TILE;
  for (int i = threadIdx.x; i < CNT_WORDS; i += BS) lds[i] = 0;
  if (threadIdx.x < RADIX) gbase[threadIdx.x] = bin_base[threadIdx.x] + hist[(size_t)threadIdx.x * ntiles + blockIdx.x];
  uint32_t key[IPT], val[IPT];
  uint16_t rank[IPT];
#pragma unroll
  for (int i = 0; i < IPT; i++) {
    const size_t k = base + (size_t)i * BS + threadIdx.x;
    key[i] = k < n ? keys_in[k] : 0xffffffffu;
    val[i] = k < n ? vals_in[k] : 0u;
  }
  __syncthreads();
  // rank inside the (row, wave) chunk among the pairs of the same digit; the chunk's per-digit counts
#pragma unroll
  for (int i = 0; i < IPT; i++) {
    const size_t k = base + (size_t)i * BS + threadIdx.x;
    const uint32_t d = k < n ? (key[i] >> shift) & 0xff : 0x100u;  // (padding matches nothing)
    unsigned long long peers = __ballot(k < n);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    rank[i] = (uint16_t)__popcll(peers & lt);
    if (k < n && (peers & lt) == 0) CNT_AT(d * CHUNKS + i * NW + wave) = (uint16_t)__popcll(peers);  // the digit's first lane
  }
  __syncthreads();
  // exclusive scan of cnt in (digit, chunk) order: RADIX * CHUNKS = 32768 counters, 64 per thread
  {
    uint16_t* mine = cnt + threadIdx.x * (PER + 2);
    uint32_t s = 0;
#pragma unroll 16
    for (int i = 0; i < PER; i++) s += mine[i];
    // block scan of the 512 partial sums: inside a wave by shuffles, across the 8 waves through LDS
    __shared__ uint32_t wsum[NW];
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(inc, off, 64);
      if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wsum[w];
    uint32_t acc = wbase + inc - s;
#pragma unroll 16
    for (int i = 0; i < PER; i++) {
      const uint32_t v = mine[i];
      mine[i] = (uint16_t)acc;
      acc += v;
    }
  }
  __syncthreads();
  if (threadIdx.x < RADIX) bin_start[threadIdx.x] = CNT_AT(threadIdx.x * CHUNKS);
  if (threadIdx.x == 0) bin_start[RADIX] = (uint32_t)min((size_t)TILE, n - base);
  // reorder through LDS
#pragma unroll
  for (int i = 0; i < IPT; i++) {
    const size_t k = base + (size_t)i * BS + threadIdx.x;
    if (k < n) {
      const uint32_t d = (key[i] >> shift) & 0xff;
      const uint32_t pos = CNT_AT(d * CHUNKS + i * NW + wave) + rank[i];
      skey[pos] = key[i];
      sval[pos] = val[i];
    }
  }
  __syncthreads();
  const uint32_t count = bin_start[RADIX];
  for (uint32_t p = threadIdx.x; p < count; p += BS) {
    const uint32_t kk = skey[p], d = (kk >> shift) & 0xff;
    const uint32_t g = gbase[d] + (p - bin_start[d]);
    keys_out[g] = kk;
    vals_out[g] = sval[p];
  }
}

constexpr int LDS_BYTES = (BS * ((RADIX * CHUNKS / BS) / 2 + 1) + 2 * TILE) * 4;
struct OwnSort {
  uint32_t *hist = nullptr, *bin_total = nullptr, *bin_base = nullptr, *ktmp = nullptr, *vtmp = nullptr;
  size_t cap = 0;
  void prepare(size_t n) {
    const size_t ntiles = (n + TILE - 1) / TILE;
    CHECK(hipMalloc(&hist, RADIX * ntiles * 4));
    CHECK(hipMalloc(&bin_total, RADIX * 4));
    CHECK(hipMalloc(&bin_base, RADIX * 4));
    CHECK(hipMalloc(&ktmp, n * 4));
    CHECK(hipMalloc(&vtmp, n * 4));
    cap = n;
    CHECK(hipFuncSetAttribute((const void*)scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  }
  // result in k_out / v_out; k_in / v_in are preserved
  void sort(const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in, uint32_t* v_out, size_t n, unsigned bits) {
    const uint32_t ntiles = (uint32_t)((n + TILE - 1) / TILE);
    const unsigned passes = (bits + 7) / 8;
    const uint32_t *ki = k_in, *vi = v_in;
    for (unsigned p = 0; p < passes; p++) {
      // ping-pong so that the last pass lands in (k_out, v_out)
      uint32_t* ko = ((passes - 1 - p) & 1) ? ktmp : k_out;
      uint32_t* vo = ((passes - 1 - p) & 1) ? vtmp : v_out;
      hipLaunchKernelGGL(hist_kernel, dim3(ntiles), dim3(BS), 0, 0, ki, n, 8 * p, ntiles, hist);
      hipLaunchKernelGGL(scan_rows_kernel, dim3(RADIX), dim3(256), 0, 0, hist, ntiles, bin_total);
      hipLaunchKernelGGL(scan_bins_kernel, dim3(1), dim3(64), 0, 0, bin_total, bin_base);
      hipLaunchKernelGGL(scatter_kernel, dim3(ntiles), dim3(BS), LDS_BYTES, 0, ki, vi, ko, vo, n, 8 * p, ntiles, hist, bin_base);
      ki = ko, vi = vo;
    }
  }
};

static void time_split(OwnSort& own, const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in, uint32_t* v_out, size_t n) {
  const uint32_t ntiles = (uint32_t)((n + TILE - 1) / TILE);
  hipEvent_t e[5];
  for (auto& x : e) CHECK(hipEventCreate(&x));
  float best[4] = {1e9f, 1e9f, 1e9f, 1e9f};
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e[0]));
    hipLaunchKernelGGL(hist_kernel, dim3(ntiles), dim3(BS), 0, 0, k_in, n, 0u, ntiles, own.hist);
    CHECK(hipEventRecord(e[1]));
    hipLaunchKernelGGL(scan_rows_kernel, dim3(RADIX), dim3(256), 0, 0, own.hist, ntiles, own.bin_total);
    CHECK(hipEventRecord(e[2]));
    hipLaunchKernelGGL(scan_bins_kernel, dim3(1), dim3(64), 0, 0, own.bin_total, own.bin_base);
    CHECK(hipEventRecord(e[3]));
    hipLaunchKernelGGL(scatter_kernel, dim3(ntiles), dim3(BS), LDS_BYTES, 0, k_in, v_in, k_out, v_out, n, 0u, ntiles, own.hist, own.bin_base);
    CHECK(hipEventRecord(e[4]));
    CHECK(hipEventSynchronize(e[4]));
    for (int i = 0; i < 4; i++) {
      float ms;
      CHECK(hipEventElapsedTime(&ms, e[i], e[i + 1]));
      best[i] = std::min(best[i], ms);
    }
  }
  printf("   one pass: hist %.3f ms, scan rows %.3f, scan bins %.3f, scatter %.3f ms\n", best[0], best[1], best[2], best[3]);
}

int main(int argc, char** argv) {
  const int log_n = argc > 1 ? atoi(argv[1]) : 24;
  const size_t n = (size_t)1 << log_n;
  std::vector<uint32_t> hk(n), hv(n);
  uint32_t *k_in, *v_in, *k_a, *v_a, *k_b, *v_b;
  CHECK(hipMalloc(&k_in, n * 4)); CHECK(hipMalloc(&v_in, n * 4));
  CHECK(hipMalloc(&k_a, n * 4)); CHECK(hipMalloc(&v_a, n * 4));
  CHECK(hipMalloc(&k_b, n * 4)); CHECK(hipMalloc(&v_b, n * 4));
  OwnSort own;
  own.prepare(n);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (unsigned bits : {16u, 17u, 20u, 24u}) {
    if (argc > 2 && (unsigned)atoi(argv[2]) != bits) continue;
    for (int skew = 0; skew < 2; skew++) {
      uint64_t s = 88172645463325252ull + bits;
      for (size_t i = 0; i < n; i++) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        uint32_t k = (uint32_t)(s >> 20) & ((1u << bits) - 1u);
        if (skew && (s & 3) == 0) k = 7;  // a quarter of the pairs in one bucket
        hk[i] = k, hv[i] = (uint32_t)i;
      }
      CHECK(hipMemcpy(k_in, hk.data(), n * 4, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(v_in, hv.data(), n * 4, hipMemcpyHostToDevice));
      size_t tb = 0;
      CHECK(rocprim::radix_sort_pairs(nullptr, tb, k_in, k_a, v_in, v_a, n, 0u, bits, (hipStream_t)0));
      void* tmp;
      CHECK(hipMalloc(&tmp, tb));
      float best_lib = 1e9f, best_own = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        CHECK(rocprim::radix_sort_pairs(tmp, tb, k_in, k_a, v_in, v_a, n, 0u, bits, (hipStream_t)0));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best_lib = std::min(best_lib, ms);
        CHECK(hipEventRecord(e0));
        own.sort(k_in, k_b, v_in, v_b, n, bits);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best_own = std::min(best_own, ms);
      }
      std::vector<uint32_t> ka(n), va(n), kb(n), vb(n);
      CHECK(hipMemcpy(ka.data(), k_a, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(va.data(), v_a, n * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(kb.data(), k_b, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(vb.data(), v_b, n * 4, hipMemcpyDeviceToHost));
      const bool same = ka == kb && va == vb;
      const unsigned passes = (bits + 7) / 8;
      printf("[BS %d IPT %d] 2^%d pairs, %2u key bits%s: rocPRIM %.3f ms, own %.3f ms (%u passes, %.2f TB/s at 20 B per pair and pass) -> %s\n", BS, IPT, log_n,
             bits, skew ? " (skewed)" : "         ", best_lib, best_own, passes, 20.0 * n * passes / (best_own * 1e-3) / 1e12,
             same ? "same order" : "MISMATCH");
      if (!skew) time_split(own, k_in, k_b, v_in, v_b, n);
      CHECK(hipFree(tmp));
    }
  }
  return 0;
}
