// Device radix sort of (key, value) pairs: the MSM's bucket sort (util/arithmetic/msm.rs:117-181 files every point into its
// bucket with a serial loop; here the (bucket, point) entries of a batch are sorted) and the Lasso access counters' stable
// (address, lookup) sort.
//
// Hand-written since round 3 (tools/ubench/radix_own.hip holds the development harness and the comparison with rocPRIM's
// onesweep, which this replaced on the Lasso path): a stable LSD sort, <= 8 bits per pass, three launches per pass -
//   hist     per-tile digit histogram (LDS atomics; 512 threads x 8 keys), written bin-major
//   scan     exclusive scan of every bin's row over the tiles, and the bin totals
//   scatter  a tile of 4096 pairs is ranked by wave-level digit matching (ballots), reordered through LDS and written out as
//            one run per digit: coalesced stores whatever the digit distribution; two workgroups share a CU's LDS so that one
//            tile's ranking overlaps another's loads and stores
// 20 B per pair and pass (4 B histogram read, 8 B in, 8 B out) against onesweep's 16 B + a shared 4 B; 2^24 pairs with
// 16-bit keys: 0.27 ms against 0.31 ms for the library (MI355X), identical (stable) order.  Key widths that are not a
// multiple of 8 are split evenly (17 bits: 6 + 6 + 5), which shrinks the ranking tables and lengthens the runs.
// Keys are u32 or u64 (the sharded access counters sort a 37-45-bit (address, global index) key); a batch of independent
// sorts - the (job, window) slabs of an MSM batch - runs as ONE launch set per pass.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "dev.hpp"

namespace lh {

namespace {
constexpr int RS_BS = 1024;              // threads per workgroup
constexpr int RS_IPT = 4;                // pairs per thread
constexpr int RS_TILE = RS_BS * RS_IPT;  // 4096 pairs per workgroup
constexpr int RS_NW = RS_BS / 64;        // waves per workgroup
constexpr int RS_CHUNKS = RS_IPT * RS_NW;  // (row, wave) chunks of 64 consecutive pairs: the unit of ranking

// One sort of a batch (device copy of a slab's plan).  A batch runs as ONE launch set per pass over the tiles of all its
// slabs (an MSM batch sorts a few dozen (job, window) slabs: three launches per pass instead of three per slab and pass).
struct RsSlab {
  const void* kin;            // keys (u32 or u64, the batch's key type), never written
  const uint32_t* vin;
  void* kout;                 // the sorted pairs land here
  uint32_t* vout;
  void* ktmp;                 // ping-pong partner (passes >= 2)
  uint32_t* vtmp;
  uint32_t* hist;             // 256 * ntiles counters, bin-major, then 256 bin totals
  uint64_t n;
  uint32_t ntiles, tile0;     // tiles of this slab, index of its first tile in the batch's flattened tile list
  uint32_t passes, rb[8];
  uint32_t shift0;            // first key bit of the sort (the passes cover bits [shift0, shift0 + sum rb))
};
// the buffers of pass q: (in) -> (out); the last pass lands in (kout, vout)
template <class K>
__device__ __forceinline__ void rs_buffers(const RsSlab& s, unsigned q, const K*& ki, const uint32_t*& vi, K*& ko, uint32_t*& vo) {
  const bool out_is_tmp = (s.passes - 1 - q) & 1, in_is_tmp = q > 0 && !out_is_tmp;
  ko = (K*)(out_is_tmp ? s.ktmp : s.kout), vo = out_is_tmp ? s.vtmp : s.vout;
  ki = (const K*)(q == 0 ? s.kin : in_is_tmp ? s.ktmp : s.kout), vi = q == 0 ? s.vin : in_is_tmp ? s.vtmp : s.vout;
}
__device__ __forceinline__ unsigned rs_shift(const RsSlab& s, unsigned q) {
  unsigned sh = s.shift0;
  for (unsigned i = 0; i < q; i++) sh += s.rb[i];
  return sh;
}
// flattened tile -> (slab, tile inside the slab): the slabs' first tiles ascend
__device__ __forceinline__ uint32_t rs_find_slab(const RsSlab* __restrict__ slabs, uint32_t count, uint32_t tile) {
  uint32_t lo = 0, hi = count;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (slabs[mid].tile0 <= tile) lo = mid;
    else hi = mid;
  }
  return lo;
}

// (512 threads x 8 keys, all loads first: with the scatter kernel's 1024 x 4 shape the histogram pass ran at the latency
// of two workgroups per CU - per launch 111 us on average in a 2^24 AND proof; 128 / 256 / 512 threads: 72 / 70 / 66 us)
template <class K, int RS_HIST_BS>
__global__ __launch_bounds__(RS_HIST_BS) void rs_hist_kernel(const RsSlab* __restrict__ slabs, uint32_t count, unsigned q) {
  __shared__ uint32_t h[256];
  const RsSlab& sl = slabs[rs_find_slab(slabs, count, blockIdx.x)];
  if (q >= sl.passes) return;
  const uint32_t tile = blockIdx.x - sl.tile0, ntiles = sl.ntiles;
  const unsigned rb = sl.rb[q], shift = rs_shift(sl, q);
  const size_t n = sl.n;
  const K* keys;
  const uint32_t* vi_;
  K* ko_;
  uint32_t* vo_;
  rs_buffers<K>(sl, q, keys, vi_, ko_, vo_);
  uint32_t* hist = sl.hist;
  const uint32_t radix = 1u << rb, mask = radix - 1u;
  for (uint32_t b = threadIdx.x; b < radix; b += RS_HIST_BS) h[b] = 0;
  __syncthreads();
  const size_t base = (size_t)tile * RS_TILE;
  K kk[RS_TILE / RS_HIST_BS];
#pragma unroll
  for (int i = 0; i < RS_TILE / RS_HIST_BS; i++) {  // (all loads first)
    const size_t k = base + (size_t)i * RS_HIST_BS + threadIdx.x;
    kk[i] = k < n ? keys[k] : (K)0;
  }
#pragma unroll
  for (int i = 0; i < RS_TILE / RS_HIST_BS; i++) {
    const size_t k = base + (size_t)i * RS_HIST_BS + threadIdx.x;
    if (k < n) atomicAdd(&h[(uint32_t)(kk[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < radix; b += RS_HIST_BS) hist[(size_t)b * ntiles + tile] = h[b];
}

// one workgroup per (bin, slab): exclusive scan of the bin's row (the tiles in order: stability across tiles), and its total
__global__ __launch_bounds__(256) void rs_scan_rows_kernel(const RsSlab* __restrict__ slabs, unsigned q) {
  __shared__ uint32_t part[256];
  const RsSlab& sl = slabs[blockIdx.y];
  if (q >= sl.passes || blockIdx.x >= (1u << sl.rb[q])) return;
  const uint32_t ntiles = sl.ntiles;
  uint32_t* bin_total = sl.hist + 256 * (size_t)ntiles;
  uint32_t* row = sl.hist + (size_t)blockIdx.x * ntiles;
  const uint32_t per = (ntiles + 255) / 256, lo = std::min(threadIdx.x * per, ntiles), hi = std::min(lo + per, ntiles);
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

// LDS of the scatter kernel: cnt[digit][chunk] as u16 - every run of PER counters (one thread's share of the scan) followed
// by one u32 of padding, so that the threads of a wave walk their runs in different banks - then the reordered tile
__host__ __device__ constexpr int rs_per(unsigned rb) { return ((1 << rb) * RS_CHUNKS + RS_BS - 1) / RS_BS; }
__host__ __device__ constexpr int rs_cnt_words(unsigned rb) { return (RS_BS * ((rs_per(rb) + 1) / 2 + 1) + 1) & ~1; }
static size_t rs_lds_bytes(unsigned rb, size_t key_bytes) { return (size_t)rs_cnt_words(rb) * 4 + RS_TILE * (key_bytes + 4); }

// XCD-aware tile order of the scatter pass: workgroups go to the 8 XCDs round-robin, each XCD with its own L2.  A tile's run
// for a digit is ~16 pairs (64 B) and continues, in the output, where the PREVIOUS tile's run ended: with tile = blockIdx
// the two halves of a 128-byte line are written through two different L2s, each evicting a partial line (a read-modify-write
// at the ECC-protected HBM).  Here workgroup b takes flattened tile (b mod 8)'s share + b / 8: an XCD works through a
// contiguous range of tiles in order, and neighbouring runs meet in ONE L2 before they leave it.
__device__ __forceinline__ uint32_t rs_xcd_tile(uint32_t b, uint32_t total, uint32_t xcd_order) {
  if (!xcd_order || total < 64) return b;
  const uint32_t qn = total / 8, r = total % 8, x = b % 8, j = b / 8;
  return x * qn + (x < r ? x : r) + j;  // (XCD x owns qn + (x < r) tiles: exactly the b with b mod 8 = x)
}
template <class K>
__global__ __launch_bounds__(RS_BS) void rs_scatter_kernel(const RsSlab* __restrict__ slabs, uint32_t num_slabs, unsigned q,
                                                           uint32_t xcd_order) {
  extern __shared__ uint32_t lds[];
  const uint32_t vb = rs_xcd_tile(blockIdx.x, gridDim.x, xcd_order);
  const RsSlab& sl = slabs[rs_find_slab(slabs, num_slabs, vb)];
  if (q >= sl.passes) return;
  const uint32_t tile = vb - sl.tile0, ntiles = sl.ntiles;
  const unsigned rb = sl.rb[q], shift = rs_shift(sl, q);
  const size_t n = sl.n;
  const K* keys_in;
  const uint32_t* vals_in;
  K* keys_out;
  uint32_t* vals_out;
  rs_buffers<K>(sl, q, keys_in, vals_in, keys_out, vals_out);
  const uint32_t* hist = sl.hist;
  const uint32_t* bin_total = sl.hist + 256 * (size_t)ntiles;
  const uint32_t radix = 1u << rb, mask = radix - 1u;
  const int per = rs_per(rb), stride = ((per + 1) / 2 + 1) * 2;  // u16 entries per thread's run incl. padding
  uint16_t* cnt = (uint16_t*)lds;
  K* skey = (K*)(lds + rs_cnt_words(rb));  // (an even number of words: 8-byte aligned)
  uint32_t* sval = (uint32_t*)(skey + RS_TILE);
  __shared__ uint32_t bin_start[257];  // start of every bin inside the sorted tile
  __shared__ uint32_t gbase[256];      // global position of the tile's first pair of every bin
  __shared__ uint32_t wsum[RS_NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = (size_t)tile * RS_TILE;
  auto cnt_at = [&](uint32_t e) -> uint16_t& { return cnt[e + (e / (uint32_t)per) * (uint32_t)(stride - per)]; };
  for (int i = threadIdx.x; i < rs_cnt_words(rb); i += RS_BS) lds[i] = 0;
  K key[RS_IPT];
  uint32_t val[RS_IPT];
  uint16_t rank[RS_IPT];
#pragma unroll
  for (int i = 0; i < RS_IPT; i++) {
    const size_t k = base + (size_t)i * RS_BS + threadIdx.x;
    key[i] = k < n ? keys_in[k] : (K)0;
    val[i] = k < n ? (vals_in ? vals_in[k] : (uint32_t)k) : 0u;  // (no values given: the pair's position)
  }
  // global bases: exclusive scan of the bin totals (256 values: the first four waves), plus this tile's offset in its bin
  if (threadIdx.x < 256) {
    const uint32_t t = threadIdx.x < radix ? bin_total[threadIdx.x] : 0u;
    uint32_t inc = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(inc, off, 64);
      if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    gbase[threadIdx.x] = inc - t;
  }
  __syncthreads();
  if (threadIdx.x < radix) {
    uint32_t wb = 0;
    for (int w = 0; w < wave; w++) wb += wsum[w];
    gbase[threadIdx.x] += wb + hist[(size_t)threadIdx.x * ntiles + tile];
  }
  __syncthreads();  // (wsum is reused below; cnt is cleared)
  // rank inside the (row, wave) chunk among the pairs of the same digit; the chunk's per-digit counts
#pragma unroll
  for (int i = 0; i < RS_IPT; i++) {
    const size_t k = base + (size_t)i * RS_BS + threadIdx.x;
    const bool live = k < n;
    const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
    unsigned long long peers = __ballot(live);
    for (unsigned b = 0; b < rb; b++) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    rank[i] = (uint16_t)__popcll(peers & lt);
    if (live && (peers & lt) == 0) cnt_at(d * RS_CHUNKS + i * RS_NW + wave) = (uint16_t)__popcll(peers);  // the digit's first lane
  }
  __syncthreads();
  // exclusive scan of cnt in (digit, chunk) order, `per` counters per thread
  {
    uint16_t* mine = cnt + threadIdx.x * stride;
    const int total = (int)radix * RS_CHUNKS;
    const int have = std::max(0, std::min(per, total - (int)threadIdx.x * per));
    uint32_t s = 0;
    for (int i = 0; i < have; i++) s += mine[i];
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(inc, off, 64);
      if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t acc = inc - s;
    for (int w = 0; w < wave; w++) acc += wsum[w];
    for (int i = 0; i < have; i++) {
      const uint32_t v = mine[i];
      mine[i] = (uint16_t)acc;
      acc += v;
    }
  }
  __syncthreads();
  if (threadIdx.x < radix) bin_start[threadIdx.x] = cnt_at(threadIdx.x * RS_CHUNKS);
  const uint32_t count = (uint32_t)std::min((size_t)RS_TILE, n - base);
  // reorder through LDS, then one coalesced run per digit
#pragma unroll
  for (int i = 0; i < RS_IPT; i++) {
    const size_t k = base + (size_t)i * RS_BS + threadIdx.x;
    if (k < n) {
      const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
      const uint32_t pos = cnt_at(d * RS_CHUNKS + i * RS_NW + wave) + rank[i];
      skey[pos] = key[i];
      sval[pos] = val[i];
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < count; p += RS_BS) {
    const K kk = skey[p];
    const uint32_t d = (uint32_t)(kk >> shift) & mask;
    const uint32_t g = gbase[d] + (p - bin_start[d]);
    keys_out[g] = kk;
    vals_out[g] = sval[p];
  }
}

struct RsPlan {
  unsigned passes, rb[8];
  size_t ntiles, hist_words, tmp_words, bytes;
};
RsPlan rs_plan(size_t n, unsigned bits, size_t key_bytes) {
  RsPlan p;
  bits = std::max(1u, std::min(bits, (unsigned)(8 * key_bytes)));
  p.passes = (bits + 7) / 8;
  unsigned left = bits;
  for (unsigned i = 0; i < 8; i++) p.rb[i] = 0;
  for (unsigned i = 0; i < p.passes; i++) {  // even split: 17 bits -> 6 + 6 + 5
    p.rb[i] = (left + (p.passes - i) - 1) / (p.passes - i);
    left -= p.rb[i];
  }
  p.ntiles = (n + RS_TILE - 1) / RS_TILE;
  p.hist_words = 256 * p.ntiles + 256;
  p.tmp_words = p.passes >= 2 ? n * (key_bytes / 4 + 1) : 0;  // keys then values
  p.bytes = (p.hist_words + p.tmp_words) * 4 + 512;
  return p;
}

// all slabs of a batch in one launch set per pass; `temp`: rs_batch_bytes() of device memory
struct RsJob {  // one sort of a batch, keys of the batch's type
  const void* keys_in;
  void* keys_out;
  const uint32_t* vals_in;
  uint32_t* vals_out;
  size_t n;
  unsigned bits;
  unsigned first_bit = 0;
};
size_t rs_batch_bytes(const RsJob* slabs, size_t count, size_t key_bytes) {
  size_t bytes = 256 + ((count * sizeof(RsSlab) + 255) & ~(size_t)255);
  for (size_t i = 0; i < count; i++) bytes += rs_plan(slabs[i].n, slabs[i].bits, key_bytes).bytes;
  return bytes;
}
template <class K>
void rs_sort_batch(Ctx& c, const RsJob* slabs, size_t count, void* temp) {
  const int side = 0;  // (one pinned descriptor staging per ctx)
  const size_t key_bytes = sizeof(K);
  hipStream_t stream = c.stream;
  std::vector<RsSlab> host;
  char* cur = (char*)(((uintptr_t)temp + 255) & ~(uintptr_t)255);
  RsSlab* d_slabs = (RsSlab*)cur;
  cur += (count * sizeof(RsSlab) + 255) & ~(size_t)255;
  uint32_t tiles = 0, max_passes = 0;
  for (size_t i = 0; i < count; i++) {
    if (!slabs[i].n) continue;
    const RsPlan p = rs_plan(slabs[i].n, slabs[i].bits, key_bytes);
    LH_REQUIRE(p.ntiles < ((size_t)1 << 31) && (size_t)tiles + p.ntiles < ((size_t)1 << 31), LH_ERR_ARG, "sort: too many pairs");
    RsSlab s;
    s.kin = slabs[i].keys_in, s.vin = slabs[i].vals_in, s.kout = slabs[i].keys_out, s.vout = slabs[i].vals_out;
    s.hist = (uint32_t*)cur;
    uint32_t* after = s.hist + ((p.hist_words + 1) & ~(size_t)1);  // (8-byte aligned for u64 keys)
    s.ktmp = after, s.vtmp = after + slabs[i].n * (key_bytes / 4);
    cur += p.bytes & ~(size_t)255;
    s.n = slabs[i].n, s.ntiles = (uint32_t)p.ntiles, s.tile0 = tiles, s.passes = p.passes;
    for (int k = 0; k < 8; k++) s.rb[k] = p.rb[k];
    s.shift0 = slabs[i].first_bit;
    tiles += s.ntiles;
    max_passes = std::max(max_passes, p.passes);
    host.push_back(s);
  }
  if (host.empty()) return;
  // (the descriptors go up through a pinned staging buffer of their own - Ctx::pin holds round messages and MSM tables that
  // may still be in flight; an async copy out of pageable memory would pin pages on the fly)
  const size_t stage_bytes = host.size() * sizeof(RsSlab);
  if (stage_bytes > c.sort_stage_bytes[side]) {
    if (c.sort_stage[side]) {
      LH_HIP(hipStreamSynchronize(stream));
      (void)hipHostFree(c.sort_stage[side]);
      c.sort_stage[side] = nullptr;
    }
    c.sort_stage_bytes[side] = std::max<size_t>(stage_bytes, 16384);
    LH_HIP(hipHostMalloc(&c.sort_stage[side], c.sort_stage_bytes[side], hipHostMallocDefault));
  }
  RsSlab* stage = (RsSlab*)c.sort_stage[side];
  LH_HIP(hipEventSynchronize(c.sort_stage_done(side)));  // (the previous batch's upload; long done in practice)
  memcpy(stage, host.data(), stage_bytes);
  LH_HIP(hipMemcpyAsync(d_slabs, stage, host.size() * sizeof(RsSlab), hipMemcpyHostToDevice, stream));
  c.opt_in_lds((const void*)rs_scatter_kernel<K>, (int)rs_lds_bytes(8, key_bytes));
  const uint32_t ns = (uint32_t)host.size();
  for (unsigned q = 0; q < max_passes; q++) {
    unsigned rb_max = 1;
    for (const RsSlab& s : host)
      if (q < s.passes) rb_max = std::max(rb_max, s.rb[q]);
    hipLaunchKernelGGL((rs_hist_kernel<K, 512>), dim3(tiles), dim3(512), 0, stream, d_slabs, ns, q);  // (128 / 256 threads: slower, round 3)
    hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1u << rb_max, ns), dim3(256), 0, stream, d_slabs, q);
    static const uint32_t xcd_order = !(getenv("LH_SORT_XCD_ORDER") && atoi(getenv("LH_SORT_XCD_ORDER")) == 0);  // (development A/B)
    hipLaunchKernelGGL(rs_scatter_kernel<K>, dim3(tiles), dim3(RS_BS), rs_lds_bytes(rb_max, key_bytes), stream, d_slabs, ns, q, xcd_order);
  }
  LH_HIP(hipGetLastError());
  // `stage` is reused by the next batch: its copy must have been consumed by then (the sorts themselves stay queued)
  LH_HIP(hipEventRecord(c.sort_stage_done(side), stream));
}
}  // namespace

void sort_pairs_u32_batched(Ctx& c, const SortSlab* slabs, size_t count) {
  std::vector<RsJob> jobs(count);
  for (size_t i = 0; i < count; i++)
    jobs[i] = RsJob{slabs[i].keys_in, slabs[i].keys_out, slabs[i].vals_in, slabs[i].vals_out, slabs[i].n, slabs[i].bits, slabs[i].first_bit};
  void* temp = c.arena.alloc(rs_batch_bytes(jobs.data(), count, 4));  // caller's ArenaScope releases it
  rs_sort_batch<uint32_t>(c, jobs.data(), count, temp);
}

void sort_pairs_u32(Ctx& c, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                    size_t n, unsigned bits, unsigned first_bit) {
  RsJob one{keys_in, keys_out, vals_in, vals_out, n, bits};
  one.first_bit = first_bit;
  void* temp = c.arena.alloc(rs_batch_bytes(&one, 1, 4));  // caller's ArenaScope releases it
  rs_sort_batch<uint32_t>(c, &one, 1, temp);
}

// 64-bit keys (the sharded access counters sort (address, global lookup index) on the address owner: 37-45 bits)
void sort_pairs_u64(Ctx& c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                    size_t n, unsigned bits) {
  const RsJob one{keys_in, keys_out, vals_in, vals_out, n, bits};
  void* temp = c.arena.alloc(rs_batch_bytes(&one, 1, 8));
  rs_sort_batch<uint64_t>(c, &one, 1, temp);
}

}  // namespace lh
