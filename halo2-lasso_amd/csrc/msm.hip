// Batched variable-base MSM over BN254 G1 for gfx950: sort-by-bucket Pippenger.
//
// Replaces variable_base_msm (reference plonkish_backend/src/util/arithmetic/msm.rs:84-181), whose
// chunk-per-thread Pippenger has no device analogue; only the affine sum is observable
// (SURVEY.md §3.4), so window size / bucket scheme are chosen for the GPU:
//
//  1. digits     every scalar -> canonical -> c-bit digits; histogram of (job, window, digit) keys
//  2. scan       exclusive prefix over the key histogram
//  3. scatter    (key, base index) pairs in key order (counting sort; order inside a bucket is free)
//  4. accumulate load-balanced segmented sum: every thread owns K CONSECUTIVE sorted entries
//                whatever the bucket sizes are (Lasso's read_ts / final_cts / dim columns are heavily
//                skewed - a thread-per-bucket scheme would serialise on the hot buckets).  The run that
//                starts a bucket is stored to the bucket array, a run that continues from the
//                previous thread's chunk goes to a continuation list that is reduced the same way
//                (K-fold shrink per level).
//  5. reduce     per (job, window): sum_d d*B[d] by 16-bucket segments (running sums + d0*T), then a
//                workgroup tree over the segments
//  6. combine    the W window sums go to the host: Horner with c doublings per window and ONE field
//                inversion to affine (a 254-doubling dependent chain is ~60 us on a CPU core and
//                milliseconds on a single GPU lane).
// Several MSMs (a batch_commit's polys, the n quotient commitments of one opening) run as ONE batch:
// the key space is (job, window, digit), so the latency-bound tails are paid once per batch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "dev.hpp"
#include "ff_host.hpp"

namespace lh {

constexpr int MSM_MAX_JOBS = 48;
constexpr uint32_t SENTINEL = 0xffffffffu;

struct MsmJobDev {
  const void* scalars;
  const G1Affine* bases;
  uint32_t is_u32;
  uint32_t n;
  uint32_t c, W;        // window bits, number of windows
  uint32_t key_base;    // first bucket key of this job
  uint32_t cnt_base;    // first histogram slot of this job (bucket keys are replicated 2^rep_log times)
  uint32_t rep_log;     // log2 of the histogram replication (spreads atomics of skewed columns)
  uint32_t seg_base;    // first reduce-segment of this job
  uint32_t seg_per_win; // segments per window
  uint32_t seg_size;    // buckets per segment
  uint32_t win_base;    // first window-sum slot of this job
};
struct MsmPlanDev {
  int num_jobs;
  MsmJobDev job[MSM_MAX_JOBS];
};

// ------------------------------------------------------------------ 1/3: digits, histogram, scatter
// Wave-aggregated atomic increment: lanes that hit the same slot elect a leader that adds the group
// size once and hands out consecutive ranks.  Lasso's committed columns are small-valued (read_ts is
// ~Poisson, final_cts/dim/E take few values), so whole waves collide on a handful of slots; a few
// rounds of aggregation peel off the popular slots, the rest falls back to one atomic per lane.
constexpr int AGG_ROUNDS = 6;
__device__ __forceinline__ uint32_t wave_agg_inc(uint32_t* arr, uint32_t slot, bool active) {
  const int lane = __lane_id();
  unsigned long long todo = __ballot(active);
  uint32_t res = 0;
#pragma unroll 1
  for (int round = 0; round < AGG_ROUNDS && todo; round++) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const uint32_t k = __shfl(slot, leader);
    const bool mine = active && slot == k;
    const unsigned long long same = __ballot(mine);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&arr[k], (uint32_t)__popcll(same));
    base = __shfl(base, leader);
    if (mine) {
      res = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
      active = false;
    }
    todo &= ~same;
  }
  if (active) res = atomicAdd(&arr[slot], 1u);
  return res;
}

template <bool SCATTER>
__device__ __forceinline__ void emit_digit(const MsmJobDev& jb, uint32_t w, uint32_t d, uint32_t rep, uint32_t i,
                                           uint32_t* __restrict__ counts_or_cursor, uint32_t* __restrict__ sorted_key,
                                           uint32_t* __restrict__ sorted_idx) {
  const uint32_t local = (w << jb.c) + d;
  const uint32_t slot = jb.cnt_base + (local << jb.rep_log) + rep;
  const uint32_t pos = wave_agg_inc(counts_or_cursor, slot, d != 0);
  if (SCATTER && d != 0) {
    sorted_key[pos] = jb.key_base + local;
    sorted_idx[pos] = i;
  }
}

template <bool SCATTER>
__global__ void msm_digits_kernel(MsmPlanDev plan, uint32_t* __restrict__ counts_or_cursor,
                                  uint32_t* __restrict__ sorted_key, uint32_t* __restrict__ sorted_idx) {
  const MsmJobDev& jb = plan.job[blockIdx.y];
  const uint32_t c = jb.c, mask = (1u << c) - 1u;
  const uint32_t rep = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & ((1u << jb.rep_log) - 1u);
  // every lane of a wave runs the same number of iterations (digits are emitted convergently)
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t iters = (jb.n + stride - 1) / stride;
  for (size_t it = 0; it < iters; it++) {
    const size_t i = it * stride + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < jb.n;
    uint32_t limb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) limb[k] = 0;
    if (jb.is_u32) {
      if (live) limb[0] = ((const uint32_t*)jb.scalars)[i];
    } else if (live) {
      Fr s = from_mont(((const Fr*)jb.scalars)[i]);
#pragma unroll
      for (int k = 0; k < 8; k++) limb[k] = s.l[k];
    }
    const int nlimbs = jb.is_u32 ? 1 : 8;
    uint64_t buf = 0;
    int have = 0;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k < nlimbs) {
        buf |= (uint64_t)limb[k] << have;
        have += 32;
        while (have >= (int)c && w < jb.W) {
          uint32_t d = (uint32_t)buf & mask;
          buf >>= c;
          have -= c;
          emit_digit<SCATTER>(jb, w, d, rep, (uint32_t)i, counts_or_cursor, sorted_key, sorted_idx);
          w++;
        }
      }
    }
    if (w < jb.W)  // top, partial window
      emit_digit<SCATTER>(jb, w, (uint32_t)buf & mask, rep, (uint32_t)i, counts_or_cursor, sorted_key, sorted_idx);
  }
}

// ------------------------------------------------------------------ 2: exclusive scan (u32)
constexpr int SCAN_TILE = 2048;  // 256 threads x 8
__global__ void scan_tile_sums_kernel(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t lds[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE;
  uint32_t s = 0;
  for (int k = 0; k < 8; k++) {
    size_t i = base + (size_t)threadIdx.x * 8 + k;
    if (i < n) s += in[i];
  }
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) lds[threadIdx.x] += lds[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = lds[0];
}
__global__ void scan_tile_offsets_kernel(uint32_t* __restrict__ tile_sums, size_t ntiles, uint32_t* __restrict__ total) {
  // single block: exclusive scan of tile_sums in place
  __shared__ uint32_t lds[256];
  size_t per = (ntiles + 255) / 256;
  size_t lo = threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
  uint32_t s = 0;
  for (size_t i = lo; i < hi; i++) s += tile_sums[i];
  lds[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; i++) {
      uint32_t v = lds[i];
      lds[i] = run;
      run += v;
    }
    *total = run;
  }
  __syncthreads();
  uint32_t run = lds[threadIdx.x];
  for (size_t i = lo; i < hi; i++) {
    uint32_t v = tile_sums[i];
    tile_sums[i] = run;
    run += v;
  }
}
__global__ void scan_apply_kernel(uint32_t* __restrict__ data, size_t n, const uint32_t* __restrict__ tile_offsets) {
  __shared__ uint32_t lds[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
  uint32_t v[8];
  uint32_t s = 0;
  for (int k = 0; k < 8; k++) {
    v[k] = base + k < n ? data[base + k] : 0u;
    s += v[k];
  }
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
    uint32_t t = (int)threadIdx.x >= off ? lds[threadIdx.x - off] : 0u;
    __syncthreads();
    lds[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = lds[threadIdx.x] - s + tile_offsets[blockIdx.x];
  for (int k = 0; k < 8; k++) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
}

// ------------------------------------------------------------------ 4: segmented accumulate
__device__ __forceinline__ const MsmJobDev& job_of_key(const MsmPlanDev& plan, uint32_t key) {
  int j = 0;
  while (j + 1 < plan.num_jobs && plan.job[j + 1].key_base <= key) j++;
  return plan.job[j];
}

// level 0: affine bases, mixed adds
__global__ __launch_bounds__(128) void msm_accumulate0_kernel(MsmPlanDev plan, const uint32_t* __restrict__ total_ptr,
                                                              const uint32_t* __restrict__ sorted_key,
                                                              const uint32_t* __restrict__ sorted_idx, uint32_t K,
                                                              G1Xyzz* __restrict__ buckets,
                                                              uint32_t* __restrict__ cont_key,
                                                              G1Xyzz* __restrict__ cont_pt, size_t nchunks,
                                                              uint32_t* __restrict__ cont_count) {
  const size_t total = *total_ptr;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nchunks; t += (size_t)gridDim.x * blockDim.x) {
    size_t p0 = t * K, p1 = p0 + K < total ? p0 + K : total;
    uint32_t ck = SENTINEL;
    if (p0 < total) {
      uint32_t cur = sorted_key[p0];
      bool cont = p0 > 0 && sorted_key[p0 - 1] == cur;
      const G1Affine* bases = job_of_key(plan, cur).bases;
      G1Xyzz acc = G1Xyzz::identity();
      for (size_t p = p0; p < p1; p++) {
        uint32_t k = sorted_key[p];
        if (k != cur) {
          if (cont) {
            ck = cur;
            cont_pt[t] = acc;
          } else {
            buckets[cur] = acc;
          }
          cont = false;
          acc = G1Xyzz::identity();
          cur = k;
          bases = job_of_key(plan, cur).bases;
        }
        acc = add_mixed(acc, bases[sorted_idx[p]]);
      }
      if (cont) {
        ck = cur;
        cont_pt[t] = acc;
      } else {
        buckets[cur] = acc;
      }
    }
    cont_key[t] = ck;
    if (ck != SENTINEL) atomicAdd(cont_count, 1u);  // per-wave combined by the compiler
  }
}

// level >= 1: (key, XYZZ) entries with sentinels; heads are ADDED into the bucket array
__global__ __launch_bounds__(128) void msm_accumulate_n_kernel(const uint32_t* __restrict__ in_key,
                                                               const G1Xyzz* __restrict__ in_pt, size_t n_in,
                                                               uint32_t K, G1Xyzz* __restrict__ buckets,
                                                               uint32_t* __restrict__ out_key,
                                                               G1Xyzz* __restrict__ out_pt, size_t nchunks,
                                                               const uint32_t* __restrict__ in_count,
                                                               uint32_t* __restrict__ out_count) {
  if (*in_count == 0) return;  // nothing continued into this level (out_count stays 0 for the next one)
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nchunks; t += (size_t)gridDim.x * blockDim.x) {
    size_t p0 = t * K, p1 = p0 + K < n_in ? p0 + K : n_in;
    uint32_t ck = SENTINEL;
    uint32_t cur = SENTINEL;
    bool cont = false;
    G1Xyzz acc = G1Xyzz::identity();
    for (size_t p = p0; p < p1; p++) {
      uint32_t k = in_key[p];
      if (k == SENTINEL) continue;
      if (k != cur) {
        if (cur != SENTINEL) {
          if (cont) {
            ck = cur;
            out_pt[t] = acc;
          } else {
            buckets[cur] = add(buckets[cur], acc);
          }
        }
        // a run continues from the previous chunk only if it starts this chunk
        cont = (p == p0) && p0 > 0 && in_key[p0 - 1] == k;
        acc = G1Xyzz::identity();
        cur = k;
      }
      acc = add(acc, in_pt[p]);
    }
    if (cur != SENTINEL) {
      if (cont) {
        ck = cur;
        out_pt[t] = acc;
      } else {
        buckets[cur] = add(buckets[cur], acc);
      }
    }
    out_key[t] = ck;
    if (ck != SENTINEL) atomicAdd(out_count, 1u);
  }
}

// ------------------------------------------------------------------ 5: bucket reduce
__device__ __forceinline__ G1Xyzz mul_small(const G1Xyzz& p, uint32_t k) {
  G1Xyzz acc = G1Xyzz::identity();
  for (int b = 31 - __clz(k | 1u); b >= 0; b--) {
    acc = dbl(acc);
    if ((k >> b) & 1u) acc = add(acc, p);
  }
  return k ? acc : G1Xyzz::identity();
}

__global__ __launch_bounds__(64) void msm_segment_reduce_kernel(MsmPlanDev plan, const G1Xyzz* __restrict__ buckets,
                                                                G1Xyzz* __restrict__ seg_out, size_t total_segs) {
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < total_segs; s += (size_t)gridDim.x * blockDim.x) {
    int j = 0;
    while (j + 1 < plan.num_jobs && plan.job[j + 1].seg_base <= s) j++;
    const MsmJobDev& jb = plan.job[j];
    uint32_t local = (uint32_t)(s - jb.seg_base);
    uint32_t w = local / jb.seg_per_win, seg = local % jb.seg_per_win;
    uint32_t d0 = seg * jb.seg_size;
    const G1Xyzz* b = buckets + jb.key_base + ((size_t)w << jb.c) + d0;
    G1Xyzz run = G1Xyzz::identity(), acc = G1Xyzz::identity();
    for (int d = (int)jb.seg_size - 1; d >= 0; d--) {
      acc = add(acc, run);
      run = add(run, b[d]);
    }
    if (d0) acc = add(acc, mul_small(run, d0));
    seg_out[s] = acc;
  }
}

// one workgroup per (job, window): sum of its segment partials
__global__ __launch_bounds__(256) void msm_window_sum_kernel(MsmPlanDev plan, const G1Xyzz* __restrict__ seg_out,
                                                             G1Xyzz* __restrict__ win_out) {
  __shared__ G1Xyzz lds[256];
  int j = 0;
  while (j + 1 < plan.num_jobs && plan.job[j + 1].win_base <= blockIdx.x) j++;
  const MsmJobDev& jb = plan.job[j];
  uint32_t w = blockIdx.x - jb.win_base;
  const G1Xyzz* src = seg_out + jb.seg_base + (size_t)w * jb.seg_per_win;
  G1Xyzz acc = G1Xyzz::identity();
  for (uint32_t i = threadIdx.x; i < jb.seg_per_win; i += blockDim.x) acc = add(acc, src[i]);
  lds[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off && (uint32_t)(threadIdx.x + off) < jb.seg_per_win)
      lds[threadIdx.x] = add(lds[threadIdx.x], lds[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) win_out[blockIdx.x] = lds[0];
}

// ------------------------------------------------------------------ host driver
static uint32_t pick_window(size_t n, uint32_t bits) {
  uint32_t lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = (int)lg - 3;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  if ((uint32_t)c > bits) c = (int)bits;
  return (uint32_t)c;
}

static inline host::G1Xyzz to_host(const G1Xyzz& p) {
  host::G1Xyzz r;
  memcpy(&r, &p, sizeof(r));
  return r;
}

void msm_batch(Ctx& c, const MsmJob* jobs, size_t num_jobs, G1Affine* out_host) {
  for (size_t base = 0; base < num_jobs; base += MSM_MAX_JOBS) {
    size_t nj = std::min(num_jobs - base, (size_t)MSM_MAX_JOBS);
    MsmPlanDev plan;
    plan.num_jobs = (int)nj;
    uint32_t key = 0, seg = 0, win = 0, cnt = 0;
    size_t max_entries = 0, max_n = 0;
    for (size_t j = 0; j < nj; j++) {
      const MsmJob& in = jobs[base + j];
      LH_REQUIRE(in.n < ((size_t)1 << 31), LH_ERR_ARG, "msm: too many points");
      MsmJobDev& jd = plan.job[j];
      uint32_t bits = in.scalars_u32 ? 32 : 254;
      jd.scalars = in.scalars;
      jd.bases = in.bases;
      jd.is_u32 = in.scalars_u32 ? 1 : 0;
      jd.n = (uint32_t)in.n;
      jd.c = pick_window(in.n ? in.n : 1, bits);
      jd.W = (bits + jd.c - 1) / jd.c;
      jd.key_base = key;
      jd.cnt_base = cnt;
      jd.rep_log = in.scalars_u32 ? 3 : 0;
      jd.seg_size = std::min<uint32_t>(16u, 1u << jd.c);
      jd.seg_per_win = (1u << jd.c) / jd.seg_size;
      jd.seg_base = seg;
      jd.win_base = win;
      key += jd.W << jd.c;
      cnt += (jd.W << jd.c) << jd.rep_log;
      seg += jd.W * jd.seg_per_win;
      win += jd.W;
      max_entries += (size_t)jd.n * jd.W;
      max_n = std::max(max_n, in.n);
    }
    const size_t nbuckets = key, nsegs = seg, nwins = win, ncounts = cnt;
    std::vector<G1Xyzz> wins(nwins);
    if (max_entries == 0) {
      for (size_t j = 0; j < nj; j++) memset(&out_host[base + j], 0, sizeof(G1Affine));
      continue;
    }
    LH_REQUIRE(max_entries < ((size_t)1 << 32), LH_ERR_ARG, "msm: batch too large for 32-bit entry indices");
    {
      ArenaScope scope(c.arena);
      uint32_t* counts = c.arena.alloc_n<uint32_t>(ncounts + 1);
      size_t ntiles = (ncounts + 1 + SCAN_TILE - 1) / SCAN_TILE;
      uint32_t* tile_sums = c.arena.alloc_n<uint32_t>(ntiles);
      uint32_t* total = c.arena.alloc_n<uint32_t>(1);
      uint32_t* skey = c.arena.alloc_n<uint32_t>(max_entries);
      uint32_t* sidx = c.arena.alloc_n<uint32_t>(max_entries);
      G1Xyzz* buckets = c.arena.alloc_n<G1Xyzz>(nbuckets);
      G1Xyzz* seg_out = c.arena.alloc_n<G1Xyzz>(nsegs);
      G1Xyzz* win_out = c.arena.alloc_n<G1Xyzz>(nwins);

      uint32_t* lvl_cnt = c.arena.alloc_n<uint32_t>(64);
      LH_HIP(hipMemsetAsync(lvl_cnt, 0, 64 * sizeof(uint32_t), c.stream));
      LH_HIP(hipMemsetAsync(counts, 0, (ncounts + 1) * sizeof(uint32_t), c.stream));
      LH_HIP(hipMemsetAsync(buckets, 0, nbuckets * sizeof(G1Xyzz), c.stream));
      dim3 g((unsigned)std::min<size_t>((max_n + 255) / 256, 2048), (unsigned)nj);
      double total_pts = 0, full_pts = 0;
      for (size_t j = 0; j < nj; j++) total_pts += plan.job[j].n, full_pts += plan.job[j].is_u32 ? 0 : plan.job[j].n;
      {
        ProfScope ps(c, "msm_digits_count", 32.0 * full_pts + 4.0 * (total_pts - full_pts), full_pts, total_pts);
        hipLaunchKernelGGL(msm_digits_kernel<false>, g, dim3(256), 0, c.stream, plan, counts, nullptr, nullptr);
      }
      {
        ProfScope ps(c, "msm_scan", 8.0 * (ncounts + 1), 0, (double)ncounts);
      hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles), dim3(256), 0, c.stream, counts, ncounts + 1,
                         tile_sums);
      hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(256), 0, c.stream, tile_sums, ntiles, total);
      hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)ntiles), dim3(256), 0, c.stream, counts, ncounts + 1,
                         tile_sums);
      }
      {
        ProfScope ps(c, "msm_digits_scatter", 32.0 * full_pts + 4.0 * (total_pts - full_pts) + 8.0 * max_entries, full_pts,
                     total_pts);
        hipLaunchKernelGGL(msm_digits_kernel<true>, g, dim3(256), 0, c.stream, plan, counts, skey, sidx);
      }

      // segmented accumulate, level 0 then K-fold shrinking continuation lists
      uint32_t K = max_entries > ((size_t)1 << 22) ? 16 : max_entries > ((size_t)1 << 18) ? 8 : 4;
      size_t nchunks = (max_entries + K - 1) / K;
      uint32_t* ckey = c.arena.alloc_n<uint32_t>(nchunks);
      G1Xyzz* cpt = c.arena.alloc_n<G1Xyzz>(nchunks);
      uint32_t h_total = 0;
      if (c.prof) {
        LH_HIP(hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, c.stream));
        c.sync();
      }
      {
        // MSM algorithmic bytes: 96 B per point (32 B scalar + 64 B base); a mixed add is 10 Fq muls
        ProfScope ps(c, "msm_accumulate0", 72.0 * h_total, 10.0 * h_total, (double)h_total);
      hipLaunchKernelGGL(msm_accumulate0_kernel, dim3((unsigned)std::min<size_t>((nchunks + 127) / 128, 1 << 16)),
                         dim3(128), 0, c.stream, plan, total, skey, sidx, K, buckets, ckey, cpt, nchunks, lvl_cnt);
      }
      size_t n_in = nchunks;
      const uint32_t K2 = 8;
      {
        ProfScope ps(c, "msm_accumulate_levels", 0, 0, (double)nchunks);
      int lvl = 0;
      while (true) {
        size_t nc = (n_in + K2 - 1) / K2;
        uint32_t* okey = c.arena.alloc_n<uint32_t>(nc);
        G1Xyzz* opt = c.arena.alloc_n<G1Xyzz>(nc);
        hipLaunchKernelGGL(msm_accumulate_n_kernel, dim3((unsigned)std::min<size_t>((nc + 127) / 128, 1 << 16)),
                           dim3(128), 0, c.stream, ckey, cpt, n_in, K2, buckets, okey, opt, nc, lvl_cnt + lvl,
                           lvl_cnt + lvl + 1);
        lvl++;
        if (n_in <= K2) break;  // a single chunk: no continuation can remain
        ckey = okey;
        cpt = opt;
        n_in = nc;
      }
      }
      {
        ProfScope ps(c, "msm_bucket_reduce", 128.0 * nbuckets, 14.0 * 2.2 * nbuckets, (double)nbuckets);
      hipLaunchKernelGGL(msm_segment_reduce_kernel, dim3((unsigned)std::min<size_t>((nsegs + 63) / 64, 1 << 16)),
                         dim3(64), 0, c.stream, plan, buckets, seg_out, nsegs);
      hipLaunchKernelGGL(msm_window_sum_kernel, dim3((unsigned)nwins), dim3(256), 0, c.stream, plan, seg_out, win_out);
      }
      LH_HIP(hipMemcpyAsync(wins.data(), win_out, nwins * sizeof(G1Xyzz), hipMemcpyDeviceToHost, c.stream));
      c.sync();
    }
    // 6: host combine  sum_w 2^(c*w) * win[w]  and normalise
    for (size_t j = 0; j < nj; j++) {
      const MsmJobDev& jd = plan.job[j];
      host::G1Xyzz acc = host::G1Xyzz::identity();
      for (int w = (int)jd.W - 1; w >= 0; w--) {
        for (uint32_t k = 0; k < jd.c; k++) acc = host::g1_dbl(acc);
        acc = host::g1_add(acc, to_host(wins[jd.win_base + w]));
      }
      host::G1Affine a = host::g1_to_affine(acc);
      memcpy(&out_host[base + j], &a, sizeof(G1Affine));
    }
  }
}

// ------------------------------------------------------------------ fixed-base multiples of G (SRS setup)
// reference kzg.rs:196-207 (window_table + fixed_base_msm + batch_normalize).  8-bit windows:
// table[w][d-1] = d * 2^(8w) * G, 32 x 255 affine points (510 KiB, L2-resident).
__global__ __launch_bounds__(128) void fixed_base_kernel(const Fr* __restrict__ scalars, size_t n,
                                                         const G1Affine* __restrict__ table,
                                                         G1Affine* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr s = from_mont(scalars[i]);
    G1Xyzz acc = G1Xyzz::identity();
    for (int k = 0; k < 8; k++) {  // rolled: limbs rotate through s.l[0] so that indexing stays static
      uint32_t limb = s.l[0];
#pragma unroll
      for (int q = 0; q < 7; q++) s.l[q] = s.l[q + 1];
      for (int b = 0; b < 4; b++) {
        uint32_t d = (limb >> (8 * b)) & 0xffu;
        if (d) acc = add_mixed(acc, table[(k * 4 + b) * 255 + (d - 1)]);
      }
    }
    G1Affine r;
    if (acc.is_identity()) {
      r.x = Fq::zero();
      r.y = Fq::zero();
    } else {
      Fq i2 = inv(mul(acc.zz, acc.zzz));
      r.x = mul(acc.x, mul(i2, acc.zzz));
      r.y = mul(acc.y, mul(i2, acc.zz));
    }
    out[i] = r;
  }
}

void k_fixed_base_mul_g(Ctx& c, const Fr* scalars, size_t n, G1Affine* out) {
  if (!n) return;
  // host-built window table of the generator (1, 2)
  std::vector<host::G1Affine> tab(32 * 255);
  host::G1Affine g{host::Fq::from_u64(1), host::Fq::from_u64(2)};
  host::G1Xyzz off = host::g1_from_affine(g);
  for (int w = 0; w < 32; w++) {
    host::G1Xyzz acc = off;
    for (int d = 0; d < 255; d++) {
      tab[w * 255 + d] = host::g1_to_affine(acc);
      acc = host::g1_add(acc, off);
    }
    off = acc;  // 256 * off
  }
  ArenaScope scope(c.arena);
  G1Affine* d_tab = c.arena.alloc_n<G1Affine>(tab.size());
  LH_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(G1Affine), hipMemcpyHostToDevice, c.stream));
  hipLaunchKernelGGL(fixed_base_kernel, dim3((unsigned)std::min<size_t>((n + 127) / 128, 1 << 16)), dim3(128), 0,
                     c.stream, scalars, n, d_tab, out);
  c.sync();  // `tab` is pageable host memory: keep it alive until the copy has run
}

}  // namespace lh
