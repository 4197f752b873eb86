// Batched variable-base MSM over BN254 G1 for gfx950: sort-by-bucket Pippenger.
//
// Replaces variable_base_msm (reference plonkish_backend/src/util/arithmetic/msm.rs:84-181), whose
// chunk-per-thread Pippenger has no device analogue; only the affine sum is observable
// (SURVEY.md §3.4), so window size / bucket scheme are chosen for the GPU:
//
//  1. digits     every scalar -> canonical -> c-bit digits -> (bucket key, point index) pairs at fixed
//                positions (coalesced, no atomics); digit d > 0 goes to bucket d - 1 of its (job, window) slab, a
//                zero digit keeps a valid key and carries the SKIP index (nothing is added for it)
//  2. sort       radix sort of the pairs by key (order inside a bucket is free).  The pairs leave step 1 grouped by
//                (job, window) and a slab's bucket range starts at a multiple of 2^(digit bits), so a big slab only
//                needs its <= 16 digit bits sorted (2 radix passes instead of 3 over the 21-23-bit global key);
//                the slabs of small jobs are sorted together by the full key
//  3. accumulate load-balanced segmented sum: every thread owns ~K CONSECUTIVE sorted entries whatever
//                the bucket sizes are (Lasso's read_ts / final_cts / dim columns are heavily skewed - a
//                thread-per-bucket scheme would serialise on the hot buckets).  Chunk boundaries snap to
//                the next bucket boundary when one is within K entries, so ordinary buckets are never
//                split; only a bucket longer than K is cut, its first piece is stored to the bucket
//                array and the following pieces go to a continuation list that is reduced the same
//                way (K-fold shrink per level).
//  5. reduce     per (job, window): sum_d d*B[d] by 16-bucket segments (running sums + d0*T), then a
//                workgroup tree over the segments
//  6. combine    the W window sums go to the host: Horner with c doublings per window and ONE field
//                inversion to affine (a 254-doubling dependent chain is ~60 us on a CPU core and
//                milliseconds on a single GPU lane).
// Several MSMs (a batch_commit's polys, the n quotient commitments of one opening) run as ONE batch:
// the key space is (job, window, digit), so the latency-bound tails are paid once per batch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <exception>
#include <functional>
#include <thread>
#include <vector>
#include <mutex>
#include "dev.hpp"
#include "ff_host.hpp"

namespace lh {

constexpr int MSM_MAX_JOBS = 48;
constexpr uint32_t SENTINEL = 0xffffffffu;   // continuation lists: "no entry"
constexpr uint32_t SKIP_IDX = 0x7fffffffu;   // sorted entries: a zero digit (nothing to add)

struct MsmJobDev {
  const void* scalars;
  const G1Affine* bases;
  uint32_t is_u32;
  uint32_t n;
  uint32_t c, W;        // window bits, number of windows
  uint32_t key_base;    // first bucket key of this job
  uint32_t entry_base;  // first (key, index) pair of this job: pair of (window w, element i) at w*n + i
  uint32_t win_stride;  // bucket slots per window (multiple of seg_size)
  uint32_t is_signed;   // Fr jobs use signed digits: buckets 1..2^(c-1), the sign travels in bit 31 of the index
  uint32_t seg_base;    // first reduce-segment of this job
  uint32_t seg_per_win; // segments per window
  uint32_t seg_size;    // buckets per segment
  uint32_t win_base;    // first window-sum slot of this job
  uint32_t presorted;   // the sorted entries of its single slab come from MsmJob::sorted_scalars / sorted_index
  uint32_t pack_shift;  // != 0: two columns packed into one (MsmJob::pack_shift): one window whose bucket index IS the
                        // packed value, reduced twice (red_W = 2 "windows" over the same buckets: low part, high part)
  uint32_t red_W;       // windows of the reduction (= W except for packed jobs)
  uint32_t nsplit;      // workgroups (shares) per window in the window-sum kernel
  uint32_t share_base;  // first share (workgroup / output slot) of this job
  uint32_t two_level;   // bucket reduction in two levels (msm_group_reduce_kernel): the segment kernel leaves A_s and T_s,
                        // groups of MSM_GROUP segments are weighed together - one small multiple per group instead of per segment
  uint32_t grp_base;    // first group slot of this job (two_level)
  uint32_t sum_per_win; // partials per window the window-sum kernel adds: groups (two_level) or segments
  uint32_t merged;      // `bases` is a window table (MsmJob::win_table: entry w * n + i = 2^(c w) * base i): all W windows
                        // fill ONE bucket set (red_W = 1), the entry's index carries the window
};
constexpr uint32_t KEY_BLOCK_BITS = 10;  // every job's key range starts at a multiple of 2^KEY_BLOCK_BITS
struct MsmPlanDev {
  int num_jobs;
  const uint8_t* job_of_block;  // device: job index of the key block key >> KEY_BLOCK_BITS
  MsmJobDev job[MSM_MAX_JOBS];
};

// ------------------------------------------------------------------ 1: digits
// OR of the canonical limbs of every scalar of a job: gives the number of significant bits, so that an
// Fr column holding small values (Lasso's output column) only emits the windows it needs.
__global__ void msm_or_limbs_kernel(MsmPlanDev plan, uint32_t* __restrict__ or_out /* [jobs][8] */) {
  const MsmJobDev& jb = plan.job[blockIdx.y];
  uint32_t acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = 0;
  if (jb.is_u32) {  // a 32-bit column of 16-bit chunk indices needs one window, not two
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += (size_t)gridDim.x * blockDim.x)
      acc[0] |= ((const uint32_t*)jb.scalars)[i];
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += (size_t)gridDim.x * blockDim.x) {
      Fr s = from_mont(((const Fr*)jb.scalars)[i]);
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] |= s.l[k];
    }
  }
  __shared__ uint32_t lds_or[8];
  if (threadIdx.x < 8) lds_or[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint32_t v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicOr(&lds_or[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 8 && lds_or[threadIdx.x]) atomicOr(&or_out[blockIdx.y * 8 + threadIdx.x], lds_or[threadIdx.x]);
}

// a column that arrives sorted by value (single unsigned window: digit = value): the sorted entry stream directly
__global__ void msm_presorted_kernel(uint32_t key_base, const uint32_t* __restrict__ sorted_scalars,
                                     const uint32_t* __restrict__ sorted_index, size_t n, uint32_t* __restrict__ skey,
                                     uint32_t* __restrict__ sidx) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const uint32_t d = sorted_scalars[p];
    skey[p] = key_base + (d ? d - 1u : 0u);
    sidx[p] = d ? sorted_index[p] : SKIP_IDX;
  }
}

__global__ void msm_emit_kernel(MsmPlanDev plan, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const MsmJobDev& jb = plan.job[blockIdx.y];
  if (jb.presorted) return;
  const uint32_t c = jb.c, mask = (1u << c) - 1u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t limb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) limb[k] = 0;
    if (jb.is_u32) {
      limb[0] = ((const uint32_t*)jb.scalars)[i];
    } else {
      Fr s = from_mont(((const Fr*)jb.scalars)[i]);
#pragma unroll
      for (int k = 0; k < 8; k++) limb[k] = s.l[k];
    }
    const int nlimbs = jb.is_u32 ? 1 : 8;
    uint64_t buf = 0;
    int have = 0;
    uint32_t w = 0;
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1);
    auto emit = [&](uint32_t raw) {
      uint32_t d = raw + carry, neg = 0;
      if (jb.is_signed) {
        carry = d > half ? 1u : 0u;
        if (carry) {
          d = (1u << c) - d;
          neg = 0x80000000u;
        }
      }
      size_t e = (size_t)jb.entry_base + (size_t)w * jb.n + i;
      // bucket index = digit - 1 (packed jobs: = digit, so that a reduce segment never straddles a change of the high part)
      keys[e] = jb.key_base + (jb.merged ? 0u : w * jb.win_stride) + (jb.pack_shift ? d : (d ? d - 1u : 0u));
      vals[e] = d ? (((uint32_t)i + (jb.merged ? w * jb.n : 0u)) | neg) : SKIP_IDX;
    };
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k < nlimbs) {
        buf |= (uint64_t)limb[k] << have;
        have += 32;
        while (have >= (int)c && w < jb.W) {
          emit((uint32_t)buf & mask);
          buf >>= c;
          have -= c;
          w++;
        }
      }
    }
    while (w < jb.W) {  // top, partial window(s): remaining bits, then the final carry
      emit((uint32_t)buf & mask);
      buf >>= c;
      w++;
    }
  }
}


// ------------------------------------------------------------------ 4: segmented accumulate
// O(1): this runs at every bucket boundary of the accumulate loop, and a wave takes the branch whenever ANY of
// its lanes crosses a boundary (nearly every iteration), so a linear scan of the job list costs a multiplication.
__device__ __forceinline__ const MsmJobDev& job_of_key(const MsmPlanDev& plan, uint32_t key) {
  return plan.job[plan.job_of_block[key >> KEY_BLOCK_BITS]];
}

// level 0: affine bases, mixed adds
// The sorted (key, index) stream of a thread's chunk is fetched 16 entries at a time with 16-byte loads and parked
// in a lane-private LDS column: one entry per iteration straight from global memory would touch 64 different
// cache lines per wave instruction and keep every line alive for 32 iterations (measured: 2.7x the algorithmic
// fetch traffic).  (Requesting the next base point one iteration ahead was tried and lost 5 %: more registers.)
constexpr int ACC_GROUP = 16;
// (137 registers, three waves per SIMD since the lazy forms; forcing four - 128 registers, 44 bytes of spills - measured the
// same 25.07 ms per 2^24 AND proof, tools/ab_acc0_lazy.sh: the allocator's choice stays)
__global__ __launch_bounds__(128)
void msm_accumulate0_kernel(MsmPlanDev plan, size_t total,
                                                              const uint32_t* __restrict__ sorted_key,
                                                              const uint32_t* __restrict__ sorted_idx, uint32_t K,
                                                              G1Xyzz* __restrict__ buckets,
                                                              uint32_t* __restrict__ cont_key,
                                                              G1Xyzz* __restrict__ cont_pt, size_t nchunks,
                                                              uint32_t* __restrict__ cont_count) {
  __shared__ uint32_t lds_key[ACC_GROUP * 128], lds_idx[ACC_GROUP * 128];
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nchunks; t += (size_t)gridDim.x * blockDim.x) {
    const size_t p0 = t * K < total ? t * K : total, p1 = p0 + K < total ? p0 + K : total;
    uint32_t ck = SENTINEL;
    if (p0 < p1) {
      uint32_t cur = sorted_key[p0];
      bool cont = p0 > 0 && sorted_key[p0 - 1] == cur;
      const G1Affine* bases = job_of_key(plan, cur).bases;
      G1Xyzz acc = G1Xyzz::identity();
      for (size_t g0 = p0; g0 < p1; g0 += ACC_GROUP) {
        // K is a multiple of 4 and the arrays are padded past the last chunk: whole 16-byte loads stay in bounds
        const uint32_t cnt = (uint32_t)(p1 - g0 < (size_t)ACC_GROUP ? p1 - g0 : (size_t)ACC_GROUP);
        for (uint32_t q = 0; q < cnt; q += 4) {
          const uint4 kv = *(const uint4*)(sorted_key + g0 + q), iv4 = *(const uint4*)(sorted_idx + g0 + q);
          lds_key[(q + 0) * 128 + threadIdx.x] = kv.x, lds_key[(q + 1) * 128 + threadIdx.x] = kv.y;
          lds_key[(q + 2) * 128 + threadIdx.x] = kv.z, lds_key[(q + 3) * 128 + threadIdx.x] = kv.w;
          lds_idx[(q + 0) * 128 + threadIdx.x] = iv4.x, lds_idx[(q + 1) * 128 + threadIdx.x] = iv4.y;
          lds_idx[(q + 2) * 128 + threadIdx.x] = iv4.z, lds_idx[(q + 3) * 128 + threadIdx.x] = iv4.w;
        }
        for (uint32_t j = 0; j < cnt; j++) {
          const uint32_t k = lds_key[j * 128 + threadIdx.x];
          if (k != cur) {
            if (cont) {
              ck = cur;
              cont_pt[t] = canon_xyzz(acc);
            } else {
              buckets[cur] = canon_xyzz(acc);
            }
            cont = false;
            acc = G1Xyzz::identity();
            cur = k;
            bases = job_of_key(plan, cur).bases;
          }
          const uint32_t iv = lds_idx[j * 128 + threadIdx.x];
          if (iv != SKIP_IDX) acc = add_mixed_lazy(acc, bases[iv & 0x7fffffffu], (iv >> 31) != 0);  // (lazy coordinates until the flush)
        }
      }
      if (cont) {
        ck = cur;
        cont_pt[t] = canon_xyzz(acc);
      } else {
        buckets[cur] = canon_xyzz(acc);
      }
    }
    cont_key[t] = ck;
    if (ck != SENTINEL) atomicAdd(cont_count, 1u);  // per-wave combined by the compiler
  }
}

// level >= 1: (key, XYZZ) entries with sentinels.  One thread per ENTRY: the thread sitting on the first
// entry of a segment sums it (a segment = a run of equal keys, cut at every multiple of G so that no
// thread adds more than G points).  The first segment of a run is ADDED into the bucket array, a
// segment that starts at a forced cut continues a run and goes to the next level's list (one slot per
// G entries).  With uniform digits a run is 1-2 entries, so this level finishes the job in one
// short step; long runs (skewed columns) shrink G-fold per level.
__global__ __launch_bounds__(128) void msm_accumulate_n_kernel(const uint32_t* __restrict__ in_key,
                                                               const G1Xyzz* __restrict__ in_pt, size_t n_in,
                                                               uint32_t G, G1Xyzz* __restrict__ buckets,
                                                               uint32_t* __restrict__ out_key,
                                                               G1Xyzz* __restrict__ out_pt,
                                                               const uint32_t* __restrict__ in_count,
                                                               uint32_t* __restrict__ out_count) {
  if (*in_count == 0) return;  // nothing continued into this level (out_count stays 0 for the next one)
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_in; p += (size_t)gridDim.x * blockDim.x) {
    const uint32_t k = in_key[p];
    const bool valid = k != SENTINEL;
    const bool prev_same = valid && p > 0 && in_key[p - 1] == k;
    const bool slot_owner = (p % G) == 0;
    const bool forced = prev_same && slot_owner;
    uint32_t ok = SENTINEL;
    if (valid && (!prev_same || forced)) {
      G1Xyzz acc = in_pt[p];
      for (size_t q = p + 1; q < n_in && (q % G) != 0 && in_key[q] == k; q++) acc = add(acc, in_pt[q]);
      if (forced) {
        ok = k;
        out_pt[p / G] = acc;
        atomicAdd(out_count, 1u);
      } else {
        buckets[k] = add(buckets[k], acc);
      }
    }
    if (slot_owner) out_key[p / G] = ok;
  }
}

// The same level for short lists (ec.cuh, quad-cooperative arithmetic): one QUAD of lanes per entry, 9 us per
// dependent addition instead of 20.  Used when the list is far too short to fill the chip anyway.
__global__ __launch_bounds__(128) void msm_accumulate_n_quad_kernel(const uint32_t* __restrict__ in_key,
                                                                    const G1Xyzz* __restrict__ in_pt, size_t n_in,
                                                                    uint32_t G, G1Xyzz* __restrict__ buckets,
                                                                    uint32_t* __restrict__ out_key,
                                                                    G1Xyzz* __restrict__ out_pt,
                                                                    const uint32_t* __restrict__ in_count,
                                                                    uint32_t* __restrict__ out_count) {
  if (*in_count == 0) return;
  const bool lead = (threadIdx.x & 3u) == 0;
  const size_t quads = ((size_t)gridDim.x * blockDim.x) >> 2;
  for (size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; p < n_in; p += quads) {
    const uint32_t k = in_key[p];
    const bool valid = k != SENTINEL;
    const bool prev_same = valid && p > 0 && in_key[p - 1] == k;
    const bool slot_owner = (p % G) == 0;
    const bool forced = prev_same && slot_owner;
    uint32_t ok = SENTINEL;
    if (valid && (!prev_same || forced)) {
      G1Xyzz acc = in_pt[p];
      for (size_t q = p + 1; q < n_in && (q % G) != 0 && in_key[q] == k; q++) acc = add_quad(acc, in_pt[q]);
      if (forced) {
        ok = k;
        if (lead) {
          out_pt[p / G] = acc;
          atomicAdd(out_count, 1u);
        }
      } else {
        const G1Xyzz sum = add_quad(buckets[k], acc);
        if (lead) buckets[k] = sum;
      }
    }
    if (slot_owner && lead) out_key[p / G] = ok;
  }
}

// The same level as a TREE for the short lists at the end of the chain (quad-cooperative arithmetic): a workgroup owns a
// tile of T consecutive slots, one quad of lanes per slot, and sums every run of equal keys inside the tile by doubling
// (slot p adds slot p + d while that slot carries the same key: log2 T dependent additions whatever the run lengths,
// where the linear levels need G - 1 additions to shrink a run G-fold).  The piece of a run that begins in the tile is
// added into its bucket; a run that continues from the previous tile leaves its piece in the next level's list, one slot
// per tile - a list shrinks T-fold per launch instead of 4-fold.
template <int T>
__global__ __launch_bounds__(4 * T) void msm_accumulate_tree_quad_kernel(const uint32_t* __restrict__ in_key,
                                                                         const G1Xyzz* __restrict__ in_pt, size_t n_in,
                                                                         G1Xyzz* __restrict__ buckets,
                                                                         uint32_t* __restrict__ out_key,
                                                                         G1Xyzz* __restrict__ out_pt,
                                                                         const uint32_t* __restrict__ in_count,
                                                                         uint32_t* __restrict__ out_count) {
  if (*in_count == 0) return;
  __shared__ G1Xyzz pts[T];
  __shared__ uint32_t keys[T];
  const int slot = (int)(threadIdx.x >> 2);
  const bool lead = (threadIdx.x & 3u) == 0;
  const size_t ntiles = (n_in + T - 1) / T;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t p = tile * T + (size_t)slot;
    const uint32_t k = p < n_in ? in_key[p] : SENTINEL;
    const bool valid = k != SENTINEL;
    const bool prev_same = valid && p > 0 && in_key[p - 1] == k;
    G1Xyzz acc = valid ? in_pt[p] : G1Xyzz::identity();
    __syncthreads();  // (the previous tile's reads of the arrays are over)
    if (lead) keys[slot] = k, pts[slot] = acc;
    __syncthreads();
    // One key over the whole tile (sorted keys: first == last) - the long runs of a derived job's list, of a skewed column:
    // a halving tree with sequential addressing (slot p < h adds slot p + h) needs 2 + 1 + 1 + ... wave-additions for the
    // T slots, because the active slots stay packed in the first waves; the doubling form below, which copes with any run
    // boundaries, keeps all T slots adding in each of its log2 T steps.
    const bool uniform = keys[0] != SENTINEL && keys[0] == keys[T - 1];  // (workgroup-uniform: the same LDS words for everyone)
    if (uniform) {
      for (int h = T / 2; h >= 1; h >>= 1) {
        if (slot < h) {
          acc = add_quad(acc, pts[slot + h]);
          if (lead) pts[slot] = acc;
        }
        __syncthreads();
      }
    } else {
      for (int d = 1; d < T; d <<= 1) {
        const bool take = valid && slot + d < T && keys[slot + d] == k;
        G1Xyzz other = G1Xyzz::identity();
        if (take) other = pts[slot + d];
        __syncthreads();
        if (take) {
          acc = add_quad(acc, other);
          if (lead) pts[slot] = acc;
        }
        __syncthreads();
      }
    }
    if (valid && !prev_same) {  // the run begins here: its sum inside the tile joins the bucket
      const G1Xyzz sum = add_quad(buckets[k], acc);
      if (lead) buckets[k] = sum;
    }
    if (slot == 0 && lead) {
      const bool cont = valid && prev_same;  // the tile opens inside a run that began earlier
      out_key[tile] = cont ? k : SENTINEL;
      if (cont) {
        out_pt[tile] = acc;
        atomicAdd(out_count, 1u);
      }
    }
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

// quad-cooperative small multiple (the reductions of batches with few segments, the group kernel)
__device__ __forceinline__ G1Xyzz mul_small_quad(const G1Xyzz& p, uint32_t k) {
  G1Xyzz acc = G1Xyzz::identity();
  for (int b = 31 - __clz(k | 1u); b >= 0; b--) {
    acc = dbl_quad(acc);
    if ((k >> b) & 1u) acc = add_quad(acc, p);
  }
  return k ? acc : G1Xyzz::identity();
}

// Two-level reduction (jobs with >= MSM_GROUP^2 segments per window, throughput-bound batches).  sum_b (b + 1) B_b over a
// window, segment s = S consecutive buckets from d0 = s S:  the running sums of the segment give A_s = sum_d d B_{d0 + d}
// (zero-based) and T_s = sum_d B_{d0 + d}, and the window's sum is  sum_s [A_s + (d0 + 1) T_s].  One small-scalar
// multiplication per SEGMENT (~1.5 log2(buckets) curve operations next to the 2 S of the running sums: 40 % of the
// reduction's work at S = 16) becomes one per GROUP of MSM_GROUP segments: over a group g (s = G g + j) the same running
// sums over the T_j give  sum_j j T_j  and  sum_j T_j, and
//   sum_j [A_j + (S (G g + j) + 1) T_j] = sum_j A_j + S sum_j j T_j + (S G g + 1) sum_j T_j.
// 2.3 instead of 3.5 curve operations per bucket.  Packed jobs (bucket index = lo | hi << shift, shift >= log2(S G)): the
// low part weighs  sum_j A_j + S sum_j j T_j + lo(g) sum_j T_j,  the high part  hi(g) sum_j T_j.
constexpr uint32_t MSM_GROUP = 16;
__global__ __launch_bounds__(64) void msm_group_reduce_kernel(MsmPlanDev plan, const G1Xyzz* __restrict__ seg_a,
                                                              const G1Xyzz* __restrict__ seg_t, G1Xyzz* __restrict__ grp_out,
                                                              size_t total_groups) {
  // one QUAD of lanes per group (ec.cuh quad-cooperative arithmetic): there are MSM_GROUP times fewer groups than segments -
  // far too few to fill the chip - and a group is a chain of ~85 dependent curve operations
  const bool lead = (threadIdx.x & 3u) == 0;
  const size_t quads = ((size_t)gridDim.x * blockDim.x) >> 2;
  for (size_t gi = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; gi < total_groups; gi += quads) {
    int j = 0;
    while (j + 1 < plan.num_jobs && plan.job[j + 1].grp_base <= gi) j++;
    const MsmJobDev& jb = plan.job[j];  // (jobs that are not two_level own no group: grp_base repeats, the search passes them)
    const uint32_t local = (uint32_t)(gi - jb.grp_base), gpw = jb.seg_per_win / MSM_GROUP;
    const uint32_t w = local / gpw, g = local % gpw;
    if (jb.pack_shift && w) continue;  // (written by the quad of "window" 0)
    const size_t s0 = (size_t)jb.seg_base + (size_t)w * jb.seg_per_win + (size_t)g * MSM_GROUP;
    G1Xyzz run = G1Xyzz::identity(), acc = G1Xyzz::identity(), sum_a = G1Xyzz::identity();
    for (int d = (int)MSM_GROUP - 1; d >= 0; d--) {
      acc = add_quad(acc, run);
      run = add_quad(run, seg_t[s0 + d]);
      sum_a = add_quad(sum_a, seg_a[s0 + d]);
    }
    for (uint32_t k = 1; k < jb.seg_size; k <<= 1) acc = dbl_quad(acc);  // S sum_j j T_j
    sum_a = add_quad(sum_a, acc);
    const uint32_t d0 = g * MSM_GROUP * jb.seg_size;  // first bucket of the group
    if (jb.pack_shift) {
      const uint32_t lo = d0 & ((1u << jb.pack_shift) - 1u), hi = d0 >> jb.pack_shift;
      const G1Xyzz high = mul_small_quad(run, hi);
      const G1Xyzz low = add_quad(sum_a, mul_small_quad(run, lo));
      if (lead) grp_out[gi + gpw] = high, grp_out[gi] = low;
    } else {
      const G1Xyzz r = add_quad(sum_a, mul_small_quad(run, d0 + 1));  // bucket index b holds digit b + 1
      if (lead) grp_out[gi] = r;
    }
  }
}

__global__ __launch_bounds__(64) void msm_segment_reduce_kernel(MsmPlanDev plan, const G1Xyzz* __restrict__ buckets,
                                                                G1Xyzz* __restrict__ seg_out, G1Xyzz* __restrict__ seg_t,
                                                                size_t total_segs) {
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < total_segs; s += (size_t)gridDim.x * blockDim.x) {
    int j = 0;
    while (j + 1 < plan.num_jobs && plan.job[j + 1].seg_base <= s) j++;
    const MsmJobDev& jb = plan.job[j];
    uint32_t local = (uint32_t)(s - jb.seg_base);
    uint32_t w = local / jb.seg_per_win, seg = local % jb.seg_per_win;
    if (jb.pack_shift && w) continue;  // (written by the thread of "window" 0)
    uint32_t d0 = seg * jb.seg_size;
    const G1Xyzz* b = buckets + jb.key_base + (jb.pack_shift ? 0 : (size_t)w * jb.win_stride) + d0;
    G1Xyzz run = G1Xyzz::identity(), acc = G1Xyzz::identity();
    for (int d = (int)jb.seg_size - 1; d >= 0; d--) {
      acc = add(acc, run);
      run = add(run, b[d]);
    }
    if (jb.two_level) {  // A_s and T_s: the multiples are the group kernel's business
      seg_out[s] = acc;
      seg_t[s] = run;
      continue;
    }
    if (jb.pack_shift) {
      // bucket index = packed value: "window" 0 weighs it with its low part (linear inside the aligned segment), 1 with
      // its high part (constant inside it); the thread of window 0 writes both
      const uint32_t lo = d0 & ((1u << jb.pack_shift) - 1u), hi = d0 >> jb.pack_shift;
      seg_out[s + jb.seg_per_win] = mul_small(run, hi);
      acc = add(acc, mul_small(run, lo));
    } else {
      acc = add(acc, mul_small(run, d0 + 1));  // bucket index b holds digit b + 1
    }
    seg_out[s] = acc;
  }
}

// quad-cooperative form for batches with few segments (latency-bound: the chain is 2 S additions + a small multiple)
__global__ __launch_bounds__(64) void msm_segment_reduce_quad_kernel(MsmPlanDev plan, const G1Xyzz* __restrict__ buckets,
                                                                     G1Xyzz* __restrict__ seg_out, size_t total_segs) {
  const bool lead = (threadIdx.x & 3u) == 0;
  const size_t quads = ((size_t)gridDim.x * blockDim.x) >> 2;
  for (size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; s < total_segs; s += quads) {
    int j = 0;
    while (j + 1 < plan.num_jobs && plan.job[j + 1].seg_base <= s) j++;
    const MsmJobDev& jb = plan.job[j];
    uint32_t local = (uint32_t)(s - jb.seg_base);
    uint32_t w = local / jb.seg_per_win, seg = local % jb.seg_per_win;
    if (jb.pack_shift && w) continue;  // (written by the thread of "window" 0)
    uint32_t d0 = seg * jb.seg_size;
    const G1Xyzz* b = buckets + jb.key_base + (jb.pack_shift ? 0 : (size_t)w * jb.win_stride) + d0;
    G1Xyzz run = G1Xyzz::identity(), acc = G1Xyzz::identity();
    for (int d = (int)jb.seg_size - 1; d >= 0; d--) {
      acc = add_quad(acc, run);
      run = add_quad(run, b[d]);
    }
    if (jb.pack_shift) {
      const uint32_t lo = d0 & ((1u << jb.pack_shift) - 1u), hi = d0 >> jb.pack_shift;
      const G1Xyzz high = mul_small_quad(run, hi);
      if (lead) seg_out[s + jb.seg_per_win] = high;
      acc = add_quad(acc, mul_small_quad(run, lo));
    } else {
      acc = add_quad(acc, mul_small_quad(run, d0 + 1));  // bucket index b holds digit b + 1
    }
    if (lead) seg_out[s] = acc;
  }
}

// `nsplit` workgroups per window of a job: each sums a contiguous share of the window's segment partials (the host adds
// the nsplit shares: a host addition is ~0.4 us, a device addition on this under-filled launch ~20 us, so a window with
// 16 K segments must not be 64 dependent additions per thread).  Per job: the few very wide windows (packed column
// pairs: 2^16 segments) take 32 shares without multiplying the host's additions for the hundreds of ordinary ones.
// The sums go straight into pinned host memory; the workgroup that finishes last publishes the flag the
// host spins on (same ticket protocol as the sum-check rounds): no device-to-host copy, no stream synchronise.
__global__ __launch_bounds__(512) void msm_window_sum_kernel(MsmPlanDev plan, const G1Xyzz* __restrict__ seg_out,
                                                             const G1Xyzz* __restrict__ grp_out, G1Xyzz* __restrict__ win_out,
                                                             ScFinishArgs fin) {
  // 128 quads of lanes (ec.cuh: quad-cooperative additions): a share of <= 1024 partials is 8 + 7 dependent additions
  __shared__ G1Xyzz lds[128];
  int j = 0;
  while (j + 1 < plan.num_jobs && plan.job[j + 1].share_base <= blockIdx.x) j++;
  const MsmJobDev& jb = plan.job[j];
  const uint32_t nsplit = jb.nsplit;
  const uint32_t w = (blockIdx.x - jb.share_base) / nsplit, part = (blockIdx.x - jb.share_base) % nsplit;
  const uint32_t q = threadIdx.x >> 2;
  const bool lead = (threadIdx.x & 3u) == 0;
  const uint32_t share = (jb.sum_per_win + nsplit - 1) / nsplit;
  const uint32_t lo = part * share, hi = min(lo + share, jb.sum_per_win);
  // (two-level jobs: the groups' partials, one per MSM_GROUP segments)
  const G1Xyzz* src = jb.two_level ? grp_out + jb.grp_base + (size_t)w * jb.sum_per_win : seg_out + jb.seg_base + (size_t)w * jb.seg_per_win;
  G1Xyzz acc = G1Xyzz::identity();
  for (uint32_t i = lo + q; i < hi; i += 128) acc = add_quad(acc, src[i]);
  if (lead) lds[q] = acc;
  __syncthreads();
  const uint32_t live = hi > lo ? hi - lo : 0;
  for (uint32_t off = 64; off > 0; off >>= 1) {
    if (q < off && q + off < live) {
      const G1Xyzz v = add_quad(lds[q], lds[q + off]);
      if (lead) lds[q] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    win_out[blockIdx.x] = lds[0];
    __threadfence_system();
    bool last = gridDim.x == 1;
    if (!last) last = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == fin.last_ticket;
    if (last) __hip_atomic_store(fin.flag, fin.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------ derived jobs (MsmJob::derived_parent)
// list entry p of derived job j: (key_base_j + T[order[p]], parent bucket order[p]); sorted by key because `order` is
// sorted by T.  The continuation-level kernels then sum the runs into the derived job's buckets.
constexpr int MSM_MAX_DERIVED = 16;
struct MsmDerivedDev {
  uint32_t count;
  uint32_t off[MSM_MAX_DERIVED + 1];  // list offsets
  uint32_t key_base[MSM_MAX_DERIVED], parent_key_base[MSM_MAX_DERIVED];
  const uint32_t* table[MSM_MAX_DERIVED];
  const uint32_t* order[MSM_MAX_DERIVED];
};
__global__ void msm_derived_gather_kernel(MsmDerivedDev dd, const G1Xyzz* __restrict__ buckets,
                                          uint32_t* __restrict__ out_key, G1Xyzz* __restrict__ out_pt) {
  const uint32_t total = dd.off[dd.count];
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
    uint32_t j = 0;
    while (j + 1 < dd.count && dd.off[j + 1] <= p) j++;
    const uint32_t d = dd.order[j][p - dd.off[j]];
    const uint32_t v = dd.table[j][d];
    // no sentinels (they would split a run into several "first" segments): an empty parent bucket is the identity,
    // and T[d] = 0 lands in the derived job's bucket 0, which the reduction weighs with 0
    // (bucket index = value - 1; a zero value has no bucket)
    out_key[p] = dd.key_base[j] + (v ? v - 1u : 0u);
    out_pt[p] = (v && d) ? buckets[dd.parent_key_base[j] + d - 1u] : G1Xyzz::identity();
  }
}

// ------------------------------------------------------------------ host driver
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
// tuning knobs (development): window = floor(log2 n) - LH_MSM_C_OFF capped at LH_MSM_C_MAX; LH_MSM_K = entries per
// accumulate thread (0: by batch size)
static const int MSM_C_OFF = env_int("LH_MSM_C_OFF", 4), MSM_C_MAX = env_int("LH_MSM_C_MAX", 17),
                 MSM_K = env_int("LH_MSM_K", 0),
                 MSM_QUAD_MAX = env_int("LH_MSM_QUAD_MAX", 262144);  // lists / segment counts up to which a quad of lanes
                                                                      // shares one curve addition (0: never)

int msm_slab_log() {
  // (2^23 while every slab was a library call of its own; the batched sort has no per-slab cost: 2^16, tools sweep in
  // profiles/README.md round 3)
  static const int v = env_int("LH_MSM_SLAB_LOG", 16);
  return v;
}

// the continuation levels of one list: linear levels (fan-in K2) while the list is long, tree levels at the end
static void msm_continuation_levels(Ctx& c, hipStream_t stream, const uint32_t* ckey, const G1Xyzz* cpt, size_t n_in, G1Xyzz* buckets,
                                    uint32_t* cnt) {
  static const int MSM_K2 = env_int("LH_MSM_K2", 4);  // continuation fan-in: a level costs ~K2 dependent additions, there
                                                      // are log_K2(chunks) levels; swept 2..16, 3-4 is best (2^16: 10.1 -> 9.3 ms)
  static const int TREE_MAX = env_int("LH_MSM_TREE_MAX", 262144);  // lists of up to this many slots go by trees (0: never; sweep: profiles/r04_ab_msm_tree.txt)
  const uint32_t K2 = (uint32_t)MSM_K2;
  int lvl = 0;
  while (true) {
    const bool tree = n_in <= (size_t)TREE_MAX;
    const size_t fan = tree ? (size_t)64 : (size_t)K2;  // (256-slot tiles measured slower: profiles/r04_ab_msm_tree.txt)
    const size_t nc = (n_in + fan - 1) / fan;
    uint32_t* okey = c.arena.alloc_n<uint32_t>(nc);
    G1Xyzz* opt = c.arena.alloc_n<G1Xyzz>(nc);
    if (tree)
      hipLaunchKernelGGL((msm_accumulate_tree_quad_kernel<64>), dim3((unsigned)std::min<size_t>(nc, 1 << 16)), dim3(256), 0,
                         stream, ckey, cpt, n_in, buckets, okey, opt, cnt + lvl, cnt + lvl + 1);
    else if (n_in <= (size_t)MSM_QUAD_MAX)  // far below one wave per SIMD: a quad of lanes per entry
      hipLaunchKernelGGL(msm_accumulate_n_quad_kernel, dim3((unsigned)((4 * n_in + 127) / 128)), dim3(128), 0, stream,
                         ckey, cpt, n_in, K2, buckets, okey, opt, cnt + lvl, cnt + lvl + 1);
    else
      hipLaunchKernelGGL(msm_accumulate_n_kernel, dim3((unsigned)std::min<size_t>((n_in + 127) / 128, 1 << 16)), dim3(128), 0,
                         stream, ckey, cpt, n_in, K2, buckets, okey, opt, cnt + lvl, cnt + lvl + 1);
    lvl++;
    LH_REQUIRE(lvl < 30, LH_ERR_ARG, "msm: continuation list too long");
    if (n_in <= fan) break;  // a single chunk / tile: no continuation can remain
    ckey = okey;
    cpt = opt;
    n_in = nc;
  }
}

static uint32_t pick_window(size_t n, uint32_t bits) {
  uint32_t lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = (int)lg - MSM_C_OFF;
  if (c < 4) c = 4;
  if (c > MSM_C_MAX) c = MSM_C_MAX;
  if ((uint32_t)c > bits) c = (int)bits;
  return (uint32_t)c;
}

static inline host::G1Xyzz to_host(const G1Xyzz& p) {
  host::G1Xyzz r;
  memcpy(&r, &p, sizeof(r));
  return r;
}

// One planned sub-batch: the jobs `idx` (positions in the caller's array) with their own key / segment / share numbering,
// buffers and flag.  A batch is one sub-batch, or - msm_pick_split - two that run as a pipeline: the first half on the ctx's
// stream, the second half's accumulation and tails on the ctx's low-priority aux stream beside the first half's tails.
namespace {
struct MsmSub {
  std::vector<size_t> idx;
  MsmPlanDev plan;
  std::vector<uint32_t> bits, sort_bits;
  std::vector<char> derived, slab;
  size_t num_derived = 0, max_entries = 0, small_entries = 0, max_n = 0, nbuckets = 0, nsegs = 0, nwins = 0, ngroups = 0, nblocks = 0;
  size_t nchunks = 0;
  uint32_t nshares = 0, K = 0, seg_size = 0;
  bool plain_reduce = false;
  double total_pts = 0, full_pts = 0;
  uint32_t *ukey = nullptr, *uidx = nullptr, *skey = nullptr, *sidx = nullptr, *lvl_cnt = nullptr, *ckey = nullptr;
  G1Xyzz *buckets = nullptr, *seg_out = nullptr, *seg_t = nullptr, *grp_out = nullptr, *cpt = nullptr;
  G1Xyzz* win_out = nullptr;  // pinned
  uint8_t* h_tab = nullptr;   // pinned staging of the key-block table
  std::vector<G1Xyzz> wins;
  size_t pin_bytes() const { return ((size_t)nshares * sizeof(G1Xyzz) + nblocks + 255) & ~(size_t)255; }
};

// the shape a job will get: window bits, windows, sorted entries and buckets (what msm_plan_sub decides, without the layout)
struct MsmShape {
  uint32_t c, W, sort_bits;
  size_t entries, buckets;
};
MsmShape msm_job_shape(const MsmJob& in, uint32_t bits, bool derived) {
  MsmShape s;
  s.c = pick_window(in.n ? in.n : 1, bits ? bits : 1);
  if (in.pack_shift) s.c = std::max<uint32_t>(bits, 4);
  const bool is_signed = !in.scalars_u32;
  if (is_signed && s.c < 2) s.c = 2;
  s.W = bits ? ((is_signed ? bits + 1 : bits) + s.c - 1) / s.c : 0;
  if (!in.n) s.W = 0;
  s.sort_bits = is_signed ? s.c - 1 : s.c;
  s.entries = derived ? 0 : (size_t)in.n * s.W;
  s.buckets = (size_t)s.W << s.sort_bits;
  return s;
}
}  // namespace

// Which jobs of a chunk go into the SECOND half of a pipelined batch (empty: no split).  The first half's tails
// (continuation levels, bucket reduction, window sums: ~0.5 ns per bucket of dependent curve additions on an under-filled
// chip) run beside the second half's accumulation (~0.085 ns per entry at the multiplier's rate), the second half's tails
// stay exposed: the jobs with the most entries per bucket go last, as many as it takes to cover the first half's tails.
// A derived job goes where its parent goes.
static std::vector<char> msm_pick_split(Ctx& c, const MsmJob* jobs, const std::vector<size_t>& idx, const std::vector<uint32_t>& bits,
                                        const std::vector<char>& derived) {
  const size_t nj = idx.size();
  std::vector<char> second(nj, 0);
  // (a helper ctx works beside its owner's kernels already: a second stream of its own there cost the proof 0.7 ms)
  if (!c.opt.msm_half_batches || c.is_helper || nj < 2) return {};
  std::vector<size_t> E(nj), T(nj);
  size_t Et = 0, Tt = 0;
  for (size_t j = 0; j < nj; j++) {
    const MsmShape s = msm_job_shape(jobs[idx[j]], bits[j], derived[j] != 0);
    E[j] = s.entries, T[j] = s.buckets;
    Et += E[j], Tt += T[j];
  }
  // derived jobs count with their parents
  std::vector<long> parent(nj, -1);
  for (size_t j = 0; j < nj; j++) {
    if (!derived[j]) continue;
    for (size_t p = 0; p < nj; p++)
      if ((long)idx[p] == (long)jobs[idx[j]].derived_parent) parent[j] = (long)p;
    if (parent[j] >= 0) T[parent[j]] += T[j], T[j] = 0;
  }
  // (LH_MSM_HALF_MIN_LOG below 16 is a test shape: every batch with two jobs that have entries is split, whatever its size)
  static const size_t min_entries = (size_t)env_int("LH_MSM_HALF_MIN_LOG", 24);
  static const int cover = env_int("LH_MSM_HALF_COVER", 12);  // entries of the second half per bucket of the first
  const bool forced = min_entries < 16;  // (16 .. 23: a lower threshold with the floors below in force - measurements)
  if (Et < ((size_t)1 << min_entries) || (!forced && Tt < ((size_t)1 << 17))) return {};
  std::vector<size_t> order;
  for (size_t j = 0; j < nj; j++)
    if (parent[j] < 0 && E[j]) order.push_back(j);
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    const double ra = (double)E[a] / (double)(T[a] + 1), rb = (double)E[b] / (double)(T[b] + 1);
    return ra != rb ? ra > rb : a < b;
  });
  size_t Eb = 0, Tb = 0, taken = 0;
  for (size_t j : order) {
    if (Eb >= (size_t)cover * (Tt - Tb) || taken + 1 == order.size()) break;
    second[j] = 1, Eb += E[j], Tb += T[j], taken++;
  }
  // worth it only when the first half has tails to hide and both halves still fill the chip
  if (!taken || Eb == 0 || Et == Eb) return {};
  if (!forced && (Tt - Tb < ((size_t)1 << 16) || Eb < ((size_t)1 << 22) || Et - Eb < ((size_t)1 << 20))) return {};
  for (size_t j = 0; j < nj; j++)
    if (parent[j] >= 0) second[j] = second[parent[j]];
  return second;
}

// layout of one sub-batch: windows, key / segment / share ranges, entry positions (no device work)
// (`whole`: the plan of the undivided batch this sub-batch is a half of - segment size, reduction form and entries per
// accumulate thread are the batch's, not the half's: a half must not fall back to the latency-bound forms of a small batch)
static void msm_plan_sub(Ctx& c, const MsmJob* jobs, MsmSub& s, const MsmSub* whole = nullptr) {
  const size_t nj = s.idx.size();
  MsmPlanDev& plan = s.plan;
  plan.num_jobs = (int)nj;
  plan.job_of_block = nullptr;
  s.sort_bits.assign(nj, 0), s.slab.assign(nj, 0);
  s.num_derived = 0;
  for (size_t j = 0; j < nj; j++) s.num_derived += s.derived[j] ? 1 : 0;
  // buckets per reduce thread: the segment kernel is a chain of 2 S additions plus a small-scalar multiplication
  // per thread.  Few buckets in total = too few threads to fill the chip = pure latency: shorter segments then
  // (S = 4: ~30 dependent curve operations instead of ~51); many buckets = throughput: S = 16 does least work.
  uint32_t seg_size = 16;
  {
    size_t est = 0;
    for (size_t j = 0; j < nj; j++) {
      const MsmJob& in = jobs[s.idx[j]];
      const uint32_t bits = s.bits[j];
      if (!in.n || !bits) continue;
      const uint32_t cw = pick_window(in.n, bits);
      est += ((size_t)(bits + cw) / cw) << (in.scalars_u32 ? cw : cw - 1);
    }
    static const int forced = env_int("LH_MSM_SEG", 0);
    seg_size = forced ? (uint32_t)forced : est <= ((size_t)1 << 18) ? 4u : est <= ((size_t)1 << 20) ? 8u : 16u;
    if (whole) seg_size = whole->seg_size;
    s.seg_size = seg_size;
  }
  uint32_t key = 0, seg = 0, win = 0;
  // a job of >= 2^LH_MSM_SLAB_LOG points sorts each of its (window) slabs by the digit bits alone
  static const int slab_log = msm_slab_log();
  for (size_t j = 0; j < nj; j++) {
    const MsmJob& in = jobs[s.idx[j]];
    MsmJobDev& jd = plan.job[j];
    const uint32_t bits = s.bits[j];
    jd.scalars = in.scalars;
    jd.is_u32 = in.scalars_u32 ? 1 : 0;
    jd.n = (uint32_t)in.n;
    jd.bases = in.bases;
    if (s.derived[j]) jd.n = 0;  // emits no (point, window) entries: its buckets are filled from the parent's
    jd.c = pick_window(in.n ? in.n : 1, bits ? bits : 1);
    jd.pack_shift = 0;
    if (in.pack_shift) {
      LH_REQUIRE(in.scalars_u32 && in.out_second && in.pack_shift >= 4 && bits <= MSM_PACK_MAX_BITS && !s.derived[j],
                 LH_ERR_ARG, "msm: bad packed job");
      jd.pack_shift = in.pack_shift;
      jd.c = std::max<uint32_t>(bits, 4);  // one window: the bucket index is the packed value
    }
    jd.is_signed = in.scalars_u32 ? 0 : 1;
    if (jd.is_signed && jd.c < 2) jd.c = 2;
    // signed digits need one extra bit of head room for the last carry
    jd.W = bits ? ((jd.is_signed ? bits + 1 : bits) + jd.c - 1) / jd.c : 0;
    if (!in.n) jd.W = 0;
    // window table (MsmJob::win_table): the windows of the table's width share one bucket set
    jd.merged = 0;
    if (in.win_table && jd.is_signed && !s.derived[j] && in.n && in.win_table_c >= 2) {
      const uint32_t Wt = (bits + 1 + in.win_table_c - 1) / in.win_table_c;
      if (Wt >= 2 && Wt <= in.win_table_W && (size_t)Wt * in.n < ((size_t)1 << 31)) {
        jd.merged = 1;
        jd.c = in.win_table_c;
        jd.W = Wt;
        jd.bases = in.win_table;
      }
    }
    // digit d > 0 lives in bucket d - 1: signed digits 1 .. 2^(c-1), unsigned 1 .. 2^c - 1
    s.sort_bits[j] = jd.is_signed ? jd.c - 1 : jd.c;
    const uint32_t nb = 1u << s.sort_bits[j];
    jd.red_W = jd.pack_shift && jd.W ? 2 : jd.merged ? 1 : jd.W;
    jd.seg_size = seg_size;
    jd.seg_per_win = (nb + jd.seg_size - 1) / jd.seg_size;
    jd.win_stride = jd.seg_per_win * jd.seg_size;
    s.slab[j] = jd.n >= (1u << slab_log) && jd.W > 0 && nb >= jd.seg_size &&
                s.sort_bits[j] <= (jd.pack_shift ? MSM_PACK_MAX_BITS : 16u);
    jd.presorted = s.slab[j] && jd.W == 1 && in.scalars_u32 && in.sorted_scalars && in.sorted_index && !s.derived[j];
    // a slab-sorted job's bucket ranges start at multiples of 2^sort_bits: the low bits of a key are the bucket index
    const uint32_t align_bits = s.slab[j] ? std::max<uint32_t>(KEY_BLOCK_BITS, s.sort_bits[j]) : KEY_BLOCK_BITS;
    key = (key + (1u << align_bits) - 1) & ~((1u << align_bits) - 1);
    jd.key_base = key;
    jd.seg_base = seg;
    jd.win_base = win;
    jd.two_level = 0, jd.grp_base = 0, jd.sum_per_win = jd.seg_per_win;  // (decided below, once the batch's size is known)
    key += (jd.merged ? 1 : jd.W) * jd.win_stride;
    seg += jd.red_W * jd.seg_per_win;
    win += jd.red_W;
    LH_REQUIRE(in.n < 0x7fffffffu, LH_ERR_ARG, "msm: too many points");
    s.max_n = std::max(s.max_n, in.n);
  }
  // entry layout: the slabs of the small jobs first (one global sort), then the big jobs' slabs
  s.max_entries = s.small_entries = 0;
  for (int pass = 0; pass < 2; pass++)
    for (size_t j = 0; j < nj; j++) {
      if ((int)s.slab[j] != pass) continue;
      plan.job[j].entry_base = (uint32_t)s.max_entries;
      s.max_entries += (size_t)plan.job[j].n * plan.job[j].W;
      LH_REQUIRE(s.max_entries < ((size_t)1 << 32), LH_ERR_ARG, "msm: batch too large for 32-bit entry indices");
      if (!pass) s.small_entries = s.max_entries;
    }
  s.nbuckets = key, s.nsegs = seg, s.nwins = win;
  if (getenv("LH_MSM_DEBUG"))
    for (size_t j = 0; j < nj; j++)
      fprintf(stderr, "[msm] job %zu n %u bits %u c %u W %u entries %zu%s%s%s%s\n", s.idx[j], plan.job[j].n, s.bits[j], plan.job[j].c,
              plan.job[j].W, (size_t)plan.job[j].n * plan.job[j].W, plan.job[j].is_signed ? " fr" : " u32",
              s.derived[j] ? " derived" : "", plan.job[j].merged ? " table" : "", plan.job[j].pack_shift ? " packed" : "");
  // two-level reduction: throughput-bound batches (the plain segment kernel runs), jobs whose windows hold at least
  // MSM_GROUP^2 segments, packed jobs only when a group never straddles a change of the high part
  static const int two_level_on = env_int("LH_MSM_TWO_LEVEL", 1);
  const bool plain_reduce = whole ? whole->plain_reduce : s.nsegs > (size_t)MSM_QUAD_MAX / 2;
  s.plain_reduce = plain_reduce;
  s.ngroups = 0;
  for (size_t j = 0; j < nj; j++) {
    MsmJobDev& jd = plan.job[j];
    jd.grp_base = (uint32_t)s.ngroups;
    const bool ok = two_level_on && plain_reduce && jd.red_W && jd.seg_size >= 2 && (jd.seg_size & (jd.seg_size - 1)) == 0 &&
                    jd.seg_per_win >= MSM_GROUP * MSM_GROUP && jd.seg_per_win % MSM_GROUP == 0 &&
                    (!jd.pack_shift || ((1u << jd.pack_shift) % (MSM_GROUP * jd.seg_size)) == 0);
    if (!ok) continue;
    jd.two_level = 1;
    jd.sum_per_win = jd.seg_per_win / MSM_GROUP;
    s.ngroups += (size_t)jd.red_W * jd.sum_per_win;
  }
  // window-sum shares: enough workgroups that no thread adds more than ~4 segment partials in sequence, few enough
  // that the host's share of the additions stays in the microseconds
  uint32_t nsplit = 1;
  {
    uint32_t max_spw = 1;
    for (size_t j = 0; j < nj; j++)
      if (plan.job[j].W && plan.job[j].sum_per_win < 16384) max_spw = std::max(max_spw, plan.job[j].sum_per_win);
    while (nsplit < 32 && max_spw / nsplit > 1024 && s.nwins * nsplit * 2 <= 4096) nsplit *= 2;
  }
  s.nshares = 0;
  for (size_t j = 0; j < nj; j++) {
    MsmJobDev& jd = plan.job[j];
    jd.nsplit = jd.sum_per_win >= 16384 ? std::max<uint32_t>(nsplit, 32u) : nsplit;
    jd.share_base = s.nshares;
    s.nshares += jd.red_W * jd.nsplit;
  }
  s.nblocks = (s.nbuckets >> KEY_BLOCK_BITS) + 1;
  s.total_pts = s.full_pts = 0;
  for (size_t j = 0; j < nj; j++) s.total_pts += plan.job[j].n, s.full_pts += plan.job[j].is_u32 ? 0 : plan.job[j].n;
  // entries per accumulate thread: enough chunks to fill the chip, few enough that the continuation list stays small
  // (measured: tools/msm_sweep.sh; 2^24 lookups 141 -> 132 ms with K 32 -> 128)
  const size_t me = s.max_entries;
  s.K = me > ((size_t)1 << 26) ? 128 : me > ((size_t)1 << 25) ? 64 : me > ((size_t)1 << 23) ? 32 : me > ((size_t)1 << 21) ? 16 : me > ((size_t)1 << 18) ? 8 : 4;
  if (MSM_K > 0) s.K = (uint32_t)MSM_K;
  if (whole) s.K = whole->K;
  s.nchunks = (me + s.K - 1) / s.K;
  s.wins.assign(s.nshares, G1Xyzz::identity());
}

// workspace of a sub-batch (the caller's ArenaScope owns it) and its share of the batch's pinned block
static void msm_sub_alloc(Ctx& c, MsmSub& s, uint8_t* pin_at) {
  s.ukey = c.arena.alloc_n<uint32_t>(s.max_entries);
  s.uidx = c.arena.alloc_n<uint32_t>(s.max_entries);
  // + 256: accumulate0 reads whole 16-byte groups up to the end of the last (padded) chunk
  s.skey = c.arena.alloc_n<uint32_t>(s.max_entries + 256);
  s.sidx = c.arena.alloc_n<uint32_t>(s.max_entries + 256);
  s.buckets = c.arena.alloc_n<G1Xyzz>(s.nbuckets);
  s.seg_out = c.arena.alloc_n<G1Xyzz>(s.nsegs);
  s.seg_t = s.ngroups ? c.arena.alloc_n<G1Xyzz>(s.nsegs) : nullptr;   // T_s of the two-level jobs' segments
  s.grp_out = s.ngroups ? c.arena.alloc_n<G1Xyzz>(s.ngroups) : nullptr;
  s.lvl_cnt = c.arena.alloc_n<uint32_t>(64);
  s.ckey = c.arena.alloc_n<uint32_t>(s.nchunks);
  s.cpt = c.arena.alloc_n<G1Xyzz>(s.nchunks);
  // pinned host memory: the window sums (written by the last kernel) followed by the key-block table staging
  s.win_out = (G1Xyzz*)pin_at;
  s.h_tab = pin_at + (size_t)s.nshares * sizeof(G1Xyzz);
}

// steps 1-2 on the ctx's stream: digits and sort
static void msm_sub_entries(Ctx& c, const MsmJob* jobs, MsmSub& s) {
  const size_t nj = s.idx.size();
  MsmPlanDev& plan = s.plan;
  {
    for (size_t j = 0; j < nj; j++) {
      const size_t b0 = plan.job[j].key_base >> KEY_BLOCK_BITS;
      const size_t b1 = j + 1 < nj ? plan.job[j + 1].key_base >> KEY_BLOCK_BITS : s.nblocks;
      for (size_t b = b0; b < b1; b++) s.h_tab[b] = (uint8_t)j;
    }
    uint8_t* d_tab = c.arena.alloc_n<uint8_t>(s.nblocks);
    LH_HIP(hipMemcpyAsync(d_tab, s.h_tab, s.nblocks, hipMemcpyHostToDevice, c.stream));
    plan.job_of_block = d_tab;
  }
  for (size_t j = 0; j < nj; j++) c.route.v[RouteStats::WIN_TABLE_JOBS] += plan.job[j].merged ? 1 : 0;
  LH_HIP(hipMemsetAsync(s.lvl_cnt, 0, 64 * sizeof(uint32_t), c.stream));
  LH_HIP(hipMemsetAsync(s.buckets, 0, s.nbuckets * sizeof(G1Xyzz), c.stream));
  unsigned key_bits = 1;
  while (((size_t)1 << key_bits) <= s.nbuckets) key_bits++;
  {
    ProfScope ps(c, "msm_digits", 32.0 * s.full_pts + 4.0 * (s.total_pts - s.full_pts) + 8.0 * s.max_entries, s.full_pts,
                 s.total_pts);
    dim3 g((unsigned)std::min<size_t>((s.max_n + 255) / 256, 2048), (unsigned)nj);
    hipLaunchKernelGGL(msm_emit_kernel, g, dim3(256), 0, c.stream, plan, s.ukey, s.uidx);
  }
  {
    ProfScope ps(c, "msm_sort", 32.0 * s.max_entries, 0, (double)s.max_entries);
    // small jobs: one sort by the whole key; big jobs: every (job, window) slab by its digit bits only - all of them
    // as ONE batch of the radix sort (sort.hip): three launches per pass for the whole MSM batch
    std::vector<SortSlab> sorts;
    if (s.small_entries) sorts.push_back(SortSlab{s.ukey, s.skey, s.uidx, s.sidx, s.small_entries, key_bits});
    for (size_t j = 0; j < nj; j++) {
      if (!s.slab[j]) continue;
      const MsmJobDev& jd = plan.job[j];
      if (jd.presorted) {
        const MsmJob& in = jobs[s.idx[j]];
        hipLaunchKernelGGL(msm_presorted_kernel, dim3((unsigned)std::min<size_t>((jd.n + 255) / 256, 4096)), dim3(256), 0,
                           c.stream, jd.key_base, in.sorted_scalars, in.sorted_index, (size_t)jd.n, s.skey + jd.entry_base,
                           s.sidx + jd.entry_base);
        continue;
      }
      if (jd.merged) {  // one bucket set: the entries of all windows are one slab
        const size_t e = jd.entry_base;
        sorts.push_back(SortSlab{s.ukey + e, s.skey + e, s.uidx + e, s.sidx + e, (size_t)jd.n * jd.W, s.sort_bits[j]});
        continue;
      }
      for (uint32_t w = 0; w < jd.W; w++) {
        const size_t e = (size_t)jd.entry_base + (size_t)w * jd.n;
        sorts.push_back(SortSlab{s.ukey + e, s.skey + e, s.uidx + e, s.sidx + e, jd.n, s.sort_bits[j]});
      }
    }
    if (!sorts.empty()) sort_pairs_u32_batched(c, sorts.data(), sorts.size());
  }
}

// step 3 on `stream`: level 0 of the segmented accumulation
static void msm_sub_accumulate(Ctx& c, MsmSub& s, hipStream_t stream) {
  MsmPlanDev& plan = s.plan;
  {
    // MSM algorithmic bytes: 96 B per point (32 B scalar + 64 B base), 68 B for a u32 column,
    // whatever the number of windows; `items` = sorted (point, window) entries, a mixed add is 10 Fq muls
    ProfScope ps(c, "msm_accumulate0", 96.0 * s.full_pts + 68.0 * (s.total_pts - s.full_pts), 10.0 * (double)s.max_entries, (double)s.max_entries);
    // (a grid capped to the chip's resident workgroups - for a helper ctx, so that the owner's latency-bound kernels find
    // wave slots, or for every ctx, persistent style - was measured in round 5 and bought nothing: profiles/README.md)
    const size_t acc_grid = std::min<size_t>((s.nchunks + 127) / 128, 1 << 16);
    Ctx::LiveRec lr;
    if (c.live) {  // (dev.hpp: the launch's span on its own stream, nothing waited for)
      memset(&lr.rec, 0, sizeof(lr.rec));
      snprintf(lr.rec.name, sizeof lr.rec.name, "msm_accumulate0");
      lr.rec.bytes = 96.0 * s.full_pts + 68.0 * (s.total_pts - s.full_pts), lr.rec.muls = 10.0 * (double)s.max_entries;
      lr.rec.items = (double)s.max_entries;
      lr.e0 = c.live_event(), lr.e1 = c.live_event(), lr.batch = c.live_batch;
      LH_HIP(hipEventRecord(lr.e0, stream));
    }
    hipLaunchKernelGGL(msm_accumulate0_kernel, dim3((unsigned)acc_grid), dim3(128), 0, stream, plan, s.max_entries, s.skey, s.sidx,
                       s.K, s.buckets, s.ckey, s.cpt, s.nchunks, s.lvl_cnt);
    if (c.live) {
      LH_HIP(hipEventRecord(lr.e1, stream));
      c.live_recs.push_back(lr);
    }
  }
}

// steps 4-5 on `stream` (the ctx's, or its aux stream beside the next half's front): continuation levels, derived jobs,
// bucket reduction, window sums; the last workgroup publishes `fin.seq` to `fin.flag`
static void msm_sub_tail(Ctx& c, const MsmJob* jobs, MsmSub& s, hipStream_t stream, const ScFinishArgs& fin) {
  const size_t nj = s.idx.size();
  MsmPlanDev& plan = s.plan;
  {
    ProfScope ps(c, "msm_accumulate_levels", 0, 0, (double)s.nchunks);
    msm_continuation_levels(c, stream, s.ckey, s.cpt, s.nchunks, s.buckets, s.lvl_cnt);
  }
  if (s.num_derived) {
    // derived jobs: (key, parent bucket) lists sorted by key, summed into the derived buckets by the same
    // continuation levels (a run shrinks K2-fold per level)
    MsmDerivedDev dd;
    memset(&dd, 0, sizeof(dd));
    for (size_t j = 0; j < nj; j++) {
      if (!s.derived[j]) continue;
      const MsmJob& in = jobs[s.idx[j]];
      size_t p = nj;
      for (size_t q = 0; q < nj; q++)
        if ((long)s.idx[q] == (long)in.derived_parent) p = q;
      LH_REQUIRE(p < nj, LH_ERR_ARG, "msm: a derived job lost its parent");
      const uint32_t k = dd.count++;
      dd.off[k + 1] = dd.off[k] + (1u << in.table_in_bits);
      dd.key_base[k] = plan.job[j].key_base;
      dd.parent_key_base[k] = plan.job[p].key_base;
      dd.table[k] = in.d_table, dd.order[k] = in.d_order;
    }
    const size_t nd = dd.off[dd.count];
    ProfScope ps(c, "msm_derived", 132.0 * nd, 14.0 * nd, (double)nd);
    uint32_t* dkey = c.arena.alloc_n<uint32_t>(nd);
    G1Xyzz* dpt = c.arena.alloc_n<G1Xyzz>(nd);
    hipLaunchKernelGGL(msm_derived_gather_kernel, dim3((unsigned)std::min<size_t>((nd + 255) / 256, 1024)), dim3(256), 0,
                       stream, dd, s.buckets, dkey, dpt);
    uint32_t* dcnt = s.lvl_cnt + 32;
    LH_HIP(hipMemsetAsync(dcnt, 1, sizeof(uint32_t), stream));  // "something continued into level 0"
    msm_continuation_levels(c, stream, dkey, dpt, nd, s.buckets, dcnt);
  }
  {
    ProfScope ps(c, "msm_bucket_reduce", 128.0 * s.nbuckets, 14.0 * 2.2 * s.nbuckets, (double)s.nbuckets);
    if (!s.plain_reduce)
      hipLaunchKernelGGL(msm_segment_reduce_quad_kernel, dim3((unsigned)((4 * s.nsegs + 63) / 64)), dim3(64), 0, stream,
                         plan, s.buckets, s.seg_out, s.nsegs);
    else
      hipLaunchKernelGGL(msm_segment_reduce_kernel, dim3((unsigned)std::min<size_t>((s.nsegs + 63) / 64, 1 << 16)),
                         dim3(64), 0, stream, plan, s.buckets, s.seg_out, s.seg_t, s.nsegs);
    if (s.ngroups)
      hipLaunchKernelGGL(msm_group_reduce_kernel, dim3((unsigned)std::min<size_t>((4 * s.ngroups + 63) / 64, 1 << 16)), dim3(64), 0,
                         stream, plan, s.seg_out, s.seg_t, s.grp_out, s.ngroups);
    hipLaunchKernelGGL(msm_window_sum_kernel, dim3(s.nshares), dim3(512), 0, stream, plan, s.seg_out, s.grp_out, s.win_out, fin);
    if (c.prof) c.sync();
  }
}

// a job whose window combine is at most ~40 doublings (~15 us on a host core): done inline - a 16-bit column of 2^16 points
// has two 12-bit windows, and handing its 12 doublings to the sleeping host pool cost ~100 us per batch of a 2^16 proof
static inline bool msm_job_is_light(const MsmJobDev& jd) { return jd.red_W >= 2 && (jd.red_W - 1) * jd.c <= 40; }

// step 6 on the host: sum_w 2^(c w) win[w] per job.  Jobs with doublings (~70 us of dependent doublings each) go to the
// host pool, the others (one window, packed pairs: a few additions) are done here
static void msm_sub_combine(MsmSub& s, std::vector<host::G1Xyzz>& sums /* [2 global job], [.. + 1]: out_second */) {
  const size_t nj = s.idx.size();
  const MsmPlanDev& plan = s.plan;
  auto share_sum = [&](const MsmJobDev& jd, uint32_t w) {
    host::G1Xyzz acc = host::G1Xyzz::identity();
    for (uint32_t part = 0; part < jd.nsplit; part++)
      acc = host::g1_add(acc, to_host(s.wins[(size_t)jd.share_base + (size_t)w * jd.nsplit + part]));
    return acc;
  };
  std::vector<size_t> heavy;
  for (size_t j = 0; j < nj; j++) {
    const MsmJobDev& jd = plan.job[j];
    if (jd.pack_shift) {  // the two "windows" are the two results
      if (jd.red_W) sums[2 * s.idx[j]] = share_sum(jd, 0), sums[2 * s.idx[j] + 1] = share_sum(jd, 1);
    } else if (jd.red_W <= 1 || jd.merged) {  // (a window table's job has one "window": no doublings)
      if (jd.red_W) sums[2 * s.idx[j]] = share_sum(jd, 0);
    } else if (msm_job_is_light(jd)) {  // a few dozen doublings: here, now (waking the pool costs more than they do)
      host::G1Xyzz acc = host::G1Xyzz::identity();
      for (int w = (int)jd.red_W - 1; w >= 0; w--) {
        for (uint32_t q = 0; q < jd.c; q++) acc = host::g1_dbl(acc);
        acc = host::g1_add(acc, share_sum(jd, (uint32_t)w));
      }
      sums[2 * s.idx[j]] = acc;
    } else {
      heavy.push_back(j);
    }
  }
  host_parallel_for(heavy.size(), [&](size_t k) {
    const size_t j = heavy[k];
    const MsmJobDev& jd = plan.job[j];
    host::G1Xyzz acc = host::G1Xyzz::identity();
    for (int w = (int)jd.red_W - 1; w >= 0; w--) {
      for (uint32_t q = 0; q < jd.c; q++) acc = host::g1_dbl(acc);
      acc = host::g1_add(acc, share_sum(jd, (uint32_t)w));
    }
    sums[2 * s.idx[j]] = acc;
  });
}

bool msm_batch(Ctx& c, const MsmJob* jobs, size_t num_jobs, G1Affine* out_host, const std::function<void()>* overlap) {
  bool overlap_done = overlap == nullptr;
  bool waited = false;  // the host waited for the ctx's stream at least once (everything queued before the batch has run)
  for (size_t base = 0; base < num_jobs; base += MSM_MAX_JOBS) {
    const size_t nj = std::min(num_jobs - base, (size_t)MSM_MAX_JOBS);
    // significant bits of every column (one cheap pass)
    std::vector<uint32_t> job_bits(nj, 32);
    {
      MsmPlanDev or_plan;  // (jobs whose width the caller promises take no part in the pass)
      or_plan.num_jobs = (int)nj;
      or_plan.job_of_block = nullptr;
      bool any_fr = false, any_unknown = false;
      size_t max_n0 = 0;
      for (size_t j = 0; j < nj; j++) {
        const MsmJob& in = jobs[base + j];
        LH_REQUIRE(in.n < ((size_t)1 << 31), LH_ERR_ARG, "msm: too many points");
        memset(&or_plan.job[j], 0, sizeof(MsmJobDev));
        or_plan.job[j].scalars = in.scalars;
        or_plan.job[j].is_u32 = in.scalars_u32 ? 1 : 0;
        or_plan.job[j].n = in.known_bits ? 0u : (uint32_t)in.n;
        any_fr |= in.n != 0;
        any_unknown |= !in.known_bits && in.n != 0;
        max_n0 = std::max(max_n0, in.n);
      }
      if (any_fr && any_unknown) {
        ArenaScope scope(c.arena);
        uint32_t* d_or = c.arena.alloc_n<uint32_t>(8 * nj);
        LH_HIP(hipMemsetAsync(d_or, 0, 8 * nj * sizeof(uint32_t), c.stream));
        dim3 g((unsigned)std::min<size_t>((max_n0 + 255) / 256, 256), (unsigned)nj);
        hipLaunchKernelGGL(msm_or_limbs_kernel, g, dim3(256), 0, c.stream, or_plan, d_or);
        uint32_t* h_or = (uint32_t*)c.pin(8 * MSM_MAX_JOBS * sizeof(uint32_t));
        LH_HIP(hipMemcpyAsync(h_or, d_or, 8 * nj * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
        c.sync();
        waited = true;
        for (size_t j = 0; j < nj; j++) {
          uint32_t bits = 0;
          for (int k = 7; k >= 0 && !bits; k--)
            if (h_or[8 * j + k]) bits = 32 * k + (32 - __builtin_clz(h_or[8 * j + k]));
          job_bits[j] = bits;  // 0: all-zero column
        }
      }
      for (size_t j = 0; j < nj; j++)
        if (jobs[base + j].known_bits) job_bits[j] = jobs[base + j].n ? std::min<uint32_t>(jobs[base + j].known_bits, jobs[base + j].scalars_u32 ? 32u : 254u) : 0;
    }
    // derived jobs (MsmJob::derived_parent): usable when the parent is a u32 column of the same points whose single
    // window is indexed by its value (window bits == significant bits <= the table's input bits)
    std::vector<char> derived(nj, 0);
    size_t num_derived = 0;
    for (size_t j = 0; j < nj; j++) {
      const MsmJob& in = jobs[base + j];
      const long p = (long)in.derived_parent - (long)base;
      if (in.derived_parent < 0 || p < 0 || p >= (long)nj || !in.d_table || !in.d_order) continue;
      const MsmJob& par = jobs[base + p];
      const uint32_t pb = job_bits[p];
      if (!in.scalars_u32 || !par.scalars_u32 || par.n != in.n || par.bases != in.bases || par.derived_parent >= 0) continue;
      if (!pb || pb > in.table_in_bits || in.table_in_bits > 20 || pick_window(par.n, pb) != pb) continue;
      if (!in.table_out_bits || num_derived == (size_t)MSM_MAX_DERIVED) continue;
      derived[j] = 1;
      num_derived++;
      job_bits[j] = std::min(job_bits[j], in.table_out_bits);  // (0 stays 0: an all-zero column)
      if (!job_bits[j] || pick_window(in.n, job_bits[j]) != job_bits[j]) derived[j] = 0, num_derived--;
    }
    // one sub-batch, or two halves that run as a pipeline (msm_pick_split)
    std::vector<size_t> all(nj);
    for (size_t j = 0; j < nj; j++) all[j] = base + j;
    const std::vector<char> second = msm_pick_split(c, jobs, all, job_bits, derived);
    std::vector<MsmSub> subs(second.empty() ? 1 : 2);
    for (size_t j = 0; j < nj; j++) {
      MsmSub& s = subs[second.empty() ? 0 : (size_t)second[j]];
      s.idx.push_back(base + j), s.bits.push_back(job_bits[j]), s.derived.push_back(derived[j]);
    }
    size_t total_entries = 0, pin_total = 0;
    MsmSub whole;
    if (subs.size() == 2) {
      whole.idx = all, whole.bits = job_bits, whole.derived = derived;
      msm_plan_sub(c, jobs, whole);
    }
    // (the second half with forms of its own - segment size, reduction kernels, entries per thread by ITS size - was measured
    //  and lost: the opening 19.8 -> 20.6 ms, tools/ab_half.sh of round 6)
    for (MsmSub& s : subs) {
      msm_plan_sub(c, jobs, s, subs.size() == 2 ? &whole : nullptr);
      total_entries += s.max_entries, pin_total += s.pin_bytes();
    }
    if (total_entries == 0) {
      for (size_t j = 0; j < nj; j++) {
        memset(&out_host[base + j], 0, sizeof(G1Affine));
        if (jobs[base + j].out_second) memset(jobs[base + j].out_second, 0, sizeof(G1Affine));
      }
      continue;
    }
    if (subs.size() == 2 && (!subs[0].max_entries || !subs[1].max_entries)) {  // (cannot happen with msm_pick_split's floors)
      subs.assign(1, whole);
      pin_total = subs[0].pin_bytes();
    }
    std::vector<host::G1Xyzz> sums_all(2 * num_jobs, host::G1Xyzz::identity());
    {
      ArenaScope scope(c.arena);
      uint8_t* pin_base = (uint8_t*)c.pin(pin_total);
      for (MsmSub& s : subs) {
        msm_sub_alloc(c, s, pin_base);
        pin_base += s.pin_bytes();
      }
      const bool piped = subs.size() == 2;
      // One batch: everything on the ctx's stream.  Two halves: both entry streams on the ctx's stream; then the FIRST half's
      // accumulation and tails stay there, and the second half's go to the aux stream, whose priority is the lowest the device
      // offers - its accumulation gets the wave slots the first half's kernels do not ask for: the drain of the first
      // accumulation (two launches on one stream would each pay their own), then everything the first half's latency-bound
      // tails leave idle.  The second half's tails end the batch.  Under the profiler: one stream, the launches one after
      // the other.
      for (MsmSub& s : subs) msm_sub_entries(c, jobs, s);
      hipStream_t side = c.stream;
      uint32_t seq_aux = 0;
      struct AuxGuard {  // an exception on the way out must not release the arena under kernels still queued on the aux stream
        Ctx& c;
        bool armed = false;
        ~AuxGuard() {
          if (armed && c.aux_stream) (void)hipStreamSynchronize(c.aux_stream);
        }
      } aux_guard{c};
      if (piped) {
        c.route.v[RouteStats::MSM_HALF_BATCHES]++;
        if (!c.prof) {
          c.aux_streams();
          LH_HIP(hipEventRecord(c.aux_ev, c.stream));
          LH_HIP(hipStreamWaitEvent(c.aux_stream, c.aux_ev, 0));
          side = c.aux_stream;
          aux_guard.armed = true;
        }
      }
      c.live_batch++;
      msm_sub_accumulate(c, subs[0], c.stream);
      if (piped) msm_sub_accumulate(c, subs[1], side);
      const uint32_t seq = c.next_seq();
      msm_sub_tail(c, jobs, subs[0], c.stream, c.finish_for(subs[0].nshares, nullptr, seq));
      if (piped) {
        seq_aux = c.next_seq();
        msm_sub_tail(c, jobs, subs[1], side, c.finish_for_aux(subs[1].nshares, seq_aux));
      }
      if (!overlap_done) overlap_done = true, (*overlap)();  // (the device is busy with this batch: the caller's host work now)
      {
        // the window combines follow when the device is through: workers awake and polling by then (batches of up to a few
        // milliseconds; a longer one lets them sleep again and pays the wake-up, which then no longer matters)
        size_t heavy_jobs = 0;
        for (const MsmSub& s : subs)
          for (size_t j = 0; j < s.idx.size(); j++)
            heavy_jobs += s.plan.job[j].red_W > 1 && !s.plan.job[j].pack_shift && !s.plan.job[j].merged && !msm_job_is_light(s.plan.job[j]);
        if (heavy_jobs > 1 && !c.prof) host_parallel_prewake(heavy_jobs, 4000);
      }
      c.host_stamp("msm:queued");
      c.wait_flag(seq);
      waited = true;
      if (piped) {
        // the first half's window sums are in: its doublings run on the host while the device works through the second half
        memcpy(subs[0].wins.data(), subs[0].win_out, (size_t)subs[0].nshares * sizeof(G1Xyzz));
        msm_sub_combine(subs[0], sums_all);
        c.host_stamp("msm:first_half_combined");
        c.wait_flag_aux(seq_aux);
        aux_guard.armed = false;  // (its last kernel has published: nothing of the batch is queued any more)
      }
      c.host_stamp("msm:window_sums");
      for (MsmSub& s : subs) {
        if (getenv("LH_MSM_DEBUG")) {
          uint32_t h_cnt[16];
          c.d2h(h_cnt, s.lvl_cnt, sizeof(h_cnt));
          fprintf(stderr, "[msm] jobs %zu entries %zu K %u nchunks %zu buckets %zu%s | continuation counts:", s.idx.size(), s.max_entries,
                  s.K, s.nchunks, s.nbuckets, piped ? (&s == &subs[0] ? " (first half)" : " (second half)") : "");
          for (int i = 0; i < 10; i++) fprintf(stderr, " %u", h_cnt[i]);
          fprintf(stderr, "\n");
        }
        if (!piped || &s == &subs[1]) memcpy(s.wins.data(), s.win_out, (size_t)s.nshares * sizeof(G1Xyzz));
      }
    }
    // 6: host combine and normalise: ONE inversion for every result of the batch (an inversion per job was ~10 us each, and a
    // job's worth of wake-up for the pool when no job needed it)
    msm_sub_combine(subs.back(), sums_all);
    std::vector<host::G1Affine> aff(2 * nj);
    host::g1_batch_to_affine(sums_all.data() + 2 * base, 2 * nj, aff.data());
    for (size_t j = 0; j < nj; j++) {
      memcpy(&out_host[base + j], &aff[2 * j], sizeof(G1Affine));
      if (jobs[base + j].pack_shift) memcpy(jobs[base + j].out_second, &aff[2 * j + 1], sizeof(G1Affine));
    }
    c.host_stamp("msm:combined");
  }
  // (a batch without entries still owes the caller its host work)
  if (!overlap_done) (*overlap)();
  return waited;
}

// ------------------------------------------------------------------ window tables (MsmJob::win_table)
// out[w * n + i] = 2^(c w) * bases[i], affine, w < W.  Built once per SRS level (mkzg.cpp srs_window_table): with the
// multiples at hand the W windows of a full-width scalar file into ONE bucket set - the bucket reduction, the window sums
// and the host's doublings of that job shrink W-fold, the additions of the accumulation stay what they were.
__global__ __launch_bounds__(128) void msm_window_table_kernel(const G1Affine* __restrict__ bases, size_t n, uint32_t cbits,
                                                               uint32_t W, G1Affine* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const G1Affine b = bases[i];
    out[i] = b;
    G1Xyzz p = G1Xyzz::from_affine(b);
    for (uint32_t w = 1; w < W; w++) {
      for (uint32_t k = 0; k < cbits; k++) p = dbl(p);
      G1Affine r;
      if (p.is_identity()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
      } else {
        const Fq i2 = inv(mul(p.zz, p.zzz));
        r.x = mul(p.x, mul(i2, p.zzz));
        r.y = mul(p.y, mul(i2, p.zz));
      }
      out[(size_t)w * n + i] = r;
      p = G1Xyzz::from_affine(r);  // (keeps the next doublings on a normalised point: zz = zzz = 1)
    }
  }
}

uint32_t msm_window_bits(size_t n) { return pick_window(n ? n : 1, 254); }

void k_msm_window_table(Ctx& c, const G1Affine* bases, size_t n, uint32_t cbits, uint32_t W, G1Affine* out) {
  if (!n || !W) return;
  hipLaunchKernelGGL(msm_window_table_kernel, dim3((unsigned)std::min<size_t>((n + 127) / 128, 1 << 16)), dim3(128), 0,
                     c.stream, bases, n, cbits, W, out);
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
