// Element-wise / streaming kernels over tables of BN254-Fr (gfx950).
// All are HBM- or integer-ALU-bound streams: 256-thread blocks, grid-stride, 16-B vector
// accesses (an Fr is two dwordx4), no LDS except for block reductions.
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include "dev.hpp"
#include "ff_host.hpp"
#include "reduce.cuh"
#include "resident.cuh"

namespace lh {

static inline dim3 grid_for(size_t n, int block = 256, size_t cap = 4096) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return dim3((unsigned)g);
}
#define GSTRIDE(i, n) \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ conversions
__global__ void fr_from_u64_kernel(const uint64_t* __restrict__ in, size_t n, Fr* __restrict__ out) {
  GSTRIDE(i, n) out[i] = from_u64<FrParams>(in[i]);
}
__global__ void fr_from_u32_kernel(const uint32_t* __restrict__ in, size_t n, Fr* __restrict__ out) {
  GSTRIDE(i, n) out[i] = from_u64<FrParams>(in[i]);
}
__global__ void fr_to_repr_kernel(const Fr* __restrict__ in, size_t n, Fr* __restrict__ out) {
  GSTRIDE(i, n) out[i] = from_mont(in[i]);
}
__global__ void fr_from_repr_kernel(const Fr* __restrict__ in, size_t n, Fr* __restrict__ out) {
  GSTRIDE(i, n) out[i] = to_mont(in[i]);
}
void k_fr_from_u64(Ctx& c, const uint64_t* in, size_t n, Fr* out) {
  if (n) hipLaunchKernelGGL(fr_from_u64_kernel, grid_for(n), 256, 0, c.stream, in, n, out);
}
void k_fr_from_u32(Ctx& c, const uint32_t* in, size_t n, Fr* out) {
  ProfScope ps(c, "fr_from_u32", 36.0 * n, 1.0 * n, (double)n);
  if (n) hipLaunchKernelGGL(fr_from_u32_kernel, grid_for(n), 256, 0, c.stream, in, n, out);
}
void k_fr_to_repr(Ctx& c, const Fr* in, size_t n, Fr* out) {
  if (n) hipLaunchKernelGGL(fr_to_repr_kernel, grid_for(n), 256, 0, c.stream, in, n, out);
}
void k_fr_from_repr(Ctx& c, const Fr* in, size_t n, Fr* out) {
  if (n) hipLaunchKernelGGL(fr_from_repr_kernel, grid_for(n), 256, 0, c.stream, in, n, out);
}

// ------------------------------------------------------------------ vector ops
template <int OP>
__global__ void fr_binop_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t n, Fr* __restrict__ out) {
  GSTRIDE(i, n) {
    Fr x = a[i], y = b[i];
    out[i] = OP == 0 ? add(x, y) : OP == 1 ? sub(x, y) : mul(x, y);
  }
}
void k_fr_binop(Ctx& c, int op, const Fr* a, const Fr* b, size_t n, Fr* out) {
  if (!n) return;
  if (op == 0) hipLaunchKernelGGL(fr_binop_kernel<0>, grid_for(n), 256, 0, c.stream, a, b, n, out);
  else if (op == 1) hipLaunchKernelGGL(fr_binop_kernel<1>, grid_for(n), 256, 0, c.stream, a, b, n, out);
  else hipLaunchKernelGGL(fr_binop_kernel<2>, grid_for(n), 256, 0, c.stream, a, b, n, out);
}

__global__ void fr_mul_chain_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t n, int iters,
                                    Fr* __restrict__ out) {
  // two independent chains per thread (elements i and i + half): the measured rate is the multiplier's, not the
  // latency of one dependent chain at whatever occupancy the launch reaches
  const size_t half = (n + 1) / 2;
  GSTRIDE(i, half) {
    const size_t j = i + half;
    const bool two = j < n;
    Fr x0 = a[i], y0 = b[i], x1 = two ? a[j] : x0, y1 = two ? b[j] : y0;
    for (int k = 0; k < iters; k++) {
      x0 = mul(x0, y0);
      x1 = mul(x1, y1);
    }
    out[i] = x0;
    if (two) out[j] = x1;
  }
}
void k_fr_mul_chain(Ctx& c, const Fr* a, const Fr* b, size_t n, int iters, Fr* out) {
  if (n) hipLaunchKernelGGL(fr_mul_chain_kernel, grid_for((n + 1) / 2, 256, 1 << 20), 256, 0, c.stream, a, b, n, iters, out);
}

// Batch inversion, Montgomery's trick per thread over a strip of CHUNK elements:
// CHUNK-1 + 3*(CHUNK-1) multiplications and ONE Fermat inversion (~380 mul) per strip.
constexpr int INV_CHUNK = 32;
__global__ void fr_batch_invert_kernel(const Fr* __restrict__ in, size_t n, Fr* __restrict__ out) {
  size_t strips = (n + INV_CHUNK - 1) / INV_CHUNK;
  GSTRIDE(s, strips) {
    size_t lo = s * INV_CHUNK, hi = lo + INV_CHUNK < n ? lo + INV_CHUNK : n;
    // forward: out[i] = product of the non-zero inputs before i
    Fr acc = Fr::one();
    for (size_t i = lo; i < hi; i++) {
      out[i] = acc;
      Fr v = in[i];
      if (!v.is_zero()) acc = mul(acc, v);
    }
    Fr iv = inv(acc);
    for (size_t i = hi; i-- > lo;) {
      Fr v = in[i];
      if (v.is_zero()) {
        out[i] = Fr::zero();
      } else {
        Fr p = out[i];
        out[i] = mul(iv, p);
        iv = mul(iv, v);
      }
    }
  }
}
void k_fr_batch_invert(Ctx& c, const Fr* in, size_t n, Fr* out) {
  if (!n) return;
  size_t strips = (n + INV_CHUNK - 1) / INV_CHUNK;
  hipLaunchKernelGGL(fr_batch_invert_kernel, grid_for(strips, 64), 64, 0, c.stream, in, n, out);
}

// ------------------------------------------------------------------ bind (fix_var)
// reference poly/multilinear.rs:599-618 `merge_into`: out[b] = e[2b] + (e[2b+1]-e[2b]) * x
__global__ void fix_var_kernel(const Fr* __restrict__ in, size_t n_out, Fr x, Fr* __restrict__ out) {
  GSTRIDE(b, n_out) {
    Fr e0 = in[2 * b], e1 = in[2 * b + 1];
    out[b] = add(mul(sub(e1, e0), x), e0);
  }
}
void k_fix_var(Ctx& c, const Fr* in, size_t n_in, const Fr& x, Fr* out) {
  ProfScope ps(c, "fix_var", 96.0 * (n_in >> 1), 1.0 * (n_in >> 1), (double)(n_in >> 1));
  size_t n_out = n_in >> 1;
  if (n_out) hipLaunchKernelGGL(fix_var_kernel, grid_for(n_out), 256, 0, c.stream, in, n_out, x, out);
}

struct PtrPack {
  const Fr* in[SC_MAX_TABLES];
  Fr* out[SC_MAX_TABLES];
};
__global__ void fix_var_multi_kernel(PtrPack p, size_t n_out, Fr x) {
  const Fr* __restrict__ in = p.in[blockIdx.y];
  Fr* __restrict__ out = p.out[blockIdx.y];
  GSTRIDE(b, n_out) {
    Fr e0 = in[2 * b], e1 = in[2 * b + 1];
    out[b] = add(mul(sub(e1, e0), x), e0);
  }
}
void k_fix_var_multi(Ctx& c, const Fr* const* in, Fr* const* out, size_t count, size_t n_in, const Fr& x) {
  ProfScope ps(c, "fix_var_multi", 96.0 * (n_in >> 1) * count, 1.0 * (n_in >> 1) * count, (double)(n_in >> 1) * count);
  size_t n_out = n_in >> 1;
  if (!n_out) return;
  for (size_t base = 0; base < count; base += SC_MAX_TABLES) {
    size_t k = count - base < (size_t)SC_MAX_TABLES ? count - base : (size_t)SC_MAX_TABLES;
    PtrPack p;
    for (size_t i = 0; i < k; i++) {
      p.in[i] = in[base + i];
      p.out[i] = out[base + i];
    }
    dim3 g = grid_for(n_out);
    g.y = (unsigned)k;
    hipLaunchKernelGGL(fix_var_multi_kernel, g, 256, 0, c.stream, p, n_out, x);
  }
}

// last bind of a sum-check (2 -> 1 entries) for `count` tables, results straight to (pinned) host memory
struct FirstPack {
  const Fr* in[SC_MAX_TABLES];
};
__global__ void bind_first_kernel(FirstPack p, int count, Fr x, Fr* __restrict__ out, uint32_t* flag, uint32_t seq) {
  int i = threadIdx.x;
  if (i < count) {
    Fr e0 = p.in[i][0], e1 = p.in[i][1];
    out[i] = add(mul(sub(e1, e0), x), e0);
  }
  // one wave: every lane's stores are ordered before lane 0's release by the system-scope fence
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __builtin_amdgcn_s_barrier();
  if (i == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void k_bind_first(Ctx& c, const Fr* const* in, size_t count, const Fr& x, Fr* out_host) {
  LH_REQUIRE(count <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "bind_first: too many tables");
  if (!count) return;
  FirstPack p;
  for (size_t i = 0; i < count; i++) p.in[i] = in[i];
  const uint32_t seq = c.next_seq();
  hipLaunchKernelGGL(bind_first_kernel, dim3(1), dim3(64), 0, c.stream, p, (int)count, x, out_host, c.flag, seq);
  c.wait_flag(seq);
}
// out[i] = first `k` entries of each of `count` tables, to (pinned) host memory
__global__ void gather_heads_kernel(FirstPack p, int count, int k, Fr* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count * k) out[i] = p.in[i / k][i % k];
}
void k_gather_heads(Ctx& c, const Fr* const* in, size_t count, int k, Fr* out_host) {
  LH_REQUIRE(count <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "gather_heads: too many tables");
  if (!count) return;
  FirstPack p;
  for (size_t i = 0; i < count; i++) p.in[i] = in[i];
  int total = (int)count * k;
  hipLaunchKernelGGL(gather_heads_kernel, dim3((total + 63) / 64), dim3(64), 0, c.stream, p, (int)count, k, out_host);
  c.sync();
}

// ------------------------------------------------------------------ eq_xy
// reference poly/multilinear.rs:91-127.  Level expansion: next[2k+1] = cur[k]*y, next[2k] = cur[k]-next[2k+1].
// In place from the top: processing k descending inside a level would race across blocks, so each
// level goes out of place between two buffers; the last level lands in `out`.
__global__ void eq_expand_kernel(const Fr* __restrict__ cur, size_t n_cur, Fr y, Fr* __restrict__ nxt) {
  GSTRIDE(k, n_cur) {
    Fr e = cur[k];
    Fr hi = mul(e, y);
    nxt[2 * k + 1] = hi;
    nxt[2 * k] = sub(e, hi);
  }
}
__global__ void fr_set_one_kernel(Fr* p) { p[0] = Fr::one(); }

// first EQ_SMALL levels of an eq table in one workgroup (LDS ping-pong): the GKR calls eq_xy once per layer
// and most layers are tiny, so one launch instead of num_vars launches is what matters there.
constexpr int EQ_SMALL = 9;
// out[(hi << lo_bits) | lo] = hi_tab[hi] * lo_tab[lo]: eq tables factor over disjoint variable sets
__global__ void eq_outer_kernel(const Fr* __restrict__ lo_tab, int lo_bits, const Fr* __restrict__ hi_tab, size_t n,
                                Fr* __restrict__ out) {
  const size_t mask = ((size_t)1 << lo_bits) - 1;
  GSTRIDE(i, n) out[i] = mul(hi_tab[i >> lo_bits], lo_tab[i & mask]);
}

// every group of <= 9 variables in ONE launch, one workgroup per group
constexpr int EQ_GROUPS = 4;
struct EqGroups {
  Fr y[EQ_GROUPS][EQ_SMALL];  // per group, expanded last-to-first
  int levels[EQ_GROUPS];
  Fr* out[EQ_GROUPS];
};
__global__ __launch_bounds__(256) void eq_groups_kernel(EqGroups g) {
  __shared__ Fr a[1 << EQ_SMALL];
  __shared__ Fr b[1 << (EQ_SMALL - 1)];
  const int levels = g.levels[blockIdx.x];
  Fr* cur = (levels & 1) ? b : a;
  Fr* nxt = (levels & 1) ? a : b;
  if (threadIdx.x == 0) cur[0] = Fr::one();
  __syncthreads();
  int n = 1;
  for (int i = 0; i < levels; i++) {
    const Fr yi = g.y[blockIdx.x][i];
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
      Fr e = cur[k];
      Fr hi = mul(e, yi);
      nxt[2 * k + 1] = hi;
      nxt[2 * k] = sub(e, hi);
    }
    __syncthreads();
    Fr* t = cur;
    cur = nxt;
    nxt = t;
    n <<= 1;
  }
  Fr* out = g.out[blockIdx.x];
  for (int k = threadIdx.x; k < n; k += blockDim.x) out[k] = cur[k];
}

// eq_xy(y) for any size: the variables are split into balanced groups of <= 9 (index bits [0,a) | [a,a+b) | ...),
// ONE launch builds every group's table (one workgroup each: a chain of <= 9 multiplications), then the table is the
// outer product of the groups (1 multiplication per entry, as the level-by-level expansion of multilinear.rs:96-121,
// but without a launch per variable).
void k_eq_xy(Ctx& c, const Fr* y, size_t num_vars, Fr* out) {
  ProfScope ps(c, "eq_xy", 64.0 * ((size_t)1 << num_vars), 1.0 * ((size_t)1 << num_vars), (double)((size_t)1 << num_vars));
  const size_t ngroups = std::max<size_t>(1, (num_vars + EQ_SMALL - 1) / EQ_SMALL);
  LH_REQUIRE(ngroups <= (size_t)EQ_GROUPS, LH_ERR_ARG, "eq_xy: too many variables");
  ArenaScope scope(c.arena);
  EqGroups g;
  size_t first[EQ_GROUPS], cnt[EQ_GROUPS];
  Fr* tab[EQ_GROUPS];
  for (size_t k = 0, done = 0; k < ngroups; k++) {
    cnt[k] = (num_vars - done + (ngroups - k) - 1) / (ngroups - k);  // balanced, larger groups first
    first[k] = done;
    done += cnt[k];
    tab[k] = ngroups == 1 ? out : c.arena.alloc_n<Fr>((size_t)1 << cnt[k]);
    g.levels[k] = (int)cnt[k];
    g.out[k] = tab[k];
    for (size_t i = 0; i < cnt[k]; i++) g.y[k][i] = y[first[k] + cnt[k] - 1 - i];  // expanded last-to-first (multilinear.rs:103)
  }
  hipLaunchKernelGGL(eq_groups_kernel, dim3((unsigned)ngroups), dim3(256), 0, c.stream, g);
  Fr* cur = tab[0];
  size_t done = cnt[0];
  for (size_t k = 1; k < ngroups; k++) {
    const bool last = k + 1 == ngroups;
    const size_t n = (size_t)1 << (done + cnt[k]);
    Fr* dst = last ? out : c.arena.alloc_n<Fr>(n);
    hipLaunchKernelGGL(eq_outer_kernel, grid_for(n), 256, 0, c.stream, cur, (int)done, tab[k], n, dst);
    cur = dst;
    done += cnt[k];
  }
}

// ------------------------------------------------------------------ sharding helpers
template <class T>
__global__ void shard_extract_kernel(const T* __restrict__ g, size_t n_local, unsigned j, unsigned rho, size_t s,
                                     T* __restrict__ local) {
  const size_t lo_mask = ((size_t)1 << j) - 1;
  GSTRIDE(i, n_local) local[i] = g[((i >> j) << (j + rho)) | (s << j) | (i & lo_mask)];
}
struct alignas(16) Bytes64 {
  uint4 a, b, c, d;
};
void k_shard_extract(Ctx& c, const void* global, size_t n_local, size_t j, size_t rho, size_t s, size_t elem,
                     void* local) {
  if (!n_local) return;
  dim3 g = grid_for(n_local);
  if (elem == 4)
    hipLaunchKernelGGL(shard_extract_kernel<uint32_t>, g, 256, 0, c.stream, (const uint32_t*)global, n_local, (unsigned)j,
                       (unsigned)rho, s, (uint32_t*)local);
  else if (elem == 32)
    hipLaunchKernelGGL(shard_extract_kernel<Fr>, g, 256, 0, c.stream, (const Fr*)global, n_local, (unsigned)j,
                       (unsigned)rho, s, (Fr*)local);
  else if (elem == 64)
    hipLaunchKernelGGL(shard_extract_kernel<Bytes64>, g, 256, 0, c.stream, (const Bytes64*)global, n_local, (unsigned)j,
                       (unsigned)rho, s, (Bytes64*)local);
  else
    throw Error(LH_ERR_ARG, "shard_extract: unsupported element size");
}
// inverse of shard_extract over the all-gathered shards (rank-major): global index (hi, s, lo) <- gathered[s][hi || lo]
template <class T>
__global__ void shard_merge_kernel(const T* __restrict__ gathered, size_t n_local, unsigned j, unsigned rho,
                                   T* __restrict__ global) {
  const size_t lo_mask = ((size_t)1 << j) - 1, s_mask = ((size_t)1 << rho) - 1;
  GSTRIDE(g, n_local << rho) {
    const size_t s = (g >> j) & s_mask, local = ((g >> (j + rho)) << j) | (g & lo_mask);
    global[g] = gathered[s * n_local + local];
  }
}
void k_shard_merge(Ctx& c, const void* gathered, size_t n_local, size_t j, size_t rho, size_t elem, void* global) {
  if (!n_local) return;
  dim3 g = grid_for(n_local << rho);
  if (elem == 4)
    hipLaunchKernelGGL(shard_merge_kernel<uint32_t>, g, 256, 0, c.stream, (const uint32_t*)gathered, n_local, (unsigned)j,
                       (unsigned)rho, (uint32_t*)global);
  else if (elem == 32)
    hipLaunchKernelGGL(shard_merge_kernel<Fr>, g, 256, 0, c.stream, (const Fr*)gathered, n_local, (unsigned)j,
                       (unsigned)rho, (Fr*)global);
  else
    throw Error(LH_ERR_ARG, "shard_merge: unsupported element size");
}

// residual tables of a sharded sum-check, gathered rank-major: with `block` = 2^(shard_bit - rounds bound) entries of
// every rank still contiguous,  out[t][(hi * R + s) * block + lo] = gathered[(s * count + t) * n_local + hi * block + lo]
// (block = 1: the shard bits have reached bit 0; block = n_local: the shard bits are the top bits, a plain concatenation)
struct InterleaveOut {
  Fr* out[SC_MAX_TABLES];
};
__global__ void gather_interleave_kernel(const Fr* __restrict__ gathered, unsigned count, size_t n_local, unsigned R,
                                         size_t block, InterleaveOut o) {
  const size_t full = n_local * R;
  GSTRIDE(e, full * count) {
    const size_t t = e / full, k = e % full, lo = k % block, q = k / block, hi = q / R, s = q % R;
    o.out[t][k] = gathered[((size_t)s * count + t) * n_local + hi * block + lo];
  }
}
void k_gather_interleave(Ctx& c, const Fr* gathered, size_t count, size_t n_local, size_t R, size_t block, Fr* const* out) {
  LH_REQUIRE(count <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "gather_interleave: too many tables");
  LH_REQUIRE(block >= 1 && n_local % block == 0, LH_ERR_ARG, "gather_interleave: bad block size");
  if (!count || !n_local) return;
  InterleaveOut o;
  for (size_t t = 0; t < count; t++) o.out[t] = out[t];
  hipLaunchKernelGGL(gather_interleave_kernel, grid_for(n_local * R * count), 256, 0, c.stream, gathered, (unsigned)count,
                     n_local, (unsigned)R, block, o);
}

// closing step of a sharded sum-check round: the D partial sums of the R ranks (all-gathered, rank-major) are added
// and published to the host exactly as a single-GPU round kernel publishes its result (pinned buffer, then the flag)
__global__ void sum_publish_kernel(const Fr* __restrict__ all, unsigned R, unsigned D, Fr* __restrict__ out_host,
                                   uint32_t* flag, uint32_t seq) {
  if (threadIdx.x == 0) {
    for (unsigned x = 0; x < D; x++) {
      Fr acc = all[x];
      for (unsigned s = 1; s < R; s++) acc = add(acc, all[(size_t)s * D + x]);
      out_host[x] = acc;
    }
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void k_sum_publish(Ctx& c, const Fr* all, size_t R, size_t D, Fr* out_host, uint32_t seq) {
  hipLaunchKernelGGL(sum_publish_kernel, dim3(1), dim3(64), 0, c.stream, all, (unsigned)R, (unsigned)D, out_host, c.flag, seq);
}

// the loopback communicator's all-gather (comm.cpp comm_attach_loopback: every peer is a copy of this rank) as ONE launch:
// block s of the result is the send buffer rotated by `rot` * s bytes (whole field elements; rot = 0: plain copies).
// (R to 2 R hipMemcpyAsync blits per collective cost a rank of an 8-rank world 4.5 ms per 2^24 AND proof - a tenth of a
// real job's collectives would have to be that slow for the emulation to be honest.)
template <class T>
__global__ void loopback_gather_kernel(const T* __restrict__ send, size_t n, size_t rot, unsigned R, T* __restrict__ recv) {
  GSTRIDE(e, n * R) {
    const size_t s = e / n, i = e % n;
    size_t k = i + rot * s;
    if (k >= n) k -= n;
    recv[e] = send[k];
  }
}
void k_loopback_gather(Ctx& c, const void* d_send, void* d_recv, size_t bytes, size_t R) {
  if (!bytes || !R) return;
  const bool vec = bytes % 16 == 0 && ((uintptr_t)d_send | (uintptr_t)d_recv) % 16 == 0;
  const size_t rot_bytes = bytes % 32 == 0 && bytes >= 64 * R ? 32 : 0;
  if (vec)
    hipLaunchKernelGGL(loopback_gather_kernel<uint4>, grid_for(bytes / 16 * R), 256, 0, c.stream, (const uint4*)d_send, bytes / 16,
                       rot_bytes / 16, (unsigned)R, (uint4*)d_recv);
  else
    hipLaunchKernelGGL(loopback_gather_kernel<uint8_t>, grid_for(bytes * R), 256, 0, c.stream, (const uint8_t*)d_send, bytes,
                       rot_bytes, (unsigned)R, (uint8_t*)d_recv);
}

// the loopback communicator's all-reduce of u64 lanes (comm.cpp comm_sum_lanes): the sum over R copies of this rank
__global__ void loopback_allreduce_lanes_kernel(const uint64_t* __restrict__ in, unsigned n, uint64_t R, uint64_t* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __hip_atomic_store(&out[i], in[i] * R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (out may be host memory)
}
void k_loopback_allreduce_lanes(Ctx& c, const uint64_t* d_in, size_t n, size_t R, uint64_t* out) {
  if (!n) return;
  hipLaunchKernelGGL(loopback_allreduce_lanes_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, c.stream, d_in, (unsigned)n,
                     (uint64_t)R, out);
}

__global__ void scale_kernel(const Fr* __restrict__ in, Fr w, size_t n, Fr* __restrict__ out) {
  GSTRIDE(i, n) out[i] = mul(in[i], w);
}
void k_scale(Ctx& c, const Fr* in, const Fr& w, size_t n, Fr* out) {
  if (n) hipLaunchKernelGGL(scale_kernel, grid_for(n), 256, 0, c.stream, in, w, n, out);
}

// ------------------------------------------------------------------ linear combination
constexpr int LC_MAX = 32;
struct LcPack {
  const Fr* p[LC_MAX];
  Fr w[LC_MAX];
};
__global__ void lincomb_kernel(LcPack pk, int count, size_t n, Fr* __restrict__ out, int accumulate) {
  GSTRIDE(i, n) {
    Fr acc = accumulate ? out[i] : Fr::zero();
    for (int k = 0; k < count; k++) acc = add(acc, mul(pk.p[k][i], pk.w[k]));
    out[i] = acc;
  }
}
// first step of an mKZG opening of g = sum_k w_k p_k without forming g: out[b] = lo + x (hi - lo) with lo = g[b],
// hi = g[b + half] (the quotient hi - lo itself is committed another way: mkzg.cpp mkzg_open, SmallOpen)
__global__ void lincomb_fold_kernel(LcPack pk, int count, size_t half, Fr x, Fr* __restrict__ out) {
  GSTRIDE(i, half) {
    Fr lo = Fr::zero(), hi = Fr::zero();
    for (int k = 0; k < count; k++) {
      lo = add(lo, mul(pk.p[k][i], pk.w[k]));
      hi = add(hi, mul(pk.p[k][half + i], pk.w[k]));
    }
    out[i] = add(lo, mul(sub(hi, lo), x));
  }
}
void k_lincomb_fold(Ctx& c, const Fr* const* polys, const Fr* w, size_t count, size_t half, const Fr& x, Fr* out) {
  LH_REQUIRE(count >= 1 && count <= (size_t)LC_MAX, LH_ERR_ARG, "lincomb_fold: bad input count");
  ProfScope ps(c, "lincomb", 32.0 * half * (2 * count + 1), (2.0 * count + 1.0) * half, (double)half);
  if (!half) return;
  LcPack pk;
  for (size_t i = 0; i < count; i++) pk.p[i] = polys[i], pk.w[i] = w[i];
  hipLaunchKernelGGL(lincomb_fold_kernel, grid_for(half), 256, 0, c.stream, pk, (int)count, half, x, out);
}
void k_lincomb(Ctx& c, const Fr* const* polys, const Fr* w, size_t count, size_t n, Fr* out) {
  ProfScope ps(c, "lincomb", 32.0 * n * (count + 1), 1.0 * n * count, (double)n);
  if (!n) return;
  if (count == 0) {
    LH_HIP(hipMemsetAsync(out, 0, n * sizeof(Fr), c.stream));
    return;
  }
  for (size_t base = 0; base < count; base += LC_MAX) {
    int k = (int)(count - base < (size_t)LC_MAX ? count - base : (size_t)LC_MAX);
    LcPack pk;
    for (int i = 0; i < k; i++) {
      pk.p[i] = polys[base + i];
      pk.w[i] = w[base + i];
    }
    hipLaunchKernelGGL(lincomb_kernel, grid_for(n), 256, 0, c.stream, pk, k, n, out, base ? 1 : 0);
  }
}

// ------------------------------------------------------------------ inner products <poly_i, weights>
constexpr int IP_MAX = 16;
struct IpPack {
  const void* p[IP_MAX];
};
template <bool U32>
__global__ void inner_products_kernel(IpPack pk, const Fr* __restrict__ w, size_t n, Fr* __restrict__ partials) {
  __shared__ Fr lds[4];
  const void* src = pk.p[blockIdx.y];
  Fr acc = Fr::zero();
  GSTRIDE(i, n) {
    Fr v;
    if (U32) {
      // small value * weight: weights are Montgomery, v canonical small -> mul(to_mont(v), w)
      v = from_u64<FrParams>(((const uint32_t*)src)[i]);
    } else {
      v = ((const Fr*)src)[i];
    }
    acc = add(acc, mul(v, w[i]));
  }
  acc = block_reduce_sum(acc, lds);
  if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}
__global__ void reduce_rows_kernel(const Fr* __restrict__ partials, int per_row, Fr* __restrict__ out) {
  __shared__ Fr lds[4];
  Fr acc = Fr::zero();
  for (int i = threadIdx.x; i < per_row; i += blockDim.x) acc = add(acc, partials[(size_t)blockIdx.x * per_row + i]);
  acc = block_reduce_sum(acc, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
template <bool U32>
static void inner_products_impl(Ctx& c, const void* const* polys, size_t count, const Fr* weights, size_t n,
                                Fr* out_host) {
  ProfScope ps(c, "inner_products", 32.0 * n * (count + 1), 1.0 * n * count, (double)n);
  if (!count) return;
  ArenaScope scope(c.arena);
  dim3 g = grid_for(n, 256, 1024);
  Fr* partials = c.arena.alloc_n<Fr>((size_t)g.x * IP_MAX);
  Fr* d_out = c.arena.alloc_n<Fr>(count);
  for (size_t base = 0; base < count; base += IP_MAX) {
    int k = (int)(count - base < (size_t)IP_MAX ? count - base : (size_t)IP_MAX);
    IpPack pk;
    for (int i = 0; i < k; i++) pk.p[i] = polys[base + i];
    dim3 gg = g;
    gg.y = k;
    hipLaunchKernelGGL(inner_products_kernel<U32>, gg, 256, 0, c.stream, pk, weights, n, partials);
    hipLaunchKernelGGL(reduce_rows_kernel, k, 256, 0, c.stream, partials, (int)g.x, d_out + base);
  }
  c.d2h(out_host, d_out, count * sizeof(Fr));
}
void k_inner_products(Ctx& c, const Fr* const* polys, size_t count, const Fr* weights, size_t n, Fr* out_host) {
  inner_products_impl<false>(c, (const void* const*)polys, count, weights, n, out_host);
}
void k_inner_products_u32(Ctx& c, const uint32_t* const* polys, size_t count, const Fr* weights, size_t n,
                          Fr* out_host) {
  inner_products_impl<true>(c, (const void* const*)polys, count, weights, n, out_host);
}

// ------------------------------------------------------------------ small-valued columns (Lasso's dim / read_ts / E / final_cts)
// A Montgomery residue W = w R mod r times a 32-bit integer v is 8 multiply-adds into a 10-limb integer accumulator
// (no reduction); a sum T of such products is reduced ONCE:  T mod r = mont(T_lo, R mod r) + (T_hi R mod r)  - the
// Montgomery product with the residue of one reduces any 256-bit integer, the high limbs re-enter as Fr::from(T_hi).
// A term costs 8 v_mad_u64_u32 instead of the 129 + 129 of from_u64 followed by mul.
struct Wide {
  uint32_t l[10];
  __device__ __forceinline__ static Wide zero() {
    Wide w;
#pragma unroll
    for (int k = 0; k < 10; k++) w.l[k] = 0;
    return w;
  }
};
__device__ __forceinline__ void wide_mac(Wide& acc, const Fr& w, uint32_t v) {
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint64_t t = (uint64_t)w.l[k] * v + acc.l[k] + carry;  // < 2^64: (2^32-1)^2 + 2 (2^32-1)
    acc.l[k] = (uint32_t)t;
    carry = t >> 32;
  }
  const uint64_t t = (uint64_t)acc.l[8] + carry;
  acc.l[8] = (uint32_t)t;
  acc.l[9] += (uint32_t)(t >> 32);
}
__device__ __forceinline__ Fr wide_reduce(const Wide& acc) {
  Fr lo;
#pragma unroll
  for (int k = 0; k < 8; k++) lo.l[k] = acc.l[k];
  const Fr one = from_u64<FrParams>(1);  // R mod r
  const uint64_t hi = (uint64_t)acc.l[8] | ((uint64_t)acc.l[9] << 32);
  Fr r = mul(lo, one);  // lo < 2^256, one < r: the product-scanning multiplication stays below 2 r (ff.cuh)
  if (hi) r = add(r, from_u64<FrParams>(hi));
  return r;
}

// The same sum with weights given TIMES R (w R^2 in memory, prescale_r on the host): one Montgomery REDUCTION of the
// 10-limb accumulator (72 multiply-adds) returns sum_k w_k v_k in Montgomery form, instead of the two full
// multiplications of wide_reduce (258).  acc < 2^320, so the result is below 2^64 + r < 2 r: one conditional subtraction.
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
// host: w (Montgomery form) -> w R (Montgomery form), the weight wide_redc expects
static Fr prescale_r(const Fr& w) {
  static const host::Fr scale = [] {
    uint64_t c[4];
    for (int k = 0; k < 4; k++) c[k] = (uint64_t)FrParams::r1(2 * k) | ((uint64_t)FrParams::r1(2 * k + 1) << 32);
    return host::Fr::from_canonical(c);  // the field element R mod r
  }();
  host::Fr h;
  memcpy(&h, &w, sizeof(h));
  h = h * scale;
  Fr out;
  memcpy(&out, &h, sizeof(out));
  return out;
}

// out[i] = sum_k w_k fr_k[i] + sum_k w'_k u32_k[i]  (u32 columns of their own lengths, zero beyond)
constexpr int LCM_MAX_FR = 8, LCM_MAX_SMALL = 24;
struct LcMixed {
  const Fr* fr[LCM_MAX_FR];
  Fr wfr[LCM_MAX_FR];
  const uint32_t* sm[LCM_MAX_SMALL];
  uint64_t sm_len[LCM_MAX_SMALL];
  Fr wsm[LCM_MAX_SMALL];
  int num_fr, num_sm;
};
__global__ void lincomb_mixed_kernel(const LcMixed* __restrict__ pkp, size_t n, Fr* __restrict__ out) {
  const LcMixed& pk = *pkp;
  GSTRIDE(i, n) {
    Wide t = Wide::zero();
    for (int k = 0; k < pk.num_sm; k++)
      if (i < pk.sm_len[k]) wide_mac(t, pk.wsm[k], pk.sm[k][i]);
    Fr acc = pk.num_sm ? wide_redc(t) : Fr::zero();  // (wsm holds the weights times R)
    for (int k = 0; k < pk.num_fr; k++) acc = add(acc, mul(pk.fr[k][i], pk.wfr[k]));
    out[i] = acc;
  }
}
void k_lincomb_mixed(Ctx& c, const Fr* const* fr, const Fr* wfr, size_t num_fr, const uint32_t* const* sm,
                     const size_t* sm_len, const Fr* wsm, size_t num_sm, size_t n, Fr* out) {
  LH_REQUIRE(num_fr <= (size_t)LCM_MAX_FR && num_sm <= (size_t)LCM_MAX_SMALL, LH_ERR_ARG, "lincomb_mixed: too many inputs");
  ProfScope ps(c, "lincomb", 32.0 * n * (num_fr + 1) + 4.0 * n * num_sm, 1.0 * n * num_fr + 0.07 * n * num_sm + 2.0 * n, (double)n);
  if (!n) return;
  LcMixed pk;
  memset(&pk, 0, sizeof(pk));
  pk.num_fr = (int)num_fr, pk.num_sm = (int)num_sm;
  for (size_t k = 0; k < num_fr; k++) pk.fr[k] = fr[k], pk.wfr[k] = wfr[k];
  for (size_t k = 0; k < num_sm; k++) pk.sm[k] = sm[k], pk.sm_len[k] = sm_len[k], pk.wsm[k] = prescale_r(wsm[k]);
  ArenaScope scope(c.arena);  // the argument block exceeds the 4 KB of kernel arguments: it travels through memory
  LcMixed* d_pk = (LcMixed*)c.arena.alloc(sizeof(LcMixed));
  LcMixed* h_pk = (LcMixed*)c.pin(65536);
  memcpy(h_pk, &pk, sizeof(pk));
  LH_HIP(hipMemcpyAsync(d_pk, h_pk, sizeof(pk), hipMemcpyHostToDevice, c.stream));
  hipLaunchKernelGGL(lincomb_mixed_kernel, grid_for(n), 256, 0, c.stream, d_pk, n, out);
  c.sync();  // the pinned staging block is reused by the next caller
}

// First step of an mKZG opening of g = sum_k coef_k col_k over 32-bit columns (SmallOpen) without g or any field-element
// view: out[i] = (1 - x) g[i] + x g[i + half] is 2 x 8 multiply-adds per column into one wide accumulator and ONE Montgomery
// reduction (the weights coef_k (1 - x), coef_k x arrive times R).  Reads 8 B per column and output entry where the fold of
// the batch opening's merged tables (k_lincomb_fold) read 64 B per table: 2^24 AND lookups, 12 columns against 4 tables.
constexpr int LCF_MAX = 24;
struct LcFoldSmall {
  const uint32_t* p[LCF_MAX];
  uint64_t len[LCF_MAX];
  Fr lo[LCF_MAX], hi[LCF_MAX];
  int count;
};
__global__ __launch_bounds__(256) void lincomb_fold_small_kernel(const LcFoldSmall* __restrict__ pkp, size_t half, Fr* __restrict__ out) {
  const LcFoldSmall& pk = *pkp;
  GSTRIDE(i, half) {
    Wide t = Wide::zero();
    for (int k = 0; k < pk.count; k++) {
      if (i < pk.len[k]) wide_mac(t, pk.lo[k], pk.p[k][i]);
      if (i + half < pk.len[k]) wide_mac(t, pk.hi[k], pk.p[k][i + half]);
    }
    out[i] = wide_redc(t);
  }
}
bool k_lincomb_fold_small(Ctx& c, const uint32_t* const* cols, const size_t* lens, const Fr* coef, size_t count, size_t half,
                          const Fr& x, Fr* out) {
  if (count < 1 || count > (size_t)LCF_MAX || !half) return false;
  ProfScope ps(c, "lincomb<fold,u32>", 8.0 * half * count + 32.0 * half, 0.15 * half * count + 0.6 * half, (double)half);
  LcFoldSmall pk;
  memset(&pk, 0, sizeof(pk));
  pk.count = (int)count;
  host::Fr hx, one = host::Fr::one();
  memcpy(&hx, &x, sizeof(hx));
  for (size_t k = 0; k < count; k++) {
    host::Fr ck;
    memcpy(&ck, &coef[k], sizeof(ck));
    const host::Fr l = ck * (one - hx), h = ck * hx;
    Fr dl, dh;
    memcpy(&dl, &l, sizeof(dl));
    memcpy(&dh, &h, sizeof(dh));
    pk.p[k] = cols[k], pk.len[k] = lens[k], pk.lo[k] = prescale_r(dl), pk.hi[k] = prescale_r(dh);
  }
  ArenaScope scope(c.arena);  // (the argument block travels through memory: 24 columns are 2 KB)
  LcFoldSmall* d_pk = (LcFoldSmall*)c.arena.alloc(sizeof(LcFoldSmall));
  LcFoldSmall* h_pk = (LcFoldSmall*)c.pin(65536);
  memcpy(h_pk, &pk, sizeof(pk));
  LH_HIP(hipMemcpyAsync(d_pk, h_pk, sizeof(pk), hipMemcpyHostToDevice, c.stream));
  hipLaunchKernelGGL(lincomb_fold_small_kernel, grid_for(half), 256, 0, c.stream, d_pk, half, out);
  c.sync();  // the pinned staging block is reused by the next caller
  return true;
}

// out[k] = <u32 column k, weights> for up to IPS_GROUP columns per launch row: the weight is loaded once per entry
constexpr int IPS_GROUP = 4;  // 6 accumulators of 10 limbs spill (256 B of scratch per lane); 4 keep 4 waves per SIMD
struct IpSmallPack {
  const uint32_t* p[IPS_GROUP];
  int count;
  uint32_t stride, off[IPS_GROUP];  // entry i of pseudo-column k is p[k][i * stride + off[k]] (the even / odd halves of a column)
};
// (the group size is a template parameter: accumulators indexed by a run-time count end up in scratch memory)
template <int G>
__global__ __launch_bounds__(256) void inner_products_small_kernel(IpSmallPack pk, const Fr* __restrict__ w, size_t n,
                                                                   Fr* __restrict__ partials) {
  __shared__ Fr lds[4];
  Wide acc[G];
#pragma unroll
  for (int k = 0; k < G; k++) acc[k] = Wide::zero();
  GSTRIDE(i, n) {
    const Fr wi = w[i];
#pragma unroll
    for (int k = 0; k < G; k++) wide_mac(acc[k], wi, pk.p[k][i * pk.stride + pk.off[k]]);
  }
#pragma unroll
  for (int k = 0; k < G; k++) {
    Fr v = block_reduce_sum(wide_reduce(acc[k]), lds);
    if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = v;
  }
}
template <int G>
static void launch_ips(Ctx& c, dim3 g, const IpSmallPack& pk, const Fr* w, size_t n, Fr* partials) {
  hipLaunchKernelGGL(inner_products_small_kernel<G>, g, 256, 0, c.stream, pk, w, n, partials);
}
static void inner_products_small_strided(Ctx& c, const uint32_t* const* polys, const uint32_t* offs, size_t count,
                                         uint32_t stride, const Fr* weights, size_t n, Fr* out_host) {
  ProfScope ps(c, "inner_products", 4.0 * n * count + 32.0 * n * ((count + IPS_GROUP - 1) / IPS_GROUP), 0.07 * n * count, (double)n);
  if (!count) return;
  ArenaScope scope(c.arena);
  // a thread's 10-limb accumulator holds up to 2^34 terms of 2^286: the grid-stride share of any n < 2^31 fits
  dim3 g = grid_for(n, 256, 1024);
  Fr* partials = c.arena.alloc_n<Fr>((size_t)g.x * IPS_GROUP);
  Fr* d_out = c.arena.alloc_n<Fr>(count);
  for (size_t base = 0; base < count; base += IPS_GROUP) {
    IpSmallPack pk;
    pk.count = (int)std::min<size_t>(IPS_GROUP, count - base);
    pk.stride = stride;
    for (int i = 0; i < IPS_GROUP; i++) {
      pk.p[i] = i < pk.count ? polys[base + i] : nullptr;
      pk.off[i] = i < pk.count && offs ? offs[base + i] : 0u;
    }
    switch (pk.count) {
      case 1: launch_ips<1>(c, g, pk, weights, n, partials); break;
      case 2: launch_ips<2>(c, g, pk, weights, n, partials); break;
      case 3: launch_ips<3>(c, g, pk, weights, n, partials); break;
      default: launch_ips<4>(c, g, pk, weights, n, partials); break;
    }
    hipLaunchKernelGGL(reduce_rows_kernel, pk.count, 256, 0, c.stream, partials, (int)g.x, d_out + base);
  }
  c.d2h(out_host, d_out, count * sizeof(Fr));
}
void k_inner_products_small(Ctx& c, const uint32_t* const* polys, size_t count, const Fr* weights, size_t n,
                            Fr* out_host) {
  inner_products_small_strided(c, polys, nullptr, count, 1, weights, n, out_host);
}
// <column, eq(y)> from the eq table of y[1..] alone (half the entries): eq(y, 2b + e) = (e ? y0 : 1 - y0) * half[b], so the
// even and the odd entries of a column are two strided pseudo-columns against the same weights
void k_inner_products_small_half(Ctx& c, const uint32_t* const* polys, size_t count, const Fr* eq_half, size_t half,
                                 const Fr& y0, Fr* out_host) {
  if (!count) return;
  std::vector<const uint32_t*> p2(2 * count);
  std::vector<uint32_t> off(2 * count);
  for (size_t k = 0; k < count; k++) p2[2 * k] = p2[2 * k + 1] = polys[k], off[2 * k] = 0, off[2 * k + 1] = 1;
  std::vector<Fr> eo(2 * count);
  inner_products_small_strided(c, p2.data(), off.data(), 2 * count, 2, eq_half, half, eo.data());
  host::Fr y, one = host::Fr::one();
  memcpy(&y, &y0, sizeof(y));
  for (size_t k = 0; k < count; k++) {
    host::Fr e, o;
    memcpy(&e, &eo[2 * k], sizeof(e));
    memcpy(&o, &eo[2 * k + 1], sizeof(o));
    const host::Fr r = (one - y) * e + y * o;
    memcpy(&out_host[k], &r, sizeof(r));
  }
}

// A degree-2 eq-factored sum-check over ONE table whose values still are a 32-bit column (Surge over the output column of a
// linear g): nothing of its first two rounds needs the challenges.  Round 0 sends q(1) = sum_b E_0[b] a[2b + 1]; round 1, after
// binding r0, sends q(1) = sum_b E_1[b] ((1 - r0) a[4b + 2] + r0 a[4b + 3]) = (1 - r0) S2 + r0 S3 with E_1[b] = E_0[2b] +
// E_0[2b + 1].  One pass over the column and E_0 makes the claim's two halves (even, odd) and S2, S3: 8 multiply-adds per term
// into wide accumulators (wide_mac above).  out_host[0..3] = even, odd, S2, S3.
__global__ __launch_bounds__(256) void inner_products_small_quads_kernel(const uint32_t* __restrict__ col, const Fr* __restrict__ e0,
                                                                         size_t quads, Fr* __restrict__ partials) {
  __shared__ Fr lds[4];
  Wide acc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) acc[k] = Wide::zero();
  GSTRIDE(b, quads) {
    const uint4 a = ((const uint4*)col)[b];
    const Fr h0 = e0[2 * b], h1 = e0[2 * b + 1];
    wide_mac(acc[0], h0, a.x), wide_mac(acc[0], h1, a.z);
    wide_mac(acc[1], h0, a.y), wide_mac(acc[1], h1, a.w);
    const Fr h = add(h0, h1);
    wide_mac(acc[2], h, a.z), wide_mac(acc[3], h, a.w);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    Fr v = block_reduce_sum(wide_reduce(acc[k]), lds);
    if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = v;
  }
}
void k_inner_products_small_quads(Ctx& c, const uint32_t* col, const Fr* eq_half, size_t quads, Fr* out_host) {
  ProfScope ps(c, "inner_products<quads>", 80.0 * quads, 0.45 * quads, (double)quads);
  LH_REQUIRE(quads >= 1, LH_ERR_ARG, "inner_products_small_quads: empty column");
  ArenaScope scope(c.arena);
  dim3 g = grid_for(quads, 256, 1024);  // (a thread's accumulators: at most 2^33 terms of 2^288 each fit 10 limbs)
  Fr* partials = c.arena.alloc_n<Fr>((size_t)g.x * 4);
  Fr* d_out = c.arena.alloc_n<Fr>(4);
  hipLaunchKernelGGL(inner_products_small_quads_kernel, g, 256, 0, c.stream, col, eq_half, quads, partials);
  hipLaunchKernelGGL(reduce_rows_kernel, 4, 256, 0, c.stream, partials, (int)g.x, d_out);
  c.d2h(out_host, d_out, 4 * sizeof(Fr));
}

// The same for the batch opening's sum-check (mkzg.cpp additive_batch_open: terms eq(y_m, .) * merged_m with merged_m a
// combination of 32-bit columns): S_t = sum_q e1[q] col[4q + t], t = 0..3, of two columns per launch against ONE
// read of e1 = eq(y[2..]) (the eq levels of a term: E_1; E_0[2q + e] = eq(y_1, e) E_1[q], so the round-0 sums over E_0 are
// (1 - y_1) S_e + y_1 S_{e+2}).  Rounds 0 and 1 of a term are host combinations of its columns' S_t - no table of the
// merged polynomial is read or even written.  out[4 k + t] (device) = S_t of column k; entries beyond a column's length are zero.
constexpr int IPQ_GROUP = 2;  // 8 ten-limb accumulators, 168 registers, 3 waves per SIMD (three columns: 243 registers and spills)
struct IpQuadPack {
  const uint32_t* p[IPQ_GROUP];
  uint64_t quads[IPQ_GROUP];  // len / 4 of each column
};
template <int G>
__global__ __launch_bounds__(256) void inner_products_quads_kernel(IpQuadPack pk, const Fr* __restrict__ e1, size_t quads,
                                                                   Fr* __restrict__ partials) {
  __shared__ Fr lds[4];
  Wide acc[G][4];
#pragma unroll
  for (int k = 0; k < G; k++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[k][t] = Wide::zero();
  GSTRIDE(q, quads) {
    const Fr h = e1[q];
#pragma unroll
    for (int k = 0; k < G; k++)
      if (q < pk.quads[k]) {
        const uint4 a = ((const uint4*)pk.p[k])[q];
        wide_mac(acc[k][0], h, a.x), wide_mac(acc[k][1], h, a.y), wide_mac(acc[k][2], h, a.z), wide_mac(acc[k][3], h, a.w);
      }
  }
#pragma unroll
  for (int k = 0; k < G; k++)
#pragma unroll
    for (int t = 0; t < 4; t++) {
      Fr v = block_reduce_sum(wide_reduce(acc[k][t]), lds);
      if (threadIdx.x == 0) partials[(size_t)(4 * k + t) * gridDim.x + blockIdx.x] = v;
    }
}
void k_inner_products_quads(Ctx& c, const uint32_t* const* cols, const size_t* lens, size_t count, const Fr* e1, size_t quads,
                            Fr* d_out) {
  ProfScope ps(c, "inner_products<quads>", 4.0 * 4 * quads * count + 32.0 * quads * ((count + IPQ_GROUP - 1) / IPQ_GROUP),
               0.25 * quads * count, (double)quads);
  if (!count) return;
  // (a thread's accumulator: at most 2^32 / 1024 / 256 terms of 2^288 - far inside ten limbs)
  dim3 g = grid_for(quads, 256, 1024);
  Fr* partials = c.arena.alloc_n<Fr>((size_t)g.x * 4 * IPQ_GROUP);  // (queued work reads it: the caller's arena scope holds it)
  for (size_t base = 0; base < count; base += IPQ_GROUP) {
    IpQuadPack pk;
    const int k = (int)std::min<size_t>(IPQ_GROUP, count - base);
    size_t most = 0;
    for (int i = 0; i < IPQ_GROUP; i++) {
      pk.p[i] = i < k ? cols[base + i] : nullptr;
      pk.quads[i] = i < k ? std::min(lens[base + i] / 4, quads) : 0;
      most = std::max<size_t>(most, pk.quads[i]);
    }
    dim3 gg = grid_for(std::max<size_t>(most, 1), 256, 1024);
    if (k == 1) hipLaunchKernelGGL(inner_products_quads_kernel<1>, gg, 256, 0, c.stream, pk, e1, most, partials);
    else hipLaunchKernelGGL(inner_products_quads_kernel<2>, gg, 256, 0, c.stream, pk, e1, most, partials);
    hipLaunchKernelGGL(reduce_rows_kernel, 4 * k, 256, 0, c.stream, partials, (int)gg.x, d_out + 4 * base);
  }
}

// Round 2 of such a term: merged_m bound with (r0, r1) straight from its columns -
//   out[i] = sum_k w_k ((1-r1)(1-r0) col_k[4i] + (1-r1) r0 col_k[4i+1] + r1 (1-r0) col_k[4i+2] + r1 r0 col_k[4i+3]),
// 4 x 8 multiply-adds per column and ONE Montgomery reduction per bound entry (the 4 weights of a column arrive times R) -
// and the round's q(0), q(1) = sum_b eq_level[b] out[2b + e] from the even / odd lanes.
constexpr int LCB_MAX = 24;
struct LcBind2 {
  const uint32_t* p[LCB_MAX];
  uint64_t quads[LCB_MAX];
  Fr w[LCB_MAX][4];
  int count;
};
__global__ __launch_bounds__(256) void lincomb_bind2_kernel(const LcBind2* __restrict__ pkp, const Fr* __restrict__ eq_level,
                                                            size_t entries, Fr* __restrict__ out, Fr* __restrict__ partials,
                                                            ScFinishArgs fin) {
  __shared__ Fr lds[4];
  __shared__ int is_last;
  const LcBind2& pk = *pkp;
  const bool odd = threadIdx.x & 1;
  Fr acc = Fr::zero();
  GSTRIDE(i, entries) {
    Wide t = Wide::zero();
    for (int k = 0; k < pk.count; k++)
      if (i < pk.quads[k]) {
        const uint4 a = ((const uint4*)pk.p[k])[i];
        wide_mac(t, pk.w[k][0], a.x), wide_mac(t, pk.w[k][1], a.y), wide_mac(t, pk.w[k][2], a.z), wide_mac(t, pk.w[k][3], a.w);
      }
    const Fr v = wide_redc(t);
    out[i] = v;
    acc = add(acc, mul(v, eq_level[i >> 1]));
  }
  const Fr q0 = block_reduce_sum(odd ? Fr::zero() : acc, lds);
  const Fr q1 = block_reduce_sum(odd ? acc : Fr::zero(), lds);
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) {
      fin.out_host[0] = q0, fin.out_host[1] = q1;
      publish_round(fin, 2);
    }
    return;
  }
  if (threadIdx.x == 0) fin_put(fin, partials, (size_t)blockIdx.x * 2, q0), fin_put(fin, partials, (size_t)blockIdx.x * 2 + 1, q1);
  if (!fin_ticket(fin, &is_last)) return;
  for (int x = 0; x < 2; x++) {
    Fr a2 = Fr::zero();
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x) a2 = add(a2, fin_get(fin, partials, (size_t)i * 2 + x));
    a2 = block_reduce_sum(a2, lds);
    if (threadIdx.x == 0) fin.out_host[x] = a2;
  }
  if (threadIdx.x == 0) publish_round(fin, 2);
}
void k_lincomb_bind2(Ctx& c, const uint32_t* const* cols, const size_t* lens, const Fr* w, size_t count, const Fr& r0, const Fr& r1,
                     const Fr* eq_level, size_t size, Fr* out, Fr* out_host) {
  LH_REQUIRE(count >= 1 && count <= (size_t)LCB_MAX && size >= 1 && !c.sc_redirect, LH_ERR_ARG, "lincomb_bind2: bad shape");
  const uint32_t seq = c.next_seq();
  ArenaScope scope(c.arena);
  const size_t entries = 2 * size;
  host::Fr h0, h1;
  const host::Fr one = host::Fr::one();
  memcpy(&h0, &r0, sizeof(h0));
  memcpy(&h1, &r1, sizeof(h1));
  const host::Fr b4[4] = {(one - h1) * (one - h0), (one - h1) * h0, h1 * (one - h0), h1 * h0};
  // (the front of the pinned block is the round loop's message buffer: the argument block is staged behind it)
  LcBind2* h_pk = (LcBind2*)((char*)c.pin(65536) + 8192);
  memset(h_pk, 0, sizeof(LcBind2));
  h_pk->count = (int)count;
  for (size_t k = 0; k < count; k++) {
    host::Fr wk;
    memcpy(&wk, &w[k], sizeof(wk));
    h_pk->p[k] = cols[k], h_pk->quads[k] = std::min(lens[k] / 4, entries);
    for (int t = 0; t < 4; t++) {
      const host::Fr x = wk * b4[t];
      Fr d;
      memcpy(&d, &x, sizeof(d));
      h_pk->w[k][t] = prescale_r(d);
    }
  }
  LcBind2* d_pk = (LcBind2*)c.arena.alloc(sizeof(LcBind2));
  LH_HIP(hipMemcpyAsync(d_pk, h_pk, sizeof(LcBind2), hipMemcpyHostToDevice, c.stream));
  const size_t g = std::min<size_t>((entries + 255) / 256, (size_t)c.num_cus * 8);
  Fr* partials = c.arena.alloc_n<Fr>(g * 2);
  const ScFinishArgs fin = c.finish_for((uint32_t)g, out_host, seq, 32.0 * (double)entries);
  {
    ProfScope ps(c, "lincomb<bind2,u32>", (16.0 * count + 32.0 + 16.0) * (double)entries, (0.25 * count + 1.6) * (double)entries, (double)size);
    hipLaunchKernelGGL(lincomb_bind2_kernel, dim3((unsigned)g), dim3(256), 0, c.stream, d_pk, eq_level, entries, out, partials, fin);
  }
  c.wait_round(seq);  // (the pinned staging block is free again: the copy in front of the kernel has run)
}

// Round 2 of the same sum-check binds r0 AND r1 straight from the column: bound entry i is
//   (1 - r1) ((1 - r0) a[4i] + r0 a[4i + 1]) + r1 ((1 - r0) a[4i + 2] + r0 a[4i + 3]),
// four 8-multiply-add terms and ONE Montgomery reduction (the four weights arrive times R: prescale_r / wide_redc), read at
// 16 bytes per lane and stored like the plain bind kernel's; the round's q(1) = sum_b E_2[b] out[2b + 1] rides on the odd
// lanes.  No field-element view of the column, and no table of 2^(n-1) entries, is ever written: at 2^24 AND lookups
// fr_from_u32 0.18 + sc_round<1,first> 0.21 + sc_round<1,bind> 0.28 + 0.18 ms of launches became this one.
struct U32Bind2 {
  Fr w[4];  // (1-r1)(1-r0), (1-r1) r0, r1 (1-r0), r1 r0 - times R
};
__global__ __launch_bounds__(256) void sc_round_u32_bind2_kernel(const uint32_t* __restrict__ col, const Fr* __restrict__ eq_level,
                                                                 U32Bind2 k, size_t entries, Fr* __restrict__ out,
                                                                 Fr* __restrict__ partials, ScFinishArgs fin) {
  __shared__ Fr lds[4];
  __shared__ int is_last;
  Fr acc = Fr::zero();
  GSTRIDE(i, entries) {
    const uint4 a = ((const uint4*)col)[i];
    Wide t = Wide::zero();
    wide_mac(t, k.w[0], a.x), wide_mac(t, k.w[1], a.y), wide_mac(t, k.w[2], a.z), wide_mac(t, k.w[3], a.w);
    const Fr v = wide_redc(t);
    out[i] = v;
    if (i & 1) acc = add(acc, mul(v, eq_level[i >> 1]));
  }
  acc = block_reduce_sum(acc, lds);
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) {
      fin.out_host[0] = acc;
      publish_round(fin, 1);
    }
    return;
  }
  if (threadIdx.x == 0) fin_put(fin, partials, blockIdx.x, acc);
  if (!fin_ticket(fin, &is_last)) return;
  Fr a2 = Fr::zero();
  for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x) a2 = add(a2, fin_get(fin, partials, i));
  a2 = block_reduce_sum(a2, lds);
  if (threadIdx.x == 0) {
    fin.out_host[0] = a2;
    publish_round(fin, 1);
  }
}
void k_sc_round_u32_bind2(Ctx& c, const uint32_t* col, const Fr* eq_level, const Fr& r0, const Fr& r1, size_t size, Fr* out,
                          Fr* out_host) {
  LH_REQUIRE(size >= 1 && !c.sc_redirect, LH_ERR_ARG, "sc_round_u32_bind2: bad shape");
  const uint32_t seq = c.next_seq();
  ArenaScope scope(c.arena);
  const size_t entries = 2 * size;
  const size_t g = std::min<size_t>((entries + 255) / 256, (size_t)c.num_cus * 8);
  Fr* partials = c.arena.alloc_n<Fr>(g);
  const ScFinishArgs fin = c.finish_for((uint32_t)g, out_host, seq, 32.0 * (double)entries);
  host::Fr h0, h1;
  const host::Fr one = host::Fr::one();
  memcpy(&h0, &r0, sizeof(h0));
  memcpy(&h1, &r1, sizeof(h1));
  const host::Fr w[4] = {(one - h1) * (one - h0), (one - h1) * h0, h1 * (one - h0), h1 * h0};
  U32Bind2 k;
  for (int i = 0; i < 4; i++) {
    Fr d;
    memcpy(&d, &w[i], sizeof(d));
    k.w[i] = prescale_r(d);
  }
  {
    // per bound entry: 16 B of column, 32 B stored, 16 B of eq level; a reduction and half a product
    ProfScope ps(c, "sc_round_u32<bind2>", 64.0 * (double)entries, 1.2 * (double)entries, (double)size);
    hipLaunchKernelGGL(sc_round_u32_bind2_kernel, dim3((unsigned)g), dim3(256), 0, c.stream, col, eq_level, k, entries, out, partials, fin);
  }
  c.wait_round(seq);
}

// ------------------------------------------------------------------ GKR layer-up
// product tree (Lasso memory check): out[i] = in[i] * in[half+i]
__global__ void tree_up_kernel(const Fr* __restrict__ in, size_t half, Fr* __restrict__ out) {
  GSTRIDE(i, half) out[i] = mul(in[i], in[half + i]);
}
// the same level of several trees of equal size in one launch (a level of a small tree is launch-bound)
__global__ void tree_up_multi_kernel(PtrPack p, size_t half) {
  const Fr* __restrict__ in = p.in[blockIdx.y];
  Fr* __restrict__ out = p.out[blockIdx.y];
  GSTRIDE(i, half) out[i] = mul(in[i], in[half + i]);
}
void k_tree_up_multi(Ctx& c, const Fr* const* in, Fr* const* out, size_t count, size_t half) {
  ProfScope ps(c, "tree_up", 96.0 * half * count, 1.0 * half * count, (double)half * count);
  if (!half) return;
  for (size_t base = 0; base < count; base += SC_MAX_TABLES) {
    size_t k = count - base < (size_t)SC_MAX_TABLES ? count - base : (size_t)SC_MAX_TABLES;
    PtrPack p;
    for (size_t i = 0; i < k; i++) {
      p.in[i] = in[base + i];
      p.out[i] = out[base + i];
    }
    dim3 g = grid_for(half);
    g.y = (unsigned)k;
    hipLaunchKernelGGL(tree_up_multi_kernel, g, 256, 0, c.stream, p, half);
  }
}
void k_tree_up(Ctx& c, const Fr* in, size_t half, Fr* out) {
  ProfScope ps(c, "tree_up", 96.0 * half, 1.0 * half, (double)half);
  if (half) hipLaunchKernelGGL(tree_up_kernel, grid_for(half), 256, 0, c.stream, in, half, out);
}
// All levels above a small level of a product tree in one workgroup per tree: `in` holds 2^(H+1) nodes
// (H <= TREE_SMALL); level h < H (2^(h+1) nodes) is written at out + (2^(h+1) - 2).
constexpr int TREE_SMALL = 9;
struct TreeTops {
  const Fr* in[SC_MAX_TABLES];
  Fr* out[SC_MAX_TABLES];
  int H[SC_MAX_TABLES];
};
__global__ __launch_bounds__(256) void tree_top_kernel(TreeTops t) {
  __shared__ Fr a[1 << TREE_SMALL];
  __shared__ Fr b[1 << (TREE_SMALL - 1)];
  const Fr* __restrict__ in = t.in[blockIdx.x];
  Fr* __restrict__ out = t.out[blockIdx.x];
  const int H = t.H[blockIdx.x];
  if (H <= 0) return;
  // level H-1 straight from global memory
  int half = 1 << H;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    Fr v = mul(in[i], in[half + i]);
    a[i] = v;
    out[(half - 2) + i] = v;
  }
  __syncthreads();
  Fr* cur = a;
  Fr* nxt = b;
  for (int h = H - 2; h >= 0; h--) {
    half = 1 << (h + 1);
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
      Fr v = mul(cur[i], cur[half + i]);
      nxt[i] = v;
      out[(half - 2) + i] = v;
    }
    __syncthreads();
    Fr* tmp = cur;
    cur = nxt;
    nxt = tmp;
  }
}
void k_tree_tops(Ctx& c, const Fr* const* in, Fr* const* out, const int* H, size_t count) {
  LH_REQUIRE(count <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "tree_tops: too many trees");
  if (!count) return;
  TreeTops t;
  for (size_t i = 0; i < count; i++) {
    LH_REQUIRE(H[i] <= TREE_SMALL, LH_ERR_ARG, "tree_tops: level too large");
    t.in[i] = in[i];
    t.out[i] = out[i];
    t.H[i] = H[i];
  }
  ProfScope ps(c, "tree_tops", 0, 0, (double)count);
  hipLaunchKernelGGL(tree_top_kernel, dim3((unsigned)count), dim3(256), 0, c.stream, t);
}

// reference fractional_sum_check.rs:62-85 `Layer::up`
__global__ void frac_up_kernel(const Fr* __restrict__ p, const Fr* __restrict__ q, size_t half, Fr* __restrict__ vp,
                               Fr* __restrict__ vq) {
  GSTRIDE(i, half) {
    Fr pl = p[i], pr = p[half + i], ql = q[i], qr = q[half + i];
    vp[i] = add(mul(pl, qr), mul(pr, ql));
    vq[i] = mul(ql, qr);
  }
}
void k_frac_up(Ctx& c, const Fr* p, const Fr* q, size_t half, Fr* vp, Fr* vq) {
  ProfScope ps(c, "frac_up", 192.0 * half, 3.0 * half, (double)half);
  if (half) hipLaunchKernelGGL(frac_up_kernel, grid_for(half), 256, 0, c.stream, p, q, half, vp, vq);
}

// ------------------------------------------------------------------ KZG quotient step
// reference pcs/multilinear.rs:86-98: q = hi - lo ; lo += (hi - lo) * x_i
__global__ void quotient_step_kernel(const Fr* __restrict__ rem, size_t half, Fr x, Fr* __restrict__ q,
                                     Fr* __restrict__ rem_out) {
  GSTRIDE(i, half) {
    Fr lo = rem[i], hi = rem[half + i];
    Fr d = sub(hi, lo);
    if (q) q[i] = d;  // (null: the caller commits this quotient another way)
    rem_out[i] = add(lo, mul(d, x));
  }
}
void k_quotient_step(Ctx& c, const Fr* rem, size_t half, const Fr& x, Fr* q, Fr* rem_out) {
  ProfScope ps(c, "quotient_step", 128.0 * half, 1.0 * half, (double)half);
  if (half) hipLaunchKernelGGL(quotient_step_kernel, grid_for(half), 256, 0, c.stream, rem, half, x, q, rem_out);
}

// ------------------------------------------------------------------ Lasso witness + fingerprints
__device__ __forceinline__ uint32_t subtable_entry(int kind, uint32_t m, uint32_t bits) {
  uint32_t h = bits >> 1;
  uint32_t x = m >> h, y = m & ((1u << h) - 1u);
  return kind == LH_SUBTABLE_IDENTITY ? m : kind == LH_SUBTABLE_AND ? (x & y) : (x ^ y);
}

// read_ts[k] = number of earlier lookups of the same address; final_cts[a] = total count of address a.
// Stable radix sort of (address, lookup index) pairs, then a lookup's rank inside its run of equal addresses is
// its position minus the run start: O(n) traffic whatever the table size (a tile x address histogram, the
// obvious counting-sort formulation, moves tiles * m counters - 2 GB per column at 2^24 lookups).
// Large columns: the scatter read_ts[sidx[i]] = rank above drags a whole line through the caches for every 4 bytes.  Here
// the ranks stay in sorted order (`ranks`, coalesced), the (lookup index, rank) pairs are partitioned by the top bits of the
// index (one or two passes of the radix sort) and every window of 2^15 consecutive read_ts entries is assembled in LDS and
// written out as whole lines.
constexpr uint32_t UNPERM_WINDOW_LOG = 15;  // read_ts entries per workgroup: 128 KB of LDS
// pairs (idx, rank) grouped by idx >> group_log (group g = positions [g << group_log, (g + 1) << group_log), idx a
// permutation); workgroup (g, h) assembles window h of group g
// All chunk columns of a table per launch (blockIdx.y = column): the steps above for up to LH_LASSO_MAX_CHUNKS columns of n
// lookups into m cells each
struct CtCols {
  const uint32_t* skey[LH_LASSO_MAX_CHUNKS];
  const uint32_t* sidx[LH_LASSO_MAX_CHUNKS];
  uint32_t* start[LH_LASSO_MAX_CHUNKS];
  uint32_t* ranks[LH_LASSO_MAX_CHUNKS];
  uint32_t* read_ts[LH_LASSO_MAX_CHUNKS];
  uint32_t* final_cts[LH_LASSO_MAX_CHUNKS];
  const uint32_t* pidx[LH_LASSO_MAX_CHUNKS];
  const uint32_t* prank[LH_LASSO_MAX_CHUNKS];
};
__global__ void lasso_run_start_cols_kernel(CtCols k, size_t n, size_t m, uint32_t* __restrict__ bad) {
  const uint32_t* __restrict__ skey = k.skey[blockIdx.y];
  uint32_t* __restrict__ start = k.start[blockIdx.y];
  GSTRIDE(i, n) {
    const uint32_t a = skey[i];
    if (a >= m) *bad = 1u;  // benign race: every writer stores the same value
    else if (i == 0 || skey[i - 1] != a) start[a] = (uint32_t)i;
  }
}
__global__ void lasso_rank_cols_kernel(CtCols k, size_t n, size_t m) {
  const uint32_t* __restrict__ skey = k.skey[blockIdx.y];
  const uint32_t* __restrict__ sidx = k.sidx[blockIdx.y];
  const uint32_t* __restrict__ start = k.start[blockIdx.y];
  uint32_t* __restrict__ read_ts = k.read_ts[blockIdx.y];
  uint32_t* __restrict__ final_cts = k.final_cts[blockIdx.y];
  GSTRIDE(i, n) {
    const uint32_t a = skey[i];
    if (a >= m) continue;  // reported through `bad`
    const uint32_t r = (uint32_t)i - start[a];
    read_ts[sidx[i]] = r;
    if (i + 1 == n || skey[i + 1] != a) final_cts[a] = r + 1;
  }
}
__global__ void lasso_rank_sorted_cols_kernel(CtCols k, size_t n, size_t m) {
  const uint32_t* __restrict__ skey = k.skey[blockIdx.y];
  const uint32_t* __restrict__ start = k.start[blockIdx.y];
  uint32_t* __restrict__ ranks = k.ranks[blockIdx.y];
  uint32_t* __restrict__ final_cts = k.final_cts[blockIdx.y];
  GSTRIDE(i, n) {
    const uint32_t a = skey[i];
    uint32_t r = 0;
    if (a < m) {
      r = (uint32_t)i - start[a];
      if (i + 1 == n || skey[i + 1] != a) final_cts[a] = r + 1;
    }
    ranks[i] = r;
  }
}
__global__ __launch_bounds__(1024) void lasso_unpermute_cols_kernel(CtCols k, unsigned group_log) {
  extern __shared__ uint32_t win[];
  const uint32_t* __restrict__ pidx = k.pidx[blockIdx.y];
  const uint32_t* __restrict__ prank = k.prank[blockIdx.y];
  const unsigned parts_log = group_log - UNPERM_WINDOW_LOG;
  const size_t g = blockIdx.x >> parts_log;
  const uint32_t h = blockIdx.x & ((1u << parts_log) - 1u);
  const size_t p0 = g << group_log, cnt = (size_t)1 << group_log;
  const uint32_t wmask = (1u << UNPERM_WINDOW_LOG) - 1u;
  for (size_t q = threadIdx.x; q < cnt; q += blockDim.x) {
    const uint32_t idx = pidx[p0 + q];
    if (((idx >> UNPERM_WINDOW_LOG) & ((1u << parts_log) - 1u)) == h) win[idx & wmask] = prank[p0 + q];
  }
  __syncthreads();
  uint32_t* out = k.read_ts[blockIdx.y] + (g << group_log) + ((size_t)h << UNPERM_WINDOW_LOG);
  for (uint32_t q = threadIdx.x; q <= wmask; q += blockDim.x) out[q] = win[q];
}
// the launch set of the columns dims[0 .. cc): batched sort, run starts, ranks, partition pass, un-permute
static void lasso_counters_launch(Ctx& c, const uint32_t* const* dims, size_t cc, size_t n, size_t m, uint32_t* const* read_ts,
                                  uint32_t* const* final_cts, uint32_t* const* keep_sorted, uint32_t* const* keep_index,
                                  uint32_t* bad) {
  ArenaScope scope(c.arena);  // (everything below is queued on the ctx's stream: the next user of this memory comes behind it)
  unsigned bits = 1;
  while (((size_t)1 << bits) < m) bits++;
  CtCols k;
  memset(&k, 0, sizeof(k));
  std::vector<uint32_t*> skey(cc), sidx(cc);
  for (size_t j = 0; j < cc; j++) {
    skey[j] = keep_sorted ? keep_sorted[j] : c.arena.alloc_n<uint32_t>(n);
    sidx[j] = keep_index ? keep_index[j] : c.arena.alloc_n<uint32_t>(n);
    k.skey[j] = skey[j], k.sidx[j] = sidx[j];
    k.start[j] = c.arena.alloc_n<uint32_t>(m);
    k.read_ts[j] = read_ts[j], k.final_cts[j] = final_cts[j];
    LH_HIP(hipMemsetAsync(final_cts[j], 0, m * sizeof(uint32_t), c.stream));
  }
  std::vector<SortSlab> sorts(cc);
  // (address, lookup index) pairs: the index is the position
  for (size_t j = 0; j < cc; j++) sorts[j] = SortSlab{dims[j], skey[j], nullptr, sidx[j], n, bits};
  sort_pairs_u32_batched(c, sorts.data(), cc);
  dim3 g = grid_for(n);
  g.y = (unsigned)cc;
  hipLaunchKernelGGL(lasso_run_start_cols_kernel, g, 256, 0, c.stream, k, n, m, bad);
  unsigned lg = 0;
  while (((size_t)1 << lg) < n) lg++;
  // (2^17 .. 2^24 lookups: one partition pass; beyond, the second pass costs what the scatter did - 2^26: 5.3 -> 5.8 ms)
  if (lg >= 17 && lg <= UNPERM_WINDOW_LOG + 9 && n == ((size_t)1 << lg)) {
    std::vector<uint32_t*> pidx(cc), prank(cc);
    for (size_t j = 0; j < cc; j++) {
      k.ranks[j] = c.arena.alloc_n<uint32_t>(n);
      pidx[j] = c.arena.alloc_n<uint32_t>(n), prank[j] = c.arena.alloc_n<uint32_t>(n);
      k.pidx[j] = pidx[j], k.prank[j] = prank[j];
    }
    hipLaunchKernelGGL(lasso_rank_sorted_cols_kernel, g, 256, 0, c.stream, k, n, m);
    // groups of 2^15 (one window per workgroup) when that takes one pass of <= 8 bits, 2^16 (two windows, the pairs read
    // twice) at 2^24
    const unsigned top = lg - UNPERM_WINDOW_LOG;
    const unsigned pbits = top == 9 ? 8 : top, group_log = lg - pbits;
    for (size_t j = 0; j < cc; j++) sorts[j] = SortSlab{sidx[j], pidx[j], k.ranks[j], prank[j], n, pbits, group_log};
    sort_pairs_u32_batched(c, sorts.data(), cc);
    c.opt_in_lds((const void*)lasso_unpermute_cols_kernel, (int)(4u << UNPERM_WINDOW_LOG));
    hipLaunchKernelGGL(lasso_unpermute_cols_kernel, dim3((unsigned)(n >> UNPERM_WINDOW_LOG), (unsigned)cc), dim3(1024),
                       4u << UNPERM_WINDOW_LOG, c.stream, k, group_log);
  } else {
    hipLaunchKernelGGL(lasso_rank_cols_kernel, g, 256, 0, c.stream, k, n, m);
  }
}
// The access counters of all `cc` chunk columns with ONE readback of the bad-index flag per call (until round 6: a blocking
// 4-byte download per column).  Small columns (launch-bound) go through every step together - one batched sort per step,
// blockIdx.y = column; from 2^22 lookups on the columns go one after the other: a column's working set (its sorted keys
// and positions, 8 n bytes) then stays in the 256 MB memory-side cache between two steps, which four columns side by
// side do not (measured at 2^24 AND lookups: 2.43 ms column by column, 2.58 ms batched).
// keep_sorted / keep_index (null, or one pointer per column): the columns' values in ascending order and the positions
// they came from - the commit MSM's entry stream for that column (MsmJob::sorted_scalars).
void k_lasso_counters(Ctx& c, const uint32_t* const* dims, size_t cc, size_t n, size_t m, uint32_t* const* read_ts,
                      uint32_t* const* final_cts, uint32_t* const* keep_sorted, uint32_t* const* keep_index) {
  LH_REQUIRE(cc >= 1 && cc <= (size_t)LH_LASSO_MAX_CHUNKS, LH_ERR_ARG, "lasso counters: bad column count");
  ProfScope ps(c, "lasso_counters", (8.0 * n + 4.0 * m) * cc, 0.0, (double)(n * cc));
  ArenaScope scope(c.arena);
  uint32_t* bad = c.arena.alloc_n<uint32_t>(1);
  LH_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), c.stream));
  if (!n) {
    for (size_t j = 0; j < cc; j++) LH_HIP(hipMemsetAsync(final_cts[j], 0, m * sizeof(uint32_t), c.stream));
    return;
  }
  const size_t step = n >= ((size_t)1 << 22) ? 1 : cc;
  for (size_t j = 0; j < cc; j += step)
    lasso_counters_launch(c, dims + j, std::min(step, cc - j), n, m, read_ts + j, final_cts + j,
                          keep_sorted ? keep_sorted + j : nullptr, keep_index ? keep_index + j : nullptr, bad);
  uint32_t h_bad = 0;
  c.d2h(&h_bad, bad, sizeof(uint32_t));
  LH_REQUIRE(!h_bad, LH_ERR_ARG, "lasso: chunk index out of range (>= 2^chunk_bits)");
}

// ---- access counters of a sharded proof (dev.hpp Shard): the lookups of a column are repartitioned by ADDRESS (owner =
// address mod R), the owner ranks every lookup of its addresses in the global lookup order, the ranks travel back.  ALL
// chunk columns of the table go through every step together (blockIdx.y = column, one batched sort): the steps and the
// three exchanges are per proof, not per column.
//
// What travels is a 32-bit key per lookup, (address on the owner) << hi_bits | hi, hi = the bits of the GLOBAL lookup index
// above the shard bits (= local index >> shard_bit).  An owner receives one segment per sender s, each in the sender's
// local order, i.e. ascending (hi, lo): the receive buffer is ordered (s, hi, lo), and a STABLE sort by (address, hi)
// leaves every address run in the order (hi, s, lo) - the global lookup order - without s or lo ever being sent.
constexpr int CS_MAX_COLS = LH_LASSO_MAX_CHUNKS;
struct CsCols {
  const uint32_t* in[CS_MAX_COLS];
  const uint32_t* in2[CS_MAX_COLS];
  uint32_t* out[CS_MAX_COLS];
  uint32_t* out2[CS_MAX_COLS];
  uint32_t n[CS_MAX_COLS];
};
__global__ void cs_owner_keys_kernel(CsCols k, size_t n, size_t m, uint32_t owner_mask, uint32_t* __restrict__ bad) {
  const uint32_t* __restrict__ dim = k.in[blockIdx.y];
  uint32_t* __restrict__ okey = k.out[blockIdx.y];
  GSTRIDE(i, n) {
    const uint32_t a = dim[i];
    okey[i] = a & owner_mask;
    if (a >= m) *bad = 1u;
  }
}
// start[col][o] = first position of owner o in the owner-sorted list (left at `n` for owners that do not occur)
__global__ void cs_owner_starts_kernel(CsCols k, size_t n, uint32_t R1, uint32_t* __restrict__ start) {
  const uint32_t* __restrict__ sown = k.in[blockIdx.y];
  GSTRIDE(p, n)
    if (p == 0 || sown[p - 1] != sown[p]) start[(size_t)blockIdx.y * R1 + sown[p]] = (uint32_t)p;
}
// send[p] = (address on its owner) << hi_bits | (local index >> shard_bit), in owner-sorted order (stable: within an owner
// the lookups stay in local order)
__global__ void cs_send_keys_kernel(CsCols k, size_t n, unsigned rho, unsigned j, unsigned hi_bits) {
  const uint32_t* __restrict__ dim = k.in[blockIdx.y];
  const uint32_t* __restrict__ sidx = k.in2[blockIdx.y];
  uint32_t* __restrict__ send = k.out[blockIdx.y];
  GSTRIDE(p, n) {
    const uint32_t li = sidx[p];
    send[p] = ((dim[li] >> rho) << hi_bits) | (li >> j);
  }
}
__global__ void cs_run_start_kernel(CsCols k, unsigned hi_bits, size_t m_loc, uint32_t* __restrict__ start) {
  const uint32_t* __restrict__ skey = k.in[blockIdx.y];
  const size_t n = k.n[blockIdx.y];
  GSTRIDE(i, n) {
    const uint32_t a = skey[i] >> hi_bits;
    if (i == 0 || (skey[i - 1] >> hi_bits) != a) start[(size_t)blockIdx.y * m_loc + a] = (uint32_t)i;
  }
}
// ret[position in the receive buffer] = rank of the lookup among the lookups of its address; counts[a] = run length
__global__ void cs_rank_kernel(CsCols k, unsigned hi_bits, size_t m_loc, const uint32_t* __restrict__ start,
                               uint32_t* __restrict__ counts) {
  const uint32_t* __restrict__ skey = k.in[blockIdx.y];
  const uint32_t* __restrict__ spos = k.in2[blockIdx.y];
  uint32_t* __restrict__ ret = k.out[blockIdx.y];
  const size_t n = k.n[blockIdx.y];
  GSTRIDE(i, n) {
    const uint32_t a = skey[i] >> hi_bits;
    const uint32_t r = (uint32_t)i - start[(size_t)blockIdx.y * m_loc + a];
    ret[spos[i]] = r;
    if (i + 1 == n || (skey[i + 1] >> hi_bits) != a) counts[(size_t)blockIdx.y * m_loc + a] = r + 1;
  }
}
__global__ void cs_scatter_kernel(CsCols k, size_t n) {
  const uint32_t* __restrict__ back = k.in[blockIdx.y];
  const uint32_t* __restrict__ sidx = k.in2[blockIdx.y];
  uint32_t* __restrict__ read_ts = k.out[blockIdx.y];
  GSTRIDE(p, n) read_ts[sidx[p]] = back[p];
}
// final_cts[col][a] = (owner a mod R).counts[col][a >> rho]; all_counts: rank-major blocks of cc * m_loc counts
__global__ void cs_final_kernel(CsCols k, const uint32_t* __restrict__ all_counts, size_t m, unsigned rho, size_t m_loc, size_t cc) {
  const uint32_t mask = (1u << rho) - 1u;
  uint32_t* __restrict__ final_cts = k.out[blockIdx.y];
  GSTRIDE(a, m) final_cts[a] = all_counts[((size_t)(a & mask) * cc + blockIdx.y) * m_loc + (a >> rho)];
}
static dim3 cs_grid(size_t n, size_t cc) {
  dim3 g = grid_for(n);
  g.y = (unsigned)cc;
  return g;
}

void k_cs_partition(Ctx& c, const uint32_t* const* dims, size_t cc, size_t n, size_t m, unsigned rho, unsigned j, unsigned hi_bits,
                    uint32_t* const* sidx, uint32_t* const* send, uint32_t* start_host, bool* bad_out) {
  LH_REQUIRE(cc >= 1 && cc <= (size_t)CS_MAX_COLS && n < ((size_t)1 << 32), LH_ERR_ARG, "sharded counters: bad shape");
  ProfScope ps(c, "lasso_counters/partition", 16.0 * n * cc, 0.0, (double)(n * cc));
  ArenaScope scope(c.arena);
  const size_t R = (size_t)1 << rho, R1 = R + 1;
  uint32_t* okey = c.arena.alloc_n<uint32_t>(cc * n);
  uint32_t* sown = c.arena.alloc_n<uint32_t>(cc * n);
  uint32_t* start = c.arena.alloc_n<uint32_t>(cc * R1 + 1);
  uint32_t* bad = start + cc * R1;
  std::vector<uint32_t> init(cc * R1 + 1, (uint32_t)n);
  init[cc * R1] = 0;
  LH_HIP(hipMemcpyAsync(start, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
  CsCols k;
  memset(&k, 0, sizeof(k));
  for (size_t q = 0; q < cc; q++) k.in[q] = dims[q], k.out[q] = okey + q * n;
  hipLaunchKernelGGL(cs_owner_keys_kernel, cs_grid(n, cc), 256, 0, c.stream, k, n, m, (uint32_t)(R - 1), bad);
  {  // stable sort by owner, values = local positions: ONE batch for all columns
    std::vector<SortSlab> slabs;
    for (size_t q = 0; q < cc; q++) slabs.push_back(SortSlab{okey + q * n, sown + q * n, nullptr, sidx[q], n, std::max(rho, 1u)});
    sort_pairs_u32_batched(c, slabs.data(), slabs.size());
  }
  for (size_t q = 0; q < cc; q++) k.in[q] = sown + q * n;
  hipLaunchKernelGGL(cs_owner_starts_kernel, cs_grid(n, cc), 256, 0, c.stream, k, n, (uint32_t)R1, start);
  for (size_t q = 0; q < cc; q++) k.in[q] = dims[q], k.in2[q] = sidx[q], k.out[q] = send[q];
  hipLaunchKernelGGL(cs_send_keys_kernel, cs_grid(n, cc), 256, 0, c.stream, k, n, rho, j, hi_bits);
  std::vector<uint32_t> h(cc * R1 + 1);
  c.d2h(h.data(), start, h.size() * sizeof(uint32_t));  // (synchronises: `init` may go)
  // (an index >= 2^chunk_bits: NOT raised here - this rank's peers are about to enter a collective and would wait forever;
  // the caller lets the verdict travel with the segment boundaries and raises on every rank)
  *bad_out = h[cc * R1] != 0;
  for (size_t q = 0; q < cc; q++) {
    // owners that do not occur start where the next one does
    uint32_t next = (uint32_t)n;
    for (size_t o = R; o-- > 0;) {
      if (h[q * R1 + o] == (uint32_t)n) h[q * R1 + o] = next;
      next = h[q * R1 + o];
    }
    for (size_t o = 0; o < R; o++) start_host[q * R1 + o] = h[q * R1 + o];
    start_host[q * R1 + R] = (uint32_t)n;
  }
}

void k_cs_rank(Ctx& c, const uint32_t* const* recv, const size_t* n_recv, size_t cc, unsigned hi_bits, unsigned a_bits, size_t m_loc,
               uint32_t* const* ret, uint32_t* counts) {
  LH_REQUIRE(cc >= 1 && cc <= (size_t)CS_MAX_COLS && hi_bits + a_bits <= 32, LH_ERR_ARG, "sharded counters: bad shape");
  size_t total = 0, n_max = 0;
  for (size_t q = 0; q < cc; q++) total += n_recv[q], n_max = std::max(n_max, n_recv[q]);
  ProfScope ps(c, "lasso_counters/rank", 24.0 * total, 0.0, (double)total);
  ArenaScope scope(c.arena);
  LH_HIP(hipMemsetAsync(counts, 0, cc * m_loc * sizeof(uint32_t), c.stream));
  if (!total) return;
  LH_REQUIRE(n_max < ((size_t)1 << 32), LH_ERR_ARG, "sharded counters: too many lookups on one owner");
  uint32_t* skey = c.arena.alloc_n<uint32_t>(total);
  uint32_t* spos = c.arena.alloc_n<uint32_t>(total);
  uint32_t* start = c.arena.alloc_n<uint32_t>(cc * m_loc);
  CsCols k;
  memset(&k, 0, sizeof(k));
  std::vector<SortSlab> slabs;
  size_t off = 0;
  for (size_t q = 0; q < cc; q++) {
    // stable sort by (address, hi), values = positions in the receive buffer
    slabs.push_back(SortSlab{recv[q], skey + off, nullptr, spos + off, n_recv[q], std::max(hi_bits + a_bits, 1u)});
    k.in[q] = skey + off, k.in2[q] = spos + off, k.out[q] = ret[q], k.n[q] = (uint32_t)n_recv[q];
    off += n_recv[q];
  }
  sort_pairs_u32_batched(c, slabs.data(), slabs.size());
  hipLaunchKernelGGL(cs_run_start_kernel, cs_grid(n_max, cc), 256, 0, c.stream, k, hi_bits, m_loc, start);
  hipLaunchKernelGGL(cs_rank_kernel, cs_grid(n_max, cc), 256, 0, c.stream, k, hi_bits, m_loc, start, counts);
}

void k_cs_scatter(Ctx& c, const uint32_t* const* back, const uint32_t* const* sidx, size_t cc, size_t n, uint32_t* const* read_ts) {
  if (!n) return;
  CsCols k;
  memset(&k, 0, sizeof(k));
  for (size_t q = 0; q < cc; q++) k.in[q] = back[q], k.in2[q] = sidx[q], k.out[q] = read_ts[q];
  hipLaunchKernelGGL(cs_scatter_kernel, cs_grid(n, cc), 256, 0, c.stream, k, n);
}
void k_cs_final(Ctx& c, const uint32_t* all_counts, size_t cc, size_t m, unsigned rho, size_t m_loc, uint32_t* const* final_cts) {
  if (!m) return;
  CsCols k;
  memset(&k, 0, sizeof(k));
  for (size_t q = 0; q < cc; q++) k.out[q] = final_cts[q];
  hipLaunchKernelGGL(cs_final_kernel, cs_grid(m, cc), 256, 0, c.stream, k, all_counts, m, rho, m_loc, cc);
}

__global__ void lasso_subtable_read_kernel(int kind, uint32_t bits, const uint32_t* __restrict__ dim, size_t n,
                                           uint32_t* __restrict__ e) {
  GSTRIDE(i, n) e[i] = subtable_entry(kind, dim[i], bits);
}
void k_lasso_subtable_read(Ctx& c, int subtable, uint32_t chunk_bits, const uint32_t* dim, size_t n, uint32_t* e) {
  if (n) hipLaunchKernelGGL(lasso_subtable_read_kernel, grid_for(n), 256, 0, c.stream, subtable, chunk_bits, dim, n, e);
}

// HyperPlonk's Lasso lookups take their chunk columns from the circuit's (field-element) polys: canonical value of
// every entry as u32; `bad` is raised by an entry that is not an index of the 2^bits-entry subtable
__global__ void fr_to_index_kernel(const Fr* __restrict__ in, size_t n, uint32_t bits, uint32_t* __restrict__ out,
                                   uint32_t* __restrict__ bad) {
  GSTRIDE(i, n) {
    const Fr v = from_mont(in[i]);
    uint32_t hi = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) hi |= v.l[k];
    if (hi || (bits < 32 && (v.l[0] >> bits))) atomicOr(bad, 1u);
    out[i] = v.l[0];
  }
}
bool k_fr_to_index(Ctx& c, const Fr* in, size_t n, uint32_t bits, uint32_t* out) {
  ArenaScope scope(c.arena);
  uint32_t* bad = c.arena.alloc_n<uint32_t>(1);
  LH_HIP(hipMemsetAsync(bad, 0, 4, c.stream));
  if (n) hipLaunchKernelGGL(fr_to_index_kernel, grid_for(n), 256, 0, c.stream, in, n, bits, out, bad);
  uint32_t h = 0;
  c.d2h(&h, bad, 4);
  return h == 0;
}
__global__ void fr_tables_equal_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t n,
                                       uint32_t* __restrict__ bad) {
  GSTRIDE(i, n) {
    const Fr x = a[i], y = b[i];
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d |= x.l[k] ^ y.l[k];
    if (d) atomicOr(bad, 1u);
  }
}
bool k_fr_tables_equal(Ctx& c, const Fr* a, const Fr* b, size_t n) {
  ArenaScope scope(c.arena);
  uint32_t* bad = c.arena.alloc_n<uint32_t>(1);
  LH_HIP(hipMemsetAsync(bad, 0, 4, c.stream));
  if (n) hipLaunchKernelGGL(fr_tables_equal_kernel, grid_for(n), 256, 0, c.stream, a, b, n, bad);
  uint32_t h = 0;
  c.d2h(&h, bad, 4);
  return h == 0;
}

__global__ void lasso_output_kernel(LassoG g, size_t n, Fr* __restrict__ a) {
  GSTRIDE(i, n) {
    Fr acc = Fr::zero();
    for (uint32_t t = 0; t < g.num_terms; t++) {
      Fr v = g.coeff[t];
      for (int k = 0; k < g.nfac[t]; k++) v = mul(v, from_u64<FrParams>(g.e[g.fac[t][k]][i]));
      acc = add(acc, v);
    }
    a[i] = acc;
  }
}
// the same as a 32-bit column when g is linear with small coefficients and the value fits (range / AND / XOR tables of
// <= 32-bit operands): no field arithmetic, 4 bytes written instead of 32
__global__ void lasso_output_small_kernel(LassoGSmall g, size_t n, uint32_t* __restrict__ a) {
  GSTRIDE(i, n) {
    uint32_t acc = 0;
    for (uint32_t t = 0; t < g.num_terms; t++) acc += g.coeff[t] * g.e[g.fac[t]][i];
    a[i] = acc;
  }
}
void k_lasso_output_small(Ctx& c, const LassoGSmall& g, size_t n, uint32_t* a) {
  ProfScope ps(c, "lasso_output", 4.0 * n * (g.num_terms + 1), 0.0, (double)n);
  if (n) hipLaunchKernelGGL(lasso_output_small_kernel, grid_for(n), 256, 0, c.stream, g, n, a);
}
void k_lasso_output(Ctx& c, const LassoG& g, size_t n, Fr* a) {
  ProfScope ps(c, "lasso_output", 36.0 * n, 2.0 * n, (double)n);
  if (n) hipLaunchKernelGGL(lasso_output_kernel, grid_for(n), 256, 0, c.stream, g, n, a);
}

__global__ void or_u32_kernel(const uint32_t* const* __restrict__ cols, size_t n, uint32_t* __restrict__ out) {
  const uint32_t* col = cols[blockIdx.y];
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc |= col[i];
  for (int off = 32; off > 0; off >>= 1) acc |= __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicOr(&out[blockIdx.y], acc);
}
void k_or_u32(Ctx& c, const uint32_t* const* cols, size_t count, size_t n, uint32_t* out_host) {
  if (!count) return;
  ArenaScope scope(c.arena);
  const uint32_t** d_cols = (const uint32_t**)c.arena.alloc(count * sizeof(uint32_t*));
  uint32_t* d_out = c.arena.alloc_n<uint32_t>(count);
  void* stage = c.pin(count * sizeof(uint32_t*));
  memcpy(stage, cols, count * sizeof(uint32_t*));
  LH_HIP(hipMemcpyAsync(d_cols, stage, count * sizeof(uint32_t*), hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemsetAsync(d_out, 0, count * sizeof(uint32_t), c.stream));
  if (n)
    hipLaunchKernelGGL(or_u32_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 256), (unsigned)count), dim3(256), 0,
                       c.stream, d_cols, n, d_out);
  c.d2h(out_host, d_out, count * sizeof(uint32_t));
}
__global__ void fill_u32_kernel(uint32_t* __restrict__ out, uint32_t value, size_t n) {
  GSTRIDE(i, n) out[i] = value;
}
void k_fill_u32(Ctx& c, uint32_t* out, uint32_t value, size_t n) {
  if (n) hipLaunchKernelGGL(fill_u32_kernel, grid_for(n), 256, 0, c.stream, out, value, n);
}
__global__ void delta_u32_kernel(const uint32_t* __restrict__ col, size_t len, size_t half, uint64_t offset,
                                 uint32_t* __restrict__ out_lo, uint32_t* __restrict__ out_hi) {
  GSTRIDE(i, half) {
    const uint64_t lo = i < len ? col[i] : 0u, hi = i + half < len ? col[i + half] : 0u;
    const uint64_t v = hi + offset - lo;
    if (out_hi) {
      out_lo[i] = (uint32_t)(v & 0xffffu);
      out_hi[i] = (uint32_t)(v >> 16);
    } else {
      out_lo[i] = (uint32_t)v;
    }
  }
}
void k_delta_u32(Ctx& c, const uint32_t* col, size_t len, size_t half, uint64_t offset, uint32_t* out_lo, uint32_t* out_hi) {
  if (half)
    hipLaunchKernelGGL(delta_u32_kernel, grid_for(half), 256, 0, c.stream, col, len, half, offset, out_lo, out_hi);
}
__global__ void pack_u32_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t shift, size_t n,
                                uint32_t* __restrict__ out) {
  GSTRIDE(i, n) out[i] = a[i] | (b[i] << shift);
}
void k_pack_u32(Ctx& c, const uint32_t* a, const uint32_t* b, uint32_t shift, size_t n, uint32_t* out) {
  if (n) hipLaunchKernelGGL(pack_u32_kernel, grid_for(n), 256, 0, c.stream, a, b, shift, n, out);
}

// fingerprint h(a, v, t) = a*gamma^2 + v*gamma + t - tau
// (gamma_r, gamma2_r, one_r: the weights times R, prescale_r - one Montgomery reduction per leaf, wide_redc)
__global__ void lasso_rw_leaves_kernel(const uint32_t* __restrict__ dim, const uint32_t* __restrict__ e,
                                       const uint32_t* __restrict__ ts, size_t n, Fr gamma_r, Fr gamma2_r, Fr one_r, Fr tau,
                                       Fr* __restrict__ rs, Fr* __restrict__ ws) {
  const Fr one = Fr::one();
  GSTRIDE(i, n) {
    // three residue-times-small-integer products into one wide accumulator, one reduction (see `Wide` above)
    Wide t = Wide::zero();
    wide_mac(t, gamma2_r, dim[i]);
    wide_mac(t, gamma_r, e[i]);
    wide_mac(t, one_r, ts[i]);
    Fr h = sub(wide_redc(t), tau);
    rs[i] = h;
    ws[i] = add(h, one);
  }
}
// the same with the first level of the two product trees: node i of the level above the leaves is
// leaf[i] * leaf[i + n/2] (the trees split on the top bit), so the thread that makes both leaves also makes their
// product - the level is not read back from HBM by a tree_up pass (2 x 32 B per leaf saved)
__global__ void lasso_rw_leaves_up_kernel(const uint32_t* __restrict__ dim, const uint32_t* __restrict__ e,
                                          const uint32_t* __restrict__ ts, size_t half, Fr gamma_r, Fr gamma2_r, Fr one_r,
                                          Fr tau, Fr* __restrict__ rs, Fr* __restrict__ ws, Fr* __restrict__ rs_up,
                                          Fr* __restrict__ ws_up) {
  const Fr one = Fr::one();
  GSTRIDE(i, half) {
    Fr h[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const size_t j = i + (k ? half : 0);
      Wide t = Wide::zero();
      wide_mac(t, gamma2_r, dim[j]);
      wide_mac(t, gamma_r, e[j]);
      wide_mac(t, one_r, ts[j]);
      h[k] = sub(wide_redc(t), tau);
      rs[j] = h[k];
    }
    const Fr w0 = add(h[0], one), w1 = add(h[1], one);
    if (ws) {  // (null: the caller's leaf layer runs over the read set alone)
      ws[i] = w0;
      ws[i + half] = w1;
    }
    rs_up[i] = mul(h[0], h[1]);
    ws_up[i] = mul(w0, w1);
  }
}
void k_lasso_rw_leaves_up(Ctx& c, const uint32_t* dim, const uint32_t* e, const uint32_t* ts, size_t n, const Fr& gamma,
                          const Fr& gamma2, const Fr& tau, Fr* rs, Fr* ws, Fr* rs_up, Fr* ws_up) {
  ProfScope ps(c, "lasso_rw_leaves", (12.0 + (ws ? 64.0 : 32.0) + 32.0) * n, 6.0 * n, (double)n);
  if (n >= 2)
    hipLaunchKernelGGL(lasso_rw_leaves_up_kernel, grid_for(n / 2), 256, 0, c.stream, dim, e, ts, n / 2, prescale_r(gamma),
                       prescale_r(gamma2), prescale_r(Fr::one()), tau, rs, ws, rs_up, ws_up);
}
void k_lasso_rw_leaves(Ctx& c, const uint32_t* dim, const uint32_t* e, const uint32_t* ts, size_t n, const Fr& gamma,
                       const Fr& gamma2, const Fr& tau, Fr* rs, Fr* ws) {
  ProfScope ps(c, "lasso_rw_leaves", (12.0 + 64.0) * n, 5.0 * n, (double)n);
  if (n)
    hipLaunchKernelGGL(lasso_rw_leaves_kernel, grid_for(n), 256, 0, c.stream, dim, e, ts, n, prescale_r(gamma),
                       prescale_r(gamma2), prescale_r(Fr::one()), tau, rs, ws);
}
__global__ void lasso_if_leaves_kernel(int kind, uint32_t bits, const uint32_t* __restrict__ final_cts, size_t m,
                                       Fr gamma, Fr gamma2, Fr tau, Fr* __restrict__ init, Fr* __restrict__ fin) {
  GSTRIDE(i, m) {
    Fr a = mul(from_u64<FrParams>(i), gamma2);
    Fr v = mul(from_u64<FrParams>(subtable_entry(kind, (uint32_t)i, bits)), gamma);
    Fr h = sub(add(a, v), tau);
    init[i] = h;
    fin[i] = add(h, from_u64<FrParams>(final_cts[i]));
  }
}
void k_lasso_if_leaves(Ctx& c, int subtable, uint32_t chunk_bits, const uint32_t* final_cts, size_t m,
                       const Fr& gamma, const Fr& gamma2, const Fr& tau, Fr* init, Fr* fin) {
  ProfScope ps(c, "lasso_if_leaves", (4.0 + 64.0) * m, 5.0 * m, (double)m);
  if (m)
    hipLaunchKernelGGL(lasso_if_leaves_kernel, grid_for(m), 256, 0, c.stream, subtable, chunk_bits, final_cts, m,
                       gamma, gamma2, tau, init, fin);
}

}  // namespace lh
