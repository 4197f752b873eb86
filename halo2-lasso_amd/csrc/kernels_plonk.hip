// LogUp / permutation witness polynomials of HyperPlonk::prove on the GPU.
//   lookup_m_poly        backend/hyperplonk/prover.rs:145-192  (HashMap table->LAST index, counts)
//   lookup_h_poly        prover.rs:206-250                     (h = 1/(gamma+f) - m/(gamma+t))
//   permutation_z_polys  prover.rs:252-345                     (products, inversion, prefix product in
//                                                               BooleanHypercube order, remap)
#include <hip/hip_runtime.h>
#include <cstring>
#include "dev.hpp"

namespace lh {

#define GSTRIDE(i, n) \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)
static inline dim3 grid_for(size_t n, int block = 256, size_t cap = 4096) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return dim3((unsigned)g);
}

// ------------------------------------------------------------------ lookup_m_poly: sort-merge join
// key = low 64 bits of the (Montgomery) value: equal values have equal keys, a key collision between
// different values is resolved by the full comparison in the probe.
__device__ __forceinline__ uint64_t key_of(const Fr& v) { return (uint64_t)v.l[0] | ((uint64_t)v.l[1] << 32); }

__global__ void m_keys_kernel(const Fr* __restrict__ table, size_t n, uint64_t* __restrict__ keys,
                              uint32_t* __restrict__ idx) {
  GSTRIDE(i, n) {
    keys[i] = key_of(table[i]);
    idx[i] = (uint32_t)i;
  }
}
__global__ void m_probe_kernel(const Fr* __restrict__ input, const Fr* __restrict__ table,
                               const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ sidx, size_t n,
                               uint32_t* __restrict__ counts, uint32_t* __restrict__ missing) {
  GSTRIDE(i, n) {
    const Fr v = input[i];
    const uint64_t k = key_of(v);
    size_t lo = 0, hi = n;
    while (lo < hi) {  // upper bound: first key > k
      size_t mid = (lo + hi) >> 1;
      if (skeys[mid] <= k) lo = mid + 1;
      else hi = mid;
    }
    // among the table rows holding exactly this value the LAST one wins (HashMap::collect, prover.rs:151).
    // The sort is stable, so row indices ascend inside a run of equal keys: walking down from the end of the
    // run, the first full match is that row (one comparison unless two different values share their low 64 bits;
    // a table padded with many equal rows does not make the probe quadratic).
    long long best = -1;
    for (size_t p = lo; p-- > 0 && skeys[p] == k;) {
      uint32_t t = sidx[p];
      if (table[t] == v) {
        best = (long long)t;
        break;
      }
    }
    // count: lanes of a wave that hit the same table row are combined before the atomic (a selector column that
    // is zero on most rows sends most inputs to ONE row; unaggregated that is a million atomics on one address)
    if (best < 0) atomicAdd(missing, 1u);  // Error::InvalidSnark("Invalid lookup input") (prover.rs:176-178)
    unsigned long long todo = __ballot(best >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const long long v = __shfl(best, leader, 64);
      const unsigned long long same = __ballot(best == v) & todo;
      if ((int)(threadIdx.x & 63) == leader) atomicAdd(&counts[v], (uint32_t)__popcll(same));
      todo &= ~same;
    }
  }
}

bool k_lookup_m(Ctx& c, const Fr* input, const Fr* table, size_t n, Fr* m_out) {
  ProfScope ps(c, "lookup_m", (64.0 + 32.0 + 24.0 * 4) * n, 0.0, (double)n);
  ArenaScope scope(c.arena);
  uint64_t* keys = c.arena.alloc_n<uint64_t>(n);
  uint64_t* skeys = c.arena.alloc_n<uint64_t>(n);
  uint32_t* idx = c.arena.alloc_n<uint32_t>(n);
  uint32_t* sidx = c.arena.alloc_n<uint32_t>(n);
  uint32_t* counts = c.arena.alloc_n<uint32_t>(n + 1);
  LH_HIP(hipMemsetAsync(counts, 0, (n + 1) * sizeof(uint32_t), c.stream));
  hipLaunchKernelGGL(m_keys_kernel, grid_for(n), 256, 0, c.stream, table, n, keys, idx);
  sort_pairs_u64(c, keys, skeys, idx, sidx, n, 64);  // stable: equal values keep their row order (prover.rs:151)
  hipLaunchKernelGGL(m_probe_kernel, grid_for(n), 256, 0, c.stream, input, table, skeys, sidx, n, counts, counts + n);
  k_fr_from_u32(c, counts, n, m_out);
  uint32_t missing = 0;
  c.d2h(&missing, counts + n, 4);
  return missing == 0;
}

// ------------------------------------------------------------------ lookup_h_poly
__global__ void add_scalar_kernel(const Fr* __restrict__ in, Fr g, size_t n, Fr* __restrict__ out) {
  GSTRIDE(i, n) out[i] = add(in[i], g);
}
__global__ void h_final_kernel(const Fr* __restrict__ hi, const Fr* __restrict__ ht, const Fr* __restrict__ m, size_t n,
                               Fr* __restrict__ h) {
  GSTRIDE(i, n) h[i] = sub(hi[i], mul(ht[i], m[i]));
}
void k_lookup_h(Ctx& c, const Fr* input, const Fr* table, const Fr* m, const Fr& gamma, size_t n, Fr* h) {
  ProfScope ps(c, "lookup_h", (3 * 32.0 + 32.0 + 4 * 64.0) * n, 8.0 * n, (double)n);
  ArenaScope scope(c.arena);
  Fr* a = c.arena.alloc_n<Fr>(n);
  Fr* b = c.arena.alloc_n<Fr>(n);
  Fr* ai = c.arena.alloc_n<Fr>(n);
  Fr* bi = c.arena.alloc_n<Fr>(n);
  hipLaunchKernelGGL(add_scalar_kernel, grid_for(n), 256, 0, c.stream, input, gamma, n, a);
  hipLaunchKernelGGL(add_scalar_kernel, grid_for(n), 256, 0, c.stream, table, gamma, n, b);
  k_fr_batch_invert(c, a, n, ai);
  k_fr_batch_invert(c, b, n, bi);
  hipLaunchKernelGGL(h_final_kernel, grid_for(n), 256, 0, c.stream, ai, bi, m, n, h);
}

// ------------------------------------------------------------------ permutation_z_polys
constexpr int PERM_MAX = 8;
struct PermPack {
  const Fr* value[PERM_MAX];
  const Fr* perm[PERM_MAX];
  uint64_t id_offset[PERM_MAX];
  int count;
};
// prod[b] = prod_k (beta * perm_k[b] + gamma + value_k[b])
__global__ void perm_den_kernel(PermPack p, Fr beta, Fr gamma, size_t n, Fr* __restrict__ prod) {
  GSTRIDE(b, n) {
    Fr acc = Fr::one();
    for (int k = 0; k < p.count; k++) acc = mul(acc, add(add(mul(beta, p.perm[k][b]), gamma), p.value[k][b]));
    prod[b] = acc;
  }
}
// prod[b] = inv[b] * prod_k (beta * (id_offset_k + row(b)) + gamma + value_k[b]); row(b) = b, or - the tables being a
// rank's shards (dev.hpp Shard: rho > 0) - the global row (hi, rank, lo) of local index b = hi || lo
__global__ void perm_num_kernel(PermPack p, Fr beta, Fr gamma, size_t n, const Fr* __restrict__ inv,
                                Fr* __restrict__ prod, unsigned sj, unsigned srho, size_t srank) {
  GSTRIDE(b, n) {
    Fr acc = inv[b];
    const size_t row = srho ? (((b >> sj) << (sj + srho)) | (srank << sj) | (b & (((size_t)1 << sj) - 1))) : b;
    for (int k = 0; k < p.count; k++) {
      Fr id = from_u64<FrParams>(p.id_offset[k] + row);
      acc = mul(acc, add(add(mul(beta, id), gamma), p.value[k][b]));
    }
    prod[b] = acc;
  }
}
// seq[(k-1)*num_chunks + c] = products[c][order[k]], k = 1 .. 2^n - 1
struct ChunkPack {
  const Fr* prod[PERM_MAX];
  Fr* z[PERM_MAX];
};
__global__ void perm_seq_kernel(ChunkPack p, int num_chunks, const uint32_t* __restrict__ order, size_t n,
                                Fr* __restrict__ seq) {
  const size_t total = (n - 1) * num_chunks;
  GSTRIDE(t, total) {
    size_t k = t / num_chunks + 1, c = t % num_chunks;
    seq[t] = p.prod[c][order[k]];
  }
}
// z_c[b] = Z[c + num_chunks * nth[b]] with Z = [0]*num_chunks ++ [1] ++ scan (prover.rs:308-344)
__global__ void perm_z_kernel(ChunkPack p, int num_chunks, const uint32_t* __restrict__ nth, size_t n,
                              const Fr* __restrict__ scan) {
  GSTRIDE(t, n * num_chunks) {
    size_t b = t / num_chunks, c = t % num_chunks;
    size_t pos = c + (size_t)num_chunks * nth[b];
    Fr v;
    if (pos < (size_t)num_chunks) v = Fr::zero();
    else if (pos == (size_t)num_chunks) v = Fr::one();
    else v = scan[pos - num_chunks - 1];
    p.z[c][b] = v;
  }
}
// the same for this rank's shard of every z (local index i <-> global row b)
__global__ void perm_z_shard_kernel(ChunkPack p, int num_chunks, const uint32_t* __restrict__ nth, size_t n_local,
                                    const Fr* __restrict__ scan, unsigned sj, unsigned srho, size_t srank) {
  GSTRIDE(t, n_local * num_chunks) {
    size_t i = t / num_chunks, c = t % num_chunks;
    const size_t b = ((i >> sj) << (sj + srho)) | (srank << sj) | (i & (((size_t)1 << sj) - 1));
    size_t pos = c + (size_t)num_chunks * nth[b];
    Fr v;
    if (pos < (size_t)num_chunks) v = Fr::zero();
    else if (pos == (size_t)num_chunks) v = Fr::one();
    else v = scan[pos - num_chunks - 1];
    p.z[c][i] = v;
  }
}

// inclusive prefix PRODUCT over Fr: tiles of 256 threads x 8 elements
constexpr int SCANP_TILE = 2048;
__global__ __launch_bounds__(256) void scanp_tile_kernel(const Fr* __restrict__ in, size_t n, Fr* __restrict__ tile_prod) {
  __shared__ Fr lds[256];
  size_t base = (size_t)blockIdx.x * SCANP_TILE + (size_t)threadIdx.x * 8;
  Fr acc = Fr::one();
  for (int k = 0; k < 8; k++)
    if (base + k < n) acc = mul(acc, in[base + k]);
  lds[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) lds[threadIdx.x] = mul(lds[threadIdx.x], lds[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_prod[blockIdx.x] = lds[0];
}
// exclusive prefix product of the tile products (one workgroup, strided chunks)
__global__ __launch_bounds__(256) void scanp_tiles_kernel(Fr* __restrict__ tile_prod, size_t ntiles) {
  __shared__ Fr lds[256];
  size_t per = (ntiles + 255) / 256;
  size_t lo = threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
  Fr acc = Fr::one();
  for (size_t i = lo; i < hi; i++) acc = mul(acc, tile_prod[i]);
  lds[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive
    Fr t = (int)threadIdx.x >= off ? lds[threadIdx.x - off] : Fr::one();
    __syncthreads();
    lds[threadIdx.x] = mul(lds[threadIdx.x], t);
    __syncthreads();
  }
  Fr run = threadIdx.x ? lds[threadIdx.x - 1] : Fr::one();
  for (size_t i = lo; i < hi; i++) {
    Fr v = tile_prod[i];
    tile_prod[i] = run;
    run = mul(run, v);
  }
}
__global__ __launch_bounds__(256) void scanp_apply_kernel(const Fr* __restrict__ in, size_t n,
                                                          const Fr* __restrict__ tile_off, Fr* __restrict__ out) {
  __shared__ Fr lds[256];
  size_t base = (size_t)blockIdx.x * SCANP_TILE + (size_t)threadIdx.x * 8;
  Fr v[8];
  Fr acc = Fr::one();
  for (int k = 0; k < 8; k++) {
    v[k] = base + k < n ? in[base + k] : Fr::one();
    acc = mul(acc, v[k]);
  }
  lds[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    Fr t = (int)threadIdx.x >= off ? lds[threadIdx.x - off] : Fr::one();
    __syncthreads();
    lds[threadIdx.x] = mul(lds[threadIdx.x], t);
    __syncthreads();
  }
  Fr run = mul(tile_off[blockIdx.x], threadIdx.x ? lds[threadIdx.x - 1] : Fr::one());
  for (int k = 0; k < 8; k++) {
    run = mul(run, v[k]);
    if (base + k < n) out[base + k] = run;
  }
}
static void fr_prefix_product(Ctx& c, const Fr* in, size_t n, Fr* out) {
  if (!n) return;
  ArenaScope scope(c.arena);
  size_t ntiles = (n + SCANP_TILE - 1) / SCANP_TILE;
  Fr* tiles = c.arena.alloc_n<Fr>(ntiles);
  hipLaunchKernelGGL(scanp_tile_kernel, dim3((unsigned)ntiles), dim3(256), 0, c.stream, in, n, tiles);
  hipLaunchKernelGGL(scanp_tiles_kernel, dim3(1), dim3(256), 0, c.stream, tiles, ntiles);
  hipLaunchKernelGGL(scanp_apply_kernel, dim3((unsigned)ntiles), dim3(256), 0, c.stream, in, n, tiles, out);
}

// values[k], perms[k]: the k-th permutation poly's column values and sigma table; z_out[c]: 2^num_vars each.
// Two steps (prover.rs:262-345): the per-row quotients prod_c[b] of every chunk c - entry-wise, so they run on a rank's
// shards as they are (`n` entries; srho > 0: local index -> global row for the identity polys) - and the z polys from
// the products in BooleanHypercube order, a prefix product over ALL rows (the one step of HyperPlonk::prove that does not
// shard, SURVEY.md 8e: a sharded prove gathers the products, runs it on every rank and keeps its own rows of z).
void k_permutation_products(Ctx& c, const Fr* const* values, const Fr* const* perms, size_t num_perm, size_t num_chunks,
                            size_t num_vars, const Fr& beta, const Fr& gamma, size_t n, size_t sj, size_t srho, size_t srank,
                            Fr* const* prod_out) {
  ProfScope ps(c, "permutation_products", (2.0 * num_perm + 3.0 * num_chunks) * 32.0 * n, (2.0 * num_perm + 4.0 * num_chunks) * n, (double)n);
  if (!num_perm) return;
  LH_REQUIRE(num_chunks >= 1 && num_chunks <= (size_t)PERM_MAX, LH_ERR_ARG, "permutation: too many z polys");
  const size_t chunk_size = (num_perm + num_chunks - 1) / num_chunks;
  LH_REQUIRE(chunk_size <= (size_t)PERM_MAX, LH_ERR_ARG, "permutation: chunk too large");
  ArenaScope scope(c.arena);
  Fr* tmp = c.arena.alloc_n<Fr>(n);
  Fr* inv = c.arena.alloc_n<Fr>(n);
  for (size_t ch = 0; ch < num_chunks; ch++) {
    PermPack pk;
    memset(&pk, 0, sizeof(pk));
    size_t lo = ch * chunk_size, hi = std::min(num_perm, lo + chunk_size);
    pk.count = (int)(hi > lo ? hi - lo : 0);
    for (size_t k = lo; k < hi; k++) {
      pk.value[k - lo] = values[k];
      pk.perm[k - lo] = perms[k];
      pk.id_offset[k - lo] = (uint64_t)k << num_vars;
    }
    hipLaunchKernelGGL(perm_den_kernel, grid_for(n), 256, 0, c.stream, pk, beta, gamma, n, tmp);
    k_fr_batch_invert(c, tmp, n, inv);
    hipLaunchKernelGGL(perm_num_kernel, grid_for(n), 256, 0, c.stream, pk, beta, gamma, n, inv, prod_out[ch], (unsigned)sj,
                       (unsigned)srho, srank);
  }
}
// prods[c]: 2^num_vars entries each (all rows); z_out[c]: 2^num_vars entries, or - srho > 0 - this rank's 2^(num_vars - srho)
void k_permutation_z_from_products(Ctx& c, const Fr* const* prods, size_t num_chunks, size_t num_vars, const uint32_t* d_order,
                                   const uint32_t* d_nth, Fr* const* z_out, size_t sj, size_t srho, size_t srank) {
  const size_t n = (size_t)1 << num_vars;
  ProfScope ps(c, "permutation_scan", 5.0 * num_chunks * 32.0 * n, 3.0 * num_chunks * n, (double)n);
  LH_REQUIRE(num_chunks >= 1 && num_chunks <= (size_t)PERM_MAX, LH_ERR_ARG, "permutation: too many z polys");
  ArenaScope scope(c.arena);
  ChunkPack cp;
  memset(&cp, 0, sizeof(cp));
  for (size_t ch = 0; ch < num_chunks; ch++) cp.prod[ch] = prods[ch], cp.z[ch] = z_out[ch];
  const size_t total = (n - 1) * num_chunks;
  Fr* seq = c.arena.alloc_n<Fr>(std::max<size_t>(total, 1));
  Fr* scan = c.arena.alloc_n<Fr>(std::max<size_t>(total, 1));
  hipLaunchKernelGGL(perm_seq_kernel, grid_for(total), 256, 0, c.stream, cp, (int)num_chunks, d_order, n, seq);
  fr_prefix_product(c, seq, total, scan);
  if (srho)
    hipLaunchKernelGGL(perm_z_shard_kernel, grid_for((n >> srho) * num_chunks), 256, 0, c.stream, cp, (int)num_chunks, d_nth,
                       n >> srho, scan, (unsigned)sj, (unsigned)srho, srank);
  else
    hipLaunchKernelGGL(perm_z_kernel, grid_for(n * num_chunks), 256, 0, c.stream, cp, (int)num_chunks, d_nth, n, scan);
}
void k_permutation_z(Ctx& c, const Fr* const* values, const Fr* const* perms, size_t num_perm, size_t num_chunks,
                     size_t num_vars, const Fr& beta, const Fr& gamma, const uint32_t* d_order, const uint32_t* d_nth,
                     Fr* const* z_out) {
  if (!num_perm) return;
  const size_t n = (size_t)1 << num_vars;
  ArenaScope scope(c.arena);
  std::vector<Fr*> prods(num_chunks);
  for (auto& pr : prods) pr = c.arena.alloc_n<Fr>(n);
  k_permutation_products(c, values, perms, num_perm, num_chunks, num_vars, beta, gamma, n, 0, 0, 0, prods.data());
  std::vector<const Fr*> cp(prods.begin(), prods.end());
  k_permutation_z_from_products(c, cp.data(), num_chunks, num_vars, d_order, d_nth, z_out, 0, 0, 0);
}

// table[rows[i]] = vals[i] on a zeroed table (instance polys, prover.rs:32-48)
__global__ void scatter_rows_kernel(const uint32_t* __restrict__ rows, const Fr* __restrict__ vals, size_t count,
                                    Fr* __restrict__ table) {
  GSTRIDE(i, count) table[rows[i]] = vals[i];
}
void k_scatter_rows(Ctx& c, const uint32_t* d_rows, const Fr* d_vals, size_t count, size_t n, Fr* table) {
  LH_HIP(hipMemsetAsync(table, 0, n * sizeof(Fr), c.stream));
  if (count) hipLaunchKernelGGL(scatter_rows_kernel, grid_for(count), 256, 0, c.stream, d_rows, d_vals, count, table);
}

}  // namespace lh
