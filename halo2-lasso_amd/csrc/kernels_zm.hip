// Kernels of the Zeromorph / univariate-KZG opening (reference pcs/multilinear/zeromorph.rs:134-199,
// pcs/univariate/kzg.rs:175-218,264-299).
//   powers         s^i, i < n                                     (setup: powers(s).take(poly_size))
//   zm_qhat        q_hat[2^n - 2^k + j] += y^k q_k[j]             (zeromorph.rs:160-171)
//   zm_combine     f[j] = z poly[j] + q_hat[j] + sum_k s_k q_k[j] (zeromorph.rs:180-184)
//   suffix Horner  S_i = f_i + x S_{i+1}: the quotient of f by X - x (univariate.rs:144-166 for a linear divisor),
//                  in chunks of 64 coefficients: per-chunk value, recursion over the chunk values with x^64, then a
//                  second pass seeded with the chunk's carry.
#include <hip/hip_runtime.h>
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

// out[i] = s^i : thread i starts from s^(64 * (i / 64)) by square-and-multiply and walks its 64-chunk
__global__ void powers_kernel(Fr s, size_t n, Fr* __restrict__ out) {
  const size_t chunks = (n + 63) / 64;
  GSTRIDE(c, chunks) {
    Fr base = Fr::one(), sq = s;
    for (size_t e = c * 64; e; e >>= 1) {
      if (e & 1) base = mul(base, sq);
      sq = mul(sq, sq);
    }
    const size_t end = (c + 1) * 64 < n ? (c + 1) * 64 : n;
    for (size_t i = c * 64; i < end; i++) {
      out[i] = base;
      base = mul(base, s);
    }
  }
}
void k_powers(Ctx& c, const Fr& s, size_t n, Fr* out) {
  if (n) hipLaunchKernelGGL(powers_kernel, grid_for((n + 63) / 64, 64), 64, 0, c.stream, s, n, out);
}

constexpr int ZM_MAX_VARS = 32;
struct ZmScalars {
  Fr v[ZM_MAX_VARS];
};
// q: the quotients flat, q_k (2^k coefficients) at offset 2^k - 1
__global__ void zm_qhat_kernel(const Fr* __restrict__ q, int num_vars, ZmScalars ypow, Fr* __restrict__ q_hat) {
  const size_t n = (size_t)1 << num_vars;
  GSTRIDE(i, n) {
    Fr acc = Fr::zero();
    for (int k = 0; k < num_vars; k++) {
      const size_t sz = (size_t)1 << k;
      if (i >= n - sz) acc = add(acc, mul(ypow.v[k], q[sz - 1 + (i - (n - sz))]));
    }
    q_hat[i] = acc;
  }
}
void k_zm_qhat(Ctx& c, const Fr* q, size_t num_vars, const Fr* ypow, Fr* q_hat) {
  ZmScalars s;
  for (size_t k = 0; k < num_vars; k++) s.v[k] = ypow[k];
  ProfScope ps(c, "zm_qhat", 96.0 * ((size_t)1 << num_vars), 2.0 * ((size_t)1 << num_vars), (double)((size_t)1 << num_vars));
  hipLaunchKernelGGL(zm_qhat_kernel, grid_for((size_t)1 << num_vars), 256, 0, c.stream, q, (int)num_vars, s, q_hat);
}

__global__ void zm_combine_kernel(const Fr* __restrict__ poly, const Fr* __restrict__ q_hat, const Fr* __restrict__ q,
                                  int num_vars, Fr z, ZmScalars qs, Fr* __restrict__ f) {
  const size_t n = (size_t)1 << num_vars;
  GSTRIDE(j, n) {
    Fr acc = add(mul(z, poly[j]), q_hat[j]);
    for (int k = 0; k < num_vars; k++) {
      const size_t sz = (size_t)1 << k;
      if (j < sz) acc = add(acc, mul(qs.v[k], q[sz - 1 + j]));
    }
    f[j] = acc;
  }
}
void k_zm_combine(Ctx& c, const Fr* poly, const Fr* q_hat, const Fr* q, size_t num_vars, const Fr& z,
                  const Fr* q_scalars, Fr* f) {
  ZmScalars s;
  for (size_t k = 0; k < num_vars; k++) s.v[k] = q_scalars[k];
  ProfScope ps(c, "zm_combine", 128.0 * ((size_t)1 << num_vars), 3.0 * ((size_t)1 << num_vars), (double)((size_t)1 << num_vars));
  hipLaunchKernelGGL(zm_combine_kernel, grid_for((size_t)1 << num_vars), 256, 0, c.stream, poly, q_hat, q, (int)num_vars,
                     z, s, f);
}

// ------------------------------------------------------------------ S_i = f_i + x S_{i+1}
constexpr size_t SH_CHUNK = 64;
// value of each chunk as a polynomial in x: L_c = sum_{j in chunk} f_j x^(j - start)
__global__ void sh_chunk_value_kernel(const Fr* __restrict__ f, size_t n, Fr x, Fr* __restrict__ L) {
  const size_t chunks = (n + SH_CHUNK - 1) / SH_CHUNK;
  GSTRIDE(c, chunks) {
    const size_t start = c * SH_CHUNK, end = start + SH_CHUNK < n ? start + SH_CHUNK : n;
    Fr acc = Fr::zero();
    for (size_t i = end; i-- > start;) acc = add(mul(acc, x), f[i]);
    L[c] = acc;
  }
}
// out[i] = S_i inside every chunk, seeded with S at the chunk's end (carry[c + 1], zero past the last chunk)
__global__ void sh_expand_kernel(const Fr* __restrict__ f, size_t n, Fr x, const Fr* __restrict__ carry,
                                 Fr* __restrict__ out) {
  const size_t chunks = (n + SH_CHUNK - 1) / SH_CHUNK;
  GSTRIDE(c, chunks) {
    const size_t start = c * SH_CHUNK, end = start + SH_CHUNK < n ? start + SH_CHUNK : n;
    Fr acc = c + 1 < chunks ? carry[c + 1] : Fr::zero();
    for (size_t i = end; i-- > start;) {
      acc = add(mul(acc, x), f[i]);
      out[i] = acc;
    }
  }
}
__global__ void sh_serial_kernel(const Fr* __restrict__ f, size_t n, Fr x, Fr* __restrict__ out) {
  if (blockIdx.x || threadIdx.x) return;
  Fr acc = Fr::zero();
  for (size_t i = n; i-- > 0;) {
    acc = add(mul(acc, x), f[i]);
    out[i] = acc;
  }
}
// out[i] = sum_{j >= i} f_j x^(j - i), i < n  (out may not alias f)
void k_suffix_horner(Ctx& c, const Fr* f, size_t n, const Fr& x, Fr* out) {
  if (!n) return;
  const Fr xd = x;
  if (n <= SH_CHUNK) {
    hipLaunchKernelGGL(sh_serial_kernel, 1, 1, 0, c.stream, f, n, xd, out);
    return;
  }
  ArenaScope scope(c.arena);
  const size_t chunks = (n + SH_CHUNK - 1) / SH_CHUNK;
  Fr* L = c.arena.alloc_n<Fr>(chunks);
  Fr* T = c.arena.alloc_n<Fr>(chunks);
  hipLaunchKernelGGL(sh_chunk_value_kernel, grid_for(chunks, 64), 64, 0, c.stream, f, n, xd, L);
  Fr x64 = x;  // ff.cuh arithmetic is host-callable
  for (int i = 0; i < 6; i++) x64 = mul(x64, x64);
  k_suffix_horner(c, L, chunks, x64, T);  // T_c = S at the start of chunk c
  hipLaunchKernelGGL(sh_expand_kernel, grid_for(chunks, 64), 64, 0, c.stream, f, n, xd, T, out);
}

}  // namespace lh
