// Kernels for general plonkish expressions (reference util/expression.rs) after symbolic expansion
// into a sum of monomials over "atoms" (a poly at a rotation, eq_xy, identity, Lagrange):
//   * sc_round_ext_kernel   sum-check round (fused bind + evaluate) with the monomial list in device
//                           memory: the HyperPlonk zero-check has ~90 monomials of up to 5 factors, too
//                           many for kernel arguments.  Replaces EvaluationsProver::evals
//                           (piop/sum_check/classic/eval.rs:102-131, 210-323) + next_round (classic.rs:90-141).
//   * rotate_gather_kernel  rotated[b] = poly[bh.rotate(b, rot)]: the round-0 index maps of eval.rs:217-226 /
//                           the materialisation of classic.rs:104-126, done once up front.
//   * identity / lagrange   the tables whose multilinear extensions the reference tracks in closed form
//                           (classic.rs:92-101).
//   * expr_rows_kernel      row-wise evaluation sum_m c_m prod atoms(b): lookup_compressed_poly
//                           (backend/hyperplonk/prover.rs:79-137).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "dev.hpp"
#include "reduce.cuh"
#include "resident.cuh"

namespace lh {

#define GSTRIDE(i, n) \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

static inline dim3 grid_for(size_t n, int block = 256, size_t cap = 4096) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return dim3((unsigned)g);
}

// ------------------------------------------------------------------ BooleanHypercube on the device (bh.rs:143-153)
__device__ __forceinline__ uint32_t bh_rotate(uint32_t b, int rot, uint32_t num_vars, uint32_t primitive, uint32_t x_inv) {
  for (int i = 0; i < rot; i++) {
    uint64_t t = (uint64_t)b << 1;
    b = (uint32_t)(t ^ ((t >> num_vars) * primitive));
  }
  for (int i = 0; i > rot; i--) b = (b >> 1) ^ ((b & 1u) * x_inv);
  return b;
}

// order[k] = bh.iter().nth(k): 0, then x^(k-1) in GF(2)[x] / primitive; nth is the inverse permutation.
__device__ __forceinline__ uint32_t gf2_mul(uint32_t a, uint32_t b, uint32_t num_vars, uint32_t primitive) {
  uint64_t r = 0;
  for (uint32_t i = 0; i < num_vars; i++)
    if ((b >> i) & 1u) r ^= (uint64_t)a << i;
  for (int i = 2 * (int)num_vars - 2; i >= (int)num_vars; i--)
    if ((r >> i) & 1u) r ^= (uint64_t)primitive << (i - (int)num_vars);
  return (uint32_t)r;
}
__global__ void bh_order_kernel(uint32_t num_vars, uint32_t primitive, uint32_t* __restrict__ order,
                                uint32_t* __restrict__ nth) {
  const size_t n = (size_t)1 << num_vars, chunks = (n + 63) / 64;
  GSTRIDE(c, chunks) {
    // x^(64 c - 1) by square-and-multiply, then 64 steps of "times x"
    size_t k = c * 64;
    uint32_t b = 1;
    if (k > 0) {
      uint32_t sq = 2u;  // x (a second chunk exists only when num_vars >= 7)
      for (size_t e = k - 1; e; e >>= 1) {
        if (e & 1) b = gf2_mul(b, sq, num_vars, primitive);
        sq = gf2_mul(sq, sq, num_vars, primitive);
      }
    }
    for (size_t j = 0; j < 64 && k + j < n; j++) {
      const size_t kk = k + j;
      uint32_t v;
      if (kk == 0) {
        v = 0;
      } else {
        v = b;
        uint64_t t = (uint64_t)b << 1;
        b = (uint32_t)(t ^ ((t >> num_vars) * primitive));
      }
      order[kk] = v;
      nth[v] = (uint32_t)kk;
    }
  }
}
void k_bh_order(Ctx& c, size_t num_vars, uint32_t primitive, uint32_t* order, uint32_t* nth) {
  const size_t chunks = (((size_t)1 << num_vars) + 63) / 64;
  hipLaunchKernelGGL(bh_order_kernel, grid_for(chunks, 64), 64, 0, c.stream, (uint32_t)num_vars, primitive, order, nth);
}

__global__ void rotate_gather_kernel(const Fr* __restrict__ poly, size_t n, int rot, uint32_t num_vars, uint32_t primitive,
                                     uint32_t x_inv, Fr* __restrict__ out) {
  GSTRIDE(b, n) out[b] = poly[bh_rotate((uint32_t)b, rot, num_vars, primitive, x_inv)];
}
void k_rotate_gather(Ctx& c, const Fr* poly, size_t num_vars, int rot, uint32_t primitive, uint32_t x_inv, Fr* out) {
  size_t n = (size_t)1 << num_vars;
  ProfScope ps(c, "rotate_gather", 64.0 * n, 0.0, (double)n);
  hipLaunchKernelGGL(rotate_gather_kernel, grid_for(n), 256, 0, c.stream, poly, n, rot, (uint32_t)num_vars, primitive,
                     x_inv, out);
}

__global__ void identity_table_kernel(size_t n, Fr* __restrict__ out) {
  GSTRIDE(b, n) out[b] = from_u64<FrParams>(b);
}
void k_identity_table(Ctx& c, size_t n, Fr* out) {
  hipLaunchKernelGGL(identity_table_kernel, grid_for(n), 256, 0, c.stream, n, out);
}
// this rank's shard of the identity table (dev.hpp Shard): local index (hi || lo) holds the global row (hi, rank, lo)
__global__ void identity_table_shard_kernel(size_t n_local, unsigned j, unsigned rho, size_t rank, Fr* __restrict__ out) {
  GSTRIDE(i, n_local) {
    const size_t lo = i & (((size_t)1 << j) - 1), hi = i >> j;
    out[i] = from_u64<FrParams>((hi << (j + rho)) | (rank << j) | lo);
  }
}
void k_identity_table_shard(Ctx& c, size_t n_local, size_t j, size_t rho, size_t rank, Fr* out) {
  hipLaunchKernelGGL(identity_table_shard_kernel, grid_for(n_local), 256, 0, c.stream, n_local, (unsigned)j, (unsigned)rho, rank, out);
}
__global__ void set_one_at_kernel(Fr* p, size_t idx) { p[idx] = Fr::one(); }
void k_one_hot_table(Ctx& c, size_t n, size_t hot, Fr* out) {
  LH_HIP(hipMemsetAsync(out, 0, n * sizeof(Fr), c.stream));
  hipLaunchKernelGGL(set_one_at_kernel, 1, 1, 0, c.stream, out, hot);
}

// ------------------------------------------------------------------ sum-check round, monomials in device memory
template <bool BIND>
__device__ __forceinline__ void load_pair_ext(const Fr* __restrict__ in, Fr* __restrict__ out, size_t b, const Fr& r,
                                              bool store, Fr& v0, Fr& v1) {
  if (BIND) {
    const Fr* p = in + 4 * b;
    Fr e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
    v0 = add(mul(sub(e1, e0), r), e0);
    v1 = add(mul(sub(e3, e2), r), e2);
    if (store) {
      out[2 * b] = v0;
      out[2 * b + 1] = v1;
    }
  } else {
    v0 = in[2 * b];
    v1 = in[2 * b + 1];
  }
}

template <int D, bool BIND>
__global__ __launch_bounds__(256) void sc_round_ext_kernel(ExtRound rd, size_t size, uint32_t tp,
                                                           Fr* __restrict__ partials, ScFinishArgs fin) {
  __shared__ Fr lds[4];
  __shared__ int is_last;
  Fr acc[D];
#pragma unroll
  for (int x = 0; x < D; x++) acc[x] = Fr::zero();
  const size_t items = size * tp;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < items; w += (size_t)gridDim.x * blockDim.x) {
    const size_t b = tp == 1 ? w : w / tp;
    const uint32_t m_lo = tp == 1 ? 0u : (uint32_t)(w % tp);
    const uint32_t m_hi = tp == 1 ? rd.num_terms : m_lo + 1u;
    for (uint32_t m = m_lo; m < m_hi; m++) {
      Fr pm[D];
      const uint32_t o0 = rd.off[m], nf = rd.off[m + 1] - o0;
      if (nf == 0) {  // constant monomial
#pragma unroll
        for (int x = 0; x < D; x++) pm[x] = rd.coeff[m];
      }
      for (uint32_t k = 0; k < nf; k++) {
        const int t = rd.fac[o0 + k];
        Fr v0, v1;
        load_pair_ext<BIND>(rd.in[t], rd.out[t], b, rd.r, rd.store[o0 + k] != 0, v0, v1);
        if (k == 0) {
          if (!rd.is_one[m]) {
            v0 = mul(v0, rd.coeff[m]);
            v1 = mul(v1, rd.coeff[m]);
          }
          Fr step = sub(v1, v0);
          pm[0] = v1;
#pragma unroll
          for (int x = 1; x < D; x++) pm[x] = add(pm[x - 1], step);
        } else {
          Fr step = sub(v1, v0);
          Fr val = v1;
          pm[0] = mul(pm[0], val);
#pragma unroll
          for (int x = 1; x < D; x++) {
            val = add(val, step);
            pm[x] = mul(pm[x], val);
          }
        }
      }
#pragma unroll
      for (int x = 0; x < D; x++) acc[x] = add(acc[x], pm[x]);
    }
  }
#pragma unroll
  for (int x = 0; x < D; x++) {
    Fr v = block_reduce_sum(acc[x], lds);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * D + x] = v;
  }
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) publish_round(fin, D);
    return;
  }
  // last-workgroup final reduction (same protocol as kernels_sumcheck.hip::finish_round)
  if (threadIdx.x < 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t t = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == fin.last_ticket;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      is_last = last;
    }
  }
  __syncthreads();
  if (!is_last) return;
  const uint32_t blocks = gridDim.x;
#pragma unroll
  for (int x = 0; x < D; x++) {
    Fr a2 = Fr::zero();
    for (uint32_t i = threadIdx.x; i < blocks; i += blockDim.x) a2 = add(a2, partials[(size_t)i * D + x]);
    a2 = block_reduce_sum(a2, lds);
    if (threadIdx.x == 0) fin.out_host[x] = a2;
  }
  if (threadIdx.x == 0) publish_round(fin, D);
}

template <int D>
static void launch_ext(Ctx& c, const ExtRound& rd, bool bind, size_t size, uint32_t tp, unsigned grid, Fr* partials,
                       const ScFinishArgs& fin) {
  if (bind)
    hipLaunchKernelGGL((sc_round_ext_kernel<D, true>), dim3(grid), dim3(256), 0, c.stream, rd, size, tp, partials, fin);
  else
    hipLaunchKernelGGL((sc_round_ext_kernel<D, false>), dim3(grid), dim3(256), 0, c.stream, rd, size, tp, partials, fin);
}

void k_sc_round_ext(Ctx& c, const ExtRound& rd, int degree, bool bind, size_t size, Fr* evals_host) {
  LH_REQUIRE(degree >= 2 && degree <= 8, LH_ERR_ARG, "sum-check degree must be in 2..8");
  LH_REQUIRE(size >= 1, LH_ERR_ARG, "sum-check round over an empty table");
  const uint32_t seq = c.next_seq();
  ArenaScope scope(c.arena);
  // one thread per (pair, monomial) while that still fits a few waves per SIMD, else one thread per pair
  const uint32_t tp = (rd.num_terms > 1 && size * rd.num_terms <= ((size_t)1 << 19)) ? rd.num_terms : 1u;
  size_t g = (size * tp + 255) / 256;
  size_t cap = (size_t)c.num_cus * 8;
  if (g > cap) g = cap;
  evals_host = c.round_out(evals_host);  // (sharded rounds: the sums stay on the device, sumcheck.cpp)
  Fr* partials = g == 1 ? evals_host : c.arena.alloc_n<Fr>(g * degree);
  const ScFinishArgs fin = c.finish_for((uint32_t)g, evals_host, seq);
  {
    char name[40];
    snprintf(name, sizeof name, "sc_round_ext<%d,%s>%s", degree, bind ? "bind" : "first", tp > 1 ? "/tp" : "");
    ProfScope ps(c, name, (bind ? 192.0 : 64.0) * (double)size * rd.num_tables, 0, (double)size);
    switch (degree) {
      case 2: launch_ext<2>(c, rd, bind, size, tp, (unsigned)g, partials, fin); break;
      case 3: launch_ext<3>(c, rd, bind, size, tp, (unsigned)g, partials, fin); break;
      case 4: launch_ext<4>(c, rd, bind, size, tp, (unsigned)g, partials, fin); break;
      case 5: launch_ext<5>(c, rd, bind, size, tp, (unsigned)g, partials, fin); break;
      case 6: launch_ext<6>(c, rd, bind, size, tp, (unsigned)g, partials, fin); break;
      case 7: launch_ext<7>(c, rd, bind, size, tp, (unsigned)g, partials, fin); break;
      default: launch_ext<8>(c, rd, bind, size, tp, (unsigned)g, partials, fin); break;
    }
  }
  c.wait_round(seq);
}

// ------------------------------------------------------------------ sum-check round as a register program
// One wave per evaluation point X (blockDim = 64 * D), one lane per pair; the register file lives in LDS
// (word w of register r of thread t at ((r * 8 + w) * blockDim + t): conflict-free), operands are registers,
// constants or tables read at X: hi + (X - 1)(hi - lo).  ~35 multiplications per point for the vanilla+lookup
// zero-check instead of ~160 in expanded-monomial form.
__device__ __forceinline__ Fr prog_reg_load(const uint32_t* regs, uint32_t r, uint32_t nthreads) {
  Fr v;
#pragma unroll
  for (int w = 0; w < 8; w++) v.l[w] = regs[(r * 8 + w) * nthreads + threadIdx.x];
  return v;
}
__device__ __forceinline__ void prog_reg_store(uint32_t* regs, uint32_t r, uint32_t nthreads, const Fr& v) {
#pragma unroll
  for (int w = 0; w < 8; w++) regs[(r * 8 + w) * nthreads + threadIdx.x] = v.l[w];
}
__device__ __forceinline__ Fr prog_operand(const ProgRound& pr, const uint32_t* regs, uint32_t kind, uint32_t idx,
                                           size_t b, int xm1, uint32_t nthreads) {
  if (kind == PROG_REG) return prog_reg_load(regs, idx, nthreads);
  if (kind == PROG_CONST) return pr.consts[idx];
  if (kind == PROG_PAIR) return pr.in[idx][b];
  const Fr* t = pr.in[idx];
  const Fr lo = t[2 * b], hi = t[2 * b + 1];
  Fr v = hi;
  if (xm1 > 0) {
    const Fr step = sub(hi, lo);
    for (int k = 0; k < xm1; k++) v = add(v, step);
  }
  return v;
}

__global__ __launch_bounds__(512) void sc_round_prog_kernel(ProgRound pr, size_t size, int D, Fr* __restrict__ partials,
                                                            ScFinishArgs fin) {
  extern __shared__ uint32_t prog_regs[];
  __shared__ int is_last;
  const uint32_t nthreads = blockDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // wave = X - 1
  // the instruction stream lives in LDS behind the register file: a scalar load from global memory per
  // instruction (~200 ns) would cost as much as the arithmetic it decodes
  uint32_t* lds_code = prog_regs + (size_t)pr.num_regs * 8 * nthreads;
  for (uint32_t i = threadIdx.x; i < 2 * pr.num_instrs; i += nthreads) lds_code[i] = pr.code[i];
  __syncthreads();
  Fr acc = Fr::zero();
  for (size_t base = (size_t)blockIdx.x * 64; base < size; base += (size_t)gridDim.x * 64) {
    const size_t b = base + lane;
    if (b < size) {
      for (uint32_t i = 0; i < pr.num_instrs; i++) {
        const uint32_t w0 = lds_code[2 * i], w1 = lds_code[2 * i + 1];
        const uint32_t op = w0 & 15u, dst = (w0 >> 4) & 15u;
        const Fr a = prog_operand(pr, prog_regs, (w0 >> 8) & 3u, w1 & 0xffffu, b, wave, nthreads);
        Fr res;
        if (op == PROG_NEG) {
          res = sub(Fr::zero(), a);
        } else if (op == PROG_MOV) {
          res = a;
        } else {
          const Fr bb = prog_operand(pr, prog_regs, (w0 >> 10) & 3u, w1 >> 16, b, wave, nthreads);
          res = op == PROG_MUL ? mul(a, bb) : op == PROG_ADD ? add(a, bb) : sub(a, bb);
        }
        prog_reg_store(prog_regs, dst, nthreads, res);
      }
      acc = add(acc, prog_reg_load(prog_regs, pr.result_reg, nthreads));
    }
  }
  acc = wave_reduce_sum(acc);
  if (gridDim.x == 1) {
    if (lane == 0) {
      fin.out_host[wave] = acc;
      __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) publish_round(fin, D);
    return;
  }
  // (the partial sums travel as self-validating lanes when the launch has a lane buffer - resident.cuh fin_put: no fence)
  if (lane == 0) {
    fin_put(fin, partials, (size_t)blockIdx.x * D + wave, acc);
    if (!fin.lanes) __threadfence();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = fin.lanes ? __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = t == fin.last_ticket;
  }
  __syncthreads();
  if (!is_last) return;
  if (!fin.lanes) __threadfence();
  Fr a2 = Fr::zero();
  for (uint32_t i = lane; i < gridDim.x; i += 64) a2 = add(a2, fin_get(fin, partials, (size_t)i * D + wave));
  a2 = wave_reduce_sum(a2);
  if (lane == 0) {
    fin.out_host[wave] = a2;
    __threadfence_system();
  }
  __syncthreads();
  if (threadIdx.x == 0) publish_round(fin, D);
}

// (`degree`: the evaluation points X = 1..degree of this launch - the expression's degree, or one fewer in an eq-factored round)
void k_sc_round_prog(Ctx& c, const ProgRound& pr, int degree, size_t size, Fr* evals_host, const JitKernel* jit) {
  LH_REQUIRE(degree >= 1 && degree <= 8, LH_ERR_ARG, "sum-check round: 1..8 evaluation points");
  LH_REQUIRE(size >= 1 && pr.num_regs >= 1 && pr.num_regs <= PROG_MAX_REGS, LH_ERR_ARG, "sum-check program: bad shape");
  const uint32_t seq = c.next_seq();
  ArenaScope scope(c.arena);
  const unsigned threads = 64u * (unsigned)degree;
  const size_t lds_bytes = (size_t)pr.num_regs * 32 * threads + (size_t)pr.num_instrs * 8;
  LH_REQUIRE(lds_bytes <= 150 * 1024, LH_ERR_ARG, "sum-check program: too large for LDS");
  c.opt_in_lds((const void*)sc_round_prog_kernel, 160 * 1024 - 64);
  size_t g = jit ? (size + 255) / 256 : (size + 63) / 64;  // compiled form: 4 groups of 64 pairs per workgroup
  // one resident set of workgroups (they loop over the pairs): a grid of 4 per CU when only 3 fit runs a second, mostly
  // empty pass
  int per_cu = 4;
  if (jit) {
    per_cu = (int)jit_blocks_per_cu(jit);
  } else {
    static int vm_per_cu[9] = {0};
    if (!vm_per_cu[degree]) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sc_round_prog_kernel, (int)threads, lds_bytes) != hipSuccess || n < 1) n = 1;
      vm_per_cu[degree] = n;
    }
    per_cu = vm_per_cu[degree];
  }
  static const bool dbg = getenv("LH_HP_DEBUG") != nullptr;
  if (dbg && size >= ((size_t)1 << 18)) fprintf(stderr, "[expr] round kernel: %d workgroups of %u threads per CU\n", per_cu, threads);
  // compiled form: one wave per workgroup and a grid of (g, degree) workgroups, all of which draw a ticket
  const size_t cap = jit ? std::max<size_t>(1, (size_t)c.num_cus * (size_t)per_cu / (size_t)degree) : (size_t)c.num_cus * (size_t)per_cu;
  if (g > cap) g = cap;
  evals_host = c.round_out(evals_host);  // (sharded rounds: the sums stay on the device, sumcheck.cpp)
  // (compiled form: a workgroup is four groups of 64 pairs, each with a partial sum per point)
  Fr* partials = (g == 1 && !jit) ? evals_host : c.arena.alloc_n<Fr>((jit ? 4 : 1) * g * degree);
  const ScFinishArgs fin = c.finish_for((uint32_t)(jit ? g * degree : g), evals_host, seq);
  // (the compiled kernel draws a ticket even when it is the launch's only workgroup - one point of an eq-factored round over a
  //  few pairs -, which finish_for does not count)
  if (jit && g * (size_t)degree == 1) c.ticket_base += 1;
  {
    char name[40];
    snprintf(name, sizeof name, jit ? "sc_round_jit<%d>" : "sc_round_prog<%d>", degree);
    ProfScope ps(c, name, 64.0 * (double)size * pr.num_tables, 0, (double)size);
    if (jit)
      jit_launch(c, jit, pr, (unsigned)degree, (unsigned)g, size, partials, fin);
    else
      hipLaunchKernelGGL(sc_round_prog_kernel, dim3((unsigned)g), dim3(threads), lds_bytes, c.stream, pr, size, degree,
                         partials, fin);
  }
  c.wait_round(seq);
}

// ------------------------------------------------------------------ plain pair sums of a few tables (the zero-check's linear part)
__global__ __launch_bounds__(256) void lin_sums_kernel(LinSums ls, size_t size, Fr* __restrict__ partials, ScFinishArgs fin) {
  __shared__ Fr lds[4];
  __shared__ int is_last;
  Fr ev = Fr::zero(), od = Fr::zero();
  for (uint32_t i = 0; i < ls.count; i++) {
    const Fr* __restrict__ t = ls.t[i];
    Fr e = Fr::zero(), o = Fr::zero();
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < size; b += (size_t)gridDim.x * blockDim.x) {
      e = add(e, t[2 * b]);
      o = add(o, t[2 * b + 1]);
    }
    ev = add(ev, mul(e, ls.coeff[i]));
    od = add(od, mul(o, ls.coeff[i]));
  }
  ev = block_reduce_sum(ev, lds);
  od = block_reduce_sum(od, lds);
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) {
      fin.out_host[0] = ev, fin.out_host[1] = od;
      publish_round(fin, 2);
    }
    return;
  }
  if (threadIdx.x == 0) {
    fin_put(fin, partials, (size_t)blockIdx.x * 2, ev), fin_put(fin, partials, (size_t)blockIdx.x * 2 + 1, od);
  }
  if (!fin_ticket(fin, &is_last)) return;
  for (int x = 0; x < 2; x++) {
    Fr a2 = Fr::zero();
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x) a2 = add(a2, fin_get(fin, partials, (size_t)i * 2 + x));
    a2 = block_reduce_sum(a2, lds);
    if (threadIdx.x == 0) fin.out_host[x] = a2;
  }
  if (threadIdx.x == 0) publish_round(fin, 2);
}
void k_lin_sums(Ctx& c, const LinSums& ls, size_t size, Fr* out_host) {
  LH_REQUIRE(ls.count >= 1 && ls.count <= (uint32_t)LIN_MAX_TABLES && size >= 1, LH_ERR_ARG, "lin_sums: bad shape");
  const size_t g = std::min<size_t>((size + 255) / 256, (size_t)c.num_cus * 4);
  // (nobody waits for this launch alone: the round's kernel behind it publishes the sequence number the host waits for;
  //  the partials live as long as the caller's arena scope - the round's)
  Fr* partials = c.arena.alloc_n<Fr>(2 * g);
  const uint32_t seq = c.next_seq();
  const ScFinishArgs fin = c.finish_for((uint32_t)g, c.round_out(out_host), seq);
  ProfScope ps(c, "lin_sums", 64.0 * (double)size * ls.count, 0, (double)size);
  hipLaunchKernelGGL(lin_sums_kernel, dim3((unsigned)g), dim3(256), 0, c.stream, ls, size, partials, fin);
}

// ------------------------------------------------------------------ row-wise evaluation of a monomial list
// out[b] = sum_m coeff_m * prod_k atom_{m,k}(b), atoms read at (rotated) row b
__global__ void expr_rows_kernel(RowsExpr e, size_t n, Fr* __restrict__ out) {
  GSTRIDE(b, n) {
    Fr acc = Fr::zero();
    for (uint32_t m = 0; m < e.num_terms; m++) {
      Fr v = e.coeff[m];
      for (uint32_t k = e.off[m]; k < e.off[m + 1]; k++) {
        const RowsAtom a = e.atoms[e.fac[k]];
        Fr f;
        if (a.kind == ROWS_ATOM_POLY) {
          f = a.table[a.rot == 0 ? b : bh_rotate((uint32_t)b, a.rot, e.num_vars, e.primitive, e.x_inv)];
        } else if (a.kind == ROWS_ATOM_IDENTITY) {
          f = from_u64<FrParams>(b);
        } else {  // Lagrange: 1 on its row
          f = (b == a.hot) ? Fr::one() : Fr::zero();
        }
        v = mul(v, f);
      }
      acc = add(acc, v);
    }
    out[b] = acc;
  }
}
void k_expr_rows(Ctx& c, const RowsExpr& e, size_t n, Fr* out) {
  ProfScope ps(c, "expr_rows", 32.0 * n * (e.num_terms + 1), 2.0 * n * e.num_terms, (double)n);
  if (n) hipLaunchKernelGGL(expr_rows_kernel, grid_for(n), 256, 0, c.stream, e, n, out);
}

}  // namespace lh
