// Sum-check round kernel: fused bind (previous challenge) + round-polynomial evaluation.
//
// Replaces the reference's per-round pair of passes
//   EvaluationsProver::evals   piop/sum_check/classic/eval.rs:102-131  (evaluate)
//   ProverState::next_round    piop/sum_check/classic.rs:90-141        (bind every table)
// and CoefficientsProver::karatsuba (classic/coeff.rs:153-202) for expressions of the shape
//   [eq *] sum_m coeff_m * prod_k table_{m,k}
// One thread owns one output pair b: in BIND mode it reads the 4 entries (4b..4b+3) of every table
// of the previous round, binds them with r (poly/multilinear.rs:609-617), stores the bound pair
// (2b, 2b+1) once and evaluates the new round polynomial on it - each table is read once and written
// once per round (96 B per bound pair instead of 160 B for separate bind and evaluate passes).
// Evaluation points X = 1..D: value(X) = v1 + (X-1)*(v1 - v0) (eval.rs:228-286); X = 0 is derived
// from the claim by the host (eval.rs:129).
#include <hip/hip_runtime.h>
#include "dev.hpp"
#include "reduce.cuh"

namespace lh {

struct ScArgs {
  ScRound rd;
  uint8_t store[LH_SC_MAX_TERMS][LH_SC_MAX_FACTORS];  // first occurrence of a table stores its bound pair
};

template <bool BIND>
__device__ __forceinline__ void load_pair(const Fr* __restrict__ in, Fr* __restrict__ out, size_t b, const Fr& r,
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

// `tp` (term parallelism) is 1 or num_terms: in the late, small rounds one thread per (pair, term)
// keeps the dependent chain of a thread at one term (~15 field multiplications) instead of the
// whole expression; the eq factor is linear, so it is applied per term before the reduction.
__device__ __forceinline__ void publish_flag(uint32_t* flag, uint32_t seq) {
  // results were written by this thread just before: release them to the host, then the sequence number
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int D, bool BIND>
__global__ __launch_bounds__(256) void sc_round_kernel(ScArgs a, size_t size, uint32_t tp,
                                                       Fr* __restrict__ partials, uint32_t* flag, uint32_t seq) {
  __shared__ Fr lds[4];
  const ScRound& rd = a.rd;
  Fr acc[D];
#pragma unroll
  for (int x = 0; x < D; x++) acc[x] = Fr::zero();

  const size_t items = size * tp;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < items; w += (size_t)gridDim.x * blockDim.x) {
    const size_t b = tp == 1 ? w : w / tp;
    const uint32_t m_lo = tp == 1 ? 0u : (uint32_t)(w % tp);
    const uint32_t m_hi = tp == 1 ? rd.num_terms : m_lo + 1u;
    Fr s[D];
#pragma unroll
    for (int x = 0; x < D; x++) s[x] = Fr::zero();
    for (uint32_t m = m_lo; m < m_hi; m++) {
      Fr pm[D];
      const int nf = rd.nfac[m];
      for (int k = 0; k < nf; k++) {
        const int t = rd.fac[m][k];
        Fr v0, v1;
        load_pair<BIND>(rd.in[t], rd.out[t], b, rd.r, a.store[m][k] != 0, v0, v1);
        if (k == 0) {
          if (!rd.coeff_is_one[m]) {
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
      for (int x = 0; x < D; x++) s[x] = add(s[x], pm[x]);
    }
    if (rd.global_eq >= 0) {
      Fr v0, v1;
      load_pair<BIND>(rd.in[rd.global_eq], rd.out[rd.global_eq], b, rd.r, m_lo == 0, v0, v1);
      Fr step = sub(v1, v0);
      Fr val = v1;
      s[0] = mul(s[0], val);
#pragma unroll
      for (int x = 1; x < D; x++) {
        val = add(val, step);
        s[x] = mul(s[x], val);
      }
    }
#pragma unroll
    for (int x = 0; x < D; x++) acc[x] = add(acc[x], s[x]);
  }
#pragma unroll
  for (int x = 0; x < D; x++) {
    Fr v = block_reduce_sum(acc[x], lds);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * D + x] = v;
  }
  if (flag && threadIdx.x == 0) publish_flag(flag, seq);  // single-workgroup launch: partials IS the host buffer
}

__global__ void sc_reduce_kernel(const Fr* __restrict__ partials, int blocks, int d, Fr* __restrict__ out,
                                 uint32_t* flag, uint32_t seq) {
  __shared__ Fr lds[4];
  for (int x = 0; x < d; x++) {
    Fr acc = Fr::zero();
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) acc = add(acc, partials[(size_t)i * d + x]);
    acc = block_reduce_sum(acc, lds);
    if (threadIdx.x == 0) out[x] = acc;
  }
  if (threadIdx.x == 0) publish_flag(flag, seq);
}

template <int D>
static void launch_round(Ctx& c, const ScArgs& a, bool bind, size_t size, uint32_t tp, unsigned grid, Fr* partials,
                         uint32_t* flag, uint32_t seq) {
  if (bind)
    hipLaunchKernelGGL((sc_round_kernel<D, true>), dim3(grid), dim3(256), 0, c.stream, a, size, tp, partials, flag, seq);
  else
    hipLaunchKernelGGL((sc_round_kernel<D, false>), dim3(grid), dim3(256), 0, c.stream, a, size, tp, partials, flag,
                       seq);
}

void k_sc_round(Ctx& c, const ScRound& rd, int degree, bool bind, size_t size, Fr* evals_host) {
  LH_REQUIRE(degree >= 1 && degree <= 6, LH_ERR_ARG, "sum-check degree must be in 1..6");
  LH_REQUIRE(size >= 1, LH_ERR_ARG, "sum-check round over an empty table");
  ScArgs a;
  a.rd = rd;
  bool seen[SC_MAX_TABLES] = {false};
  if (rd.global_eq >= 0) seen[rd.global_eq] = true;  // stored by the eq load
  for (uint32_t m = 0; m < rd.num_terms; m++)
    for (int k = 0; k < rd.nfac[m]; k++) {
      int t = rd.fac[m][k];
      a.store[m][k] = seen[t] ? 0 : 1;
      seen[t] = true;
    }
  // small rounds: one work item per (pair, term); a single workgroup writes the sums straight into the
  // pinned host buffer (no second kernel, no copy): the round trip is launch + kernel + one stream sync.
  const uint32_t tp = (rd.num_terms > 1 && size * rd.num_terms <= ((size_t)1 << 16)) ? rd.num_terms : 1u;
  const size_t items = size * tp;
  ArenaScope scope(c.arena);
  size_t g = (items + 255) / 256;
  size_t cap = (size_t)c.num_cus * 4;
  if (g > cap) g = cap;
  Fr* partials = g == 1 ? evals_host : c.arena.alloc_n<Fr>(g * degree);
  const uint32_t seq = c.next_seq();
  uint32_t* kflag = g == 1 ? c.flag : nullptr;
  {
    // algorithmic bytes: fused round = bind bytes only, 96 B per bound entry = 192 B per
    // pair and table; the unfused first round reads 64 B per pair and table.
    size_t tabs = 0;
    for (int t = 0; t < SC_MAX_TABLES; t++) tabs += seen[t] ? 1 : 0;
    double nfac = 0, ncoef = 0;
    for (uint32_t m = 0; m < rd.num_terms; m++) nfac += rd.nfac[m], ncoef += rd.coeff_is_one[m] ? 0 : 2;
    double muls_pair = (nfac - rd.num_terms) * degree + ncoef + (rd.global_eq >= 0 ? degree : 0) +
                       (bind ? 2.0 * (nfac + (rd.global_eq >= 0 ? 1 : 0)) : 0.0);
    char name[40];
    snprintf(name, sizeof name, "sc_round<%d,%s>%s", degree, bind ? "bind" : "first", tp > 1 ? "/tp" : "");
    ProfScope ps(c, name, (bind ? 192.0 : 64.0) * (double)size * (double)tabs, muls_pair * (double)size, (double)size);
  switch (degree) {
    case 1: launch_round<1>(c, a, bind, size, tp, (unsigned)g, partials, kflag, seq); break;
    case 2: launch_round<2>(c, a, bind, size, tp, (unsigned)g, partials, kflag, seq); break;
    case 3: launch_round<3>(c, a, bind, size, tp, (unsigned)g, partials, kflag, seq); break;
    case 4: launch_round<4>(c, a, bind, size, tp, (unsigned)g, partials, kflag, seq); break;
    case 5: launch_round<5>(c, a, bind, size, tp, (unsigned)g, partials, kflag, seq); break;
    default: launch_round<6>(c, a, bind, size, tp, (unsigned)g, partials, kflag, seq); break;
  }
  }
  if (g > 1)
    hipLaunchKernelGGL(sc_reduce_kernel, dim3(1), dim3(256), 0, c.stream, partials, (int)g, degree, evals_host, c.flag,
                       seq);
  c.wait_flag(seq);
}

}  // namespace lh
