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
#include <stdlib.h>
#include <algorithm>
#include "dev.hpp"
#include "reduce.cuh"
#include "resident.cuh"

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

// Final cross-workgroup reduction WITHOUT a second launch.  Every workgroup has stored its D partial sums
// (threads of wave 0, plain stores); it then draws a ticket from a counter that only ever grows (the host
// knows its value before the launch, so nothing has to be zeroed).  The workgroup that draws the last
// ticket of the launch re-reads all partials, sums them and publishes the result to the host.
// Hand-off protocol: cdna_hip_programming.md Guideline 16 (agent-scope release by every producer,
// agent-scope acquire by the one consumer, never placement-dependent).
typedef ScFinishArgs ScFinish;

template <int D>
__device__ __forceinline__ void finish_round(const ScFinish& f, const Fr* __restrict__ partials, Fr* lds) {
  __shared__ int is_last;
  if (!fin_ticket(f, &is_last)) return;
  const uint32_t blocks = gridDim.x;
#pragma unroll
  for (int x = 0; x < D; x++) {
    Fr acc = Fr::zero();
    for (uint32_t i = threadIdx.x; i < blocks; i += blockDim.x) acc = add(acc, fin_get(f, partials, (size_t)i * D + x));
    acc = block_reduce_sum(acc, lds);
    if (threadIdx.x == 0) f.out_host[x] = acc;
  }
  if (threadIdx.x == 0) publish_round(f, D);
}

// Occupancy of the streaming round kernels: the degree-2 instantiations (GKR layers, Surge, the batch opening: the
// ones that run at 2^24) are asked for 4 waves per SIMD instead of the 3 the register allocator settles on by itself
// (measured 2^24 AND proof: -0.9 ms, tools/ab_waves.sh); higher degrees would spill heavily and keep the default.
#ifndef LH_SC_WAVES_D2
#define LH_SC_WAVES_D2 4
#endif
#define LH_SC_WAVES_ATTR(D) \
  __attribute__((amdgpu_waves_per_eu((D) <= 2 && LH_SC_WAVES_D2 ? LH_SC_WAVES_D2 : 1, (D) <= 2 && LH_SC_WAVES_D2 ? LH_SC_WAVES_D2 : 8)))
template <int D, bool BIND>
__global__ __launch_bounds__(256) LH_SC_WAVES_ATTR(D) void sc_round_kernel(ScArgs a, size_t size, uint32_t tp,
                                                       Fr* __restrict__ partials, ScFinish fin) {
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
      if (nf == 0) {  // a constant (times the eq factor below)
#pragma unroll
        for (int x = 0; x < D; x++) pm[x] = rd.coeff_is_one[m] ? Fr::one() : rd.coeff[m];
      }
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
    if (rd.eq_level) {  // eq factoring: one entry per pair, the same at every X
      const Fr e = rd.eq_level[b];
#pragma unroll
      for (int x = 0; x < D; x++) s[x] = mul(s[x], e);
    } else if (rd.global_eq >= 0) {
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
    if (threadIdx.x == 0) fin_put(fin, partials, (size_t)blockIdx.x * D + x, v);
  }
  if (gridDim.x == 1) {  // single workgroup: `partials` IS the host buffer
    if (threadIdx.x == 0) publish_round(fin, D);
    return;
  }
  finish_round<D>(fin, partials, lds);
}

// ------------------------------------------------------------------ one lane per BOUND ENTRY (the degree-2 factored rounds)
// The streaming kernel above gives a lane a whole pair: in BIND mode it reads 4 consecutive entries (128 B) per table, so
// every dwordx4 of a wave touches 64 different cache lines - the rounds moved exactly their algorithmic bytes (PMC) at
// 2.3-3.7 TB/s while the plain bind kernel, one lane per output entry (64 B per lane), streams at 5.2-6.2 TB/s.  Here lane
// i owns bound entry i = (pair i / 2, side i & 1): it reads in[2i], in[2i+1], binds and stores out[i] exactly like the
// bind kernel, and evaluates ONE point of q: the odd lane holds v1 = the pair's value at X = 1 as it is; the even lane
// holds v0 and takes v1 from its neighbour (8 DPP moves per table) for the value at X = 2, 2 v1 - v0.  Same products per
// pair as before, spread over two lanes; half the registers per lane.
__device__ __forceinline__ Fr dpp_xor1(const Fr& v) {  // the value of lane ^ 1 (quad_perm [1,0,3,2])
  Fr o;
#pragma unroll
  for (int i = 0; i < 8; i++) o.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], 0xB1, 0xF, 0xF, true);
  return o;
}
template <bool BIND>
__device__ __forceinline__ Fr load_entry(const Fr* __restrict__ in, Fr* __restrict__ out, size_t i, const Fr& r, bool store) {
  if (BIND) {
    const Fr e0 = in[2 * i], e1 = in[2 * i + 1];
    const Fr v = add(mul(sub(e1, e0), r), e0);
    if (store) out[i] = v;
    return v;
  }
  return in[i];
}
// the pair's value at this lane's point: X = 1 on odd lanes (v1), X = 2 on even lanes (2 v1 - v0)
__device__ __forceinline__ Fr at_lane_point(const Fr& v, bool odd) {
  const Fr o = dpp_xor1(v);
  return odd ? v : sub(dbl(o), v);
}
// masked block reductions: even lanes' sum, then odd lanes' sum
__device__ __forceinline__ void reduce_by_parity(const Fr& acc, bool odd, Fr* lds, Fr& even_sum, Fr& odd_sum) {
  even_sum = block_reduce_sum(odd ? Fr::zero() : acc, lds);
  odd_sum = block_reduce_sum(odd ? acc : Fr::zero(), lds);
}

// q(1), q(2) of  sum_b eq_level[b] * sum_m coeff_m prod_k table_{m,k}  (ScRound with eq_level): partials[.][0] = q(1)
// occupancy hints of the entry-per-lane kernels (waves per SIMD; 0 = the register allocator's choice): A/B knobs at build time
#ifndef LH_E_WAVES_E2
#define LH_E_WAVES_E2 0
#endif
#ifndef LH_E_WAVES_OPEN
#define LH_E_WAVES_OPEN 0
#endif
#ifndef LH_E_WAVES_RW
#define LH_E_WAVES_RW 0
#endif
#define LH_E_WAVES_ATTR(W) __attribute__((amdgpu_waves_per_eu((W) ? (W) : 1, (W) ? (W) : 8)))
template <bool BIND>
__global__ __launch_bounds__(256) LH_E_WAVES_ATTR(LH_E_WAVES_E2) void sc_round_e2_kernel(ScArgs a, size_t size, Fr* __restrict__ partials, ScFinish fin) {
  __shared__ Fr lds[4];
  const ScRound& rd = a.rd;
  Fr acc = Fr::zero();
  const size_t items = 2 * size;
  const bool odd = threadIdx.x & 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t)gridDim.x * blockDim.x) {
    Fr s = Fr::zero();
    for (uint32_t m = 0; m < rd.num_terms; m++) {
      const int nf = rd.nfac[m];
      Fr pm = rd.coeff_is_one[m] ? Fr::one() : rd.coeff[m];  // (nf == 0: a constant)
      for (int k = 0; k < nf; k++) {
        const int t = rd.fac[m][k];
        const Fr val = at_lane_point(load_entry<BIND>(rd.in[t], rd.out[t], i, rd.r, a.store[m][k] != 0), odd);
        pm = k == 0 ? (rd.coeff_is_one[m] ? val : mul(val, rd.coeff[m])) : mul(pm, val);
      }
      s = add(s, pm);
    }
    acc = add(acc, mul(s, rd.eq_level[i >> 1]));
  }
  Fr q2, q1;
  reduce_by_parity(acc, odd, lds, q2, q1);
  if (threadIdx.x == 0) {
    fin_put(fin, partials, (size_t)blockIdx.x * 2, q1);
    fin_put(fin, partials, (size_t)blockIdx.x * 2 + 1, q2);
  }
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) publish_round(fin, 2);
    return;
  }
  finish_round<2>(fin, partials, lds);
}

// The product-pair shape (ScRound::pp): q(1), q(2) of sum_b eq_level[b] sum_m c_m l_m r_m, one lane per bound entry as
// above, the products of four terms in ONE Montgomery reduction (ff.cuh dot_scan: 4 x 64 + 65 multiply-adds instead of
// 4 x 129) and the coefficients folded into the left factors at the first bind (ScRound::pp == 2: l'_m = c_m l_m is
// stored; before that - round 0 - they are applied on the way, afterwards they are one).  Per lane of an 8-tree layer:
// 16 binds + 2 four-term reductions (~5 products' worth) + the eq entry instead of 16 + 16 + 1.
// Loads first: the entries of a PAIR of terms (four tables, eight 32-byte entries per lane when binding) are requested
// before anything is computed or stored - written table by table (load, bind, store, next table) the stores keep the
// compiler from moving the next table's loads up, and a wave has one table's 4 KB in flight at a time.
#ifndef LH_PP_WAVES
#define LH_PP_WAVES 3  // (build-time A/B knob, tools/ab_pp_waves.sh: 4 fits 128 registers with 100 B of spills)
#endif
template <bool BIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LH_PP_WAVES, 8))) void sc_round_pp_kernel(ScArgs a, size_t size, Fr* __restrict__ partials, ScFinish fin) {
  __shared__ Fr lds[4];
  const ScRound& rd = a.rd;
  Fr acc = Fr::zero();
  const size_t items = 2 * size;
  const bool odd = threadIdx.x & 1;
  const bool fold = BIND && rd.pp == 2;
  const uint32_t K = rd.num_terms;
  const Fr rch = rd.r;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t)gridDim.x * blockDim.x) {
    Fr s = Fr::zero();
#ifndef LH_PP_GROUP
#define LH_PP_GROUP 4  // terms per shared reduction (build-time A/B knob: 2 = a reduction per pair of terms, fewer registers)
#endif
    for (uint32_t m0 = 0; m0 < K; m0 += LH_PP_GROUP) {
      Fr av[LH_PP_GROUP], bv[LH_PP_GROUP];
#pragma unroll
      for (int h2 = 0; h2 < LH_PP_GROUP / 2; h2++) {
        // the four tables of terms m0 + 2 h2, m0 + 2 h2 + 1 (a term past the end reads term 0's tables again, unused)
        const uint32_t ma = m0 + 2 * h2, mb = ma + 1;
        const bool va = ma < K, vb = mb < K;
        const int t0 = rd.fac[va ? ma : 0][0], t1 = rd.fac[va ? ma : 0][1], t2 = rd.fac[vb ? mb : 0][0], t3 = rd.fac[vb ? mb : 0][1];
        const Fr* __restrict__ p0 = rd.in[t0];
        const Fr* __restrict__ p1 = rd.in[t1];
        const Fr* __restrict__ p2 = rd.in[t2];
        const Fr* __restrict__ p3 = rd.in[t3];
        Fr x0, x1, x2, x3;
        if (BIND) {
          const Fr e00 = p0[2 * i], e01 = p0[2 * i + 1], e10 = p1[2 * i], e11 = p1[2 * i + 1];
          const Fr e20 = p2[2 * i], e21 = p2[2 * i + 1], e30 = p3[2 * i], e31 = p3[2 * i + 1];
          x0 = add(mul(sub(e01, e00), rch), e00);
          x1 = add(mul(sub(e11, e10), rch), e10);
          x2 = add(mul(sub(e21, e20), rch), e20);
          x3 = add(mul(sub(e31, e30), rch), e30);
        } else {
          x0 = p0[i], x1 = p1[i], x2 = p2[i], x3 = p3[i];
        }
        // a folding round stores l' = c l; any other binding round stores the bound entries as they are and applies a
        // coefficient that is still there to the lane's value only (sc_pp_fold = 2 runs every round that way)
        if (BIND && !fold) {
          if (va) rd.out[t0][i] = x0, rd.out[t1][i] = x1;
          if (vb) rd.out[t2][i] = x2, rd.out[t3][i] = x3;
        }
        if (va && !rd.coeff_is_one[ma]) x0 = mul(x0, rd.coeff[ma]);
        if (vb && !rd.coeff_is_one[mb]) x2 = mul(x2, rd.coeff[mb]);
        if (BIND && fold) {
          if (va) rd.out[t0][i] = x0, rd.out[t1][i] = x1;
          if (vb) rd.out[t2][i] = x2, rd.out[t3][i] = x3;
        }
        av[2 * h2] = va ? at_lane_point(x0, odd) : Fr::zero();
        bv[2 * h2] = va ? at_lane_point(x1, odd) : Fr::zero();
        av[2 * h2 + 1] = vb ? at_lane_point(x2, odd) : Fr::zero();
        bv[2 * h2 + 1] = vb ? at_lane_point(x3, odd) : Fr::zero();
      }
      s = add(s, dot<FrParams, LH_PP_GROUP>(av, bv));
    }
    acc = add(acc, mul(s, rd.eq_level[i >> 1]));
  }
  Fr q2, q1;
  reduce_by_parity(acc, odd, lds, q2, q1);
  if (threadIdx.x == 0) {
    fin_put(fin, partials, (size_t)blockIdx.x * 2, q1);
    fin_put(fin, partials, (size_t)blockIdx.x * 2 + 1, q2);
  }
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) publish_round(fin, 2);
    return;
  }
  finish_round<2>(fin, partials, lds);
}

// ------------------------------------------------------------------ small / medium rounds: LDS-staged
// The late rounds of every sum-check (and whole GKR layers near the root) are latency-bound: few pairs,
// but one thread walking a whole term is a chain of ~15 dependent field multiplications (~1 us each
// on a lone wave).  Here a workgroup owns P pairs and works in three short phases:
//   1. every thread binds ONE or two table entries (1 multiplication) into LDS (and stores them),
//   2. every thread evaluates ONE (pair, term, X) item from LDS (nfac multiplications),
//   3. per (term, X) the P values are summed, scaled by the term's coefficient and combined.
// The dependent chain is ~1 + nfac + 1 multiplications whatever the expression size.
constexpr int LDS_VALS = 1280;   // bound entries held per workgroup (40 KB)
constexpr int LDS_ITEMS = 512;   // (pair, term, X) items per workgroup (16 KB)
struct ScLdsArgs {
  ScArgs a;
  uint8_t used[SC_MAX_TABLES];   // compact list of the tables the expression touches
  uint8_t slot_of[SC_MAX_TABLES];
  uint32_t num_used;
  uint32_t P;                    // pairs per workgroup
  uint32_t vals_entries;         // LDS carve-up: vals[vals_entries] then red[]
};

template <int D, bool BIND>
__global__ __launch_bounds__(256) void sc_round_lds_kernel(ScLdsArgs g, size_t size, Fr* __restrict__ partials,
                                                           ScFinish fin) {
  // sized by the launch (vals: 2 P entries per used table, at least the 4 reduction slots; red: one slot per item, at
  // least one per (term, X) group): a fixed 56 KB would cap the CU at two workgroups whatever P is
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const ScRound& rd = g.a.rd;
  const uint32_t P = g.P, U = g.num_used;
  Fr* vals = (Fr*)lds_raw;
  Fr* red = vals + g.vals_entries;
  const size_t b0 = (size_t)blockIdx.x * P;
  // phase 1: entry e = (slot u, local pair pl, which) -> vals[(u*P + pl)*2 + which]
  for (uint32_t e = threadIdx.x; e < U * P * 2; e += blockDim.x) {
    const uint32_t which = e & 1u, pl = (e >> 1) % P, u = (e >> 1) / P;
    const size_t b = b0 + pl;
    Fr v = Fr::zero();
    if (b < size) {
      const int t = g.used[u];
      if (BIND) {
        const Fr* p = rd.in[t] + 4 * b + 2 * which;
        Fr e0 = p[0], e1 = p[1];
        v = add(mul(sub(e1, e0), rd.r), e0);
        rd.out[t][2 * b + which] = v;
      } else {
        v = rd.in[t][2 * b + which];
      }
    }
    vals[e] = v;
  }
  __syncthreads();
  // phase 2: item it = ((m*D + x)*P + pl)
  const uint32_t items = rd.num_terms * D * P;
  for (uint32_t it = threadIdx.x; it < items; it += blockDim.x) {
    const uint32_t pl = it % P, mx = it / P, x = mx % D, m = mx / D;
    Fr acc = Fr::zero();
    if (b0 + pl < size) {
      const int nf = rd.nfac[m];
      for (int k = 0; k <= nf; k++) {
        int t;
        if (k < nf) t = rd.fac[m][k];
        else if (rd.global_eq >= 0) t = rd.global_eq;
        else break;
        const Fr* pv = vals + ((uint32_t)g.slot_of[t] * P + pl) * 2;
        Fr v0 = pv[0], v1 = pv[1];
        Fr val = v1;
        if (x > 0) {
          Fr step = sub(v1, v0);
          for (uint32_t j = 0; j < x; j++) val = add(val, step);
        }
        acc = k == 0 ? val : mul(acc, val);
      }
    }
    red[it] = acc;
  }
  __syncthreads();
  // phase 3a: per (term, X): sum over the pairs, times the coefficient
  const uint32_t groups = rd.num_terms * D;
  Fr gsum = Fr::zero();
  if (threadIdx.x < groups) {
    for (uint32_t pl = 0; pl < P; pl++) gsum = add(gsum, red[threadIdx.x * P + pl]);
    const uint32_t m = threadIdx.x / D;
    if (!rd.coeff_is_one[m]) gsum = mul(gsum, rd.coeff[m]);
  }
  __syncthreads();
  if (threadIdx.x < groups) red[threadIdx.x] = gsum;
  __syncthreads();
  // phase 3b: per X: sum over the terms
  if (threadIdx.x < D) {
    Fr s = Fr::zero();
    for (uint32_t m = 0; m < rd.num_terms; m++) s = add(s, red[m * D + threadIdx.x]);
    fin_put(fin, partials, (size_t)blockIdx.x * D + threadIdx.x, s);
    if (gridDim.x == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  }
  if (gridDim.x == 1) {  // single workgroup: `partials` IS the host buffer
    __syncthreads();
    if (threadIdx.x == 0) publish_round(fin, D);
    return;
  }
  finish_round<D>(fin, partials, vals);  // vals is free again: 4 slots of it serve as reduction scratch
}

template <int D>
static void launch_lds(Ctx& c, const ScLdsArgs& g, bool bind, size_t size, unsigned grid, Fr* partials,
                       const ScFinish& fin) {
  const size_t red_entries = std::max<size_t>((size_t)g.a.rd.num_terms * D * g.P, (size_t)g.a.rd.num_terms * D);
  const size_t lds = ((size_t)g.vals_entries + red_entries) * sizeof(Fr);
  if (bind)
    hipLaunchKernelGGL((sc_round_lds_kernel<D, true>), dim3(grid), dim3(256), lds, c.stream, g, size, partials, fin);
  else
    hipLaunchKernelGGL((sc_round_lds_kernel<D, false>), dim3(grid), dim3(256), lds, c.stream, g, size, partials, fin);
}

template <int D>
static void launch_round(Ctx& c, const ScArgs& a, bool bind, size_t size, uint32_t tp, unsigned grid, Fr* partials,
                         const ScFinish& fin) {
  if (bind)
    hipLaunchKernelGGL((sc_round_kernel<D, true>), dim3(grid), dim3(256), 0, c.stream, a, size, tp, partials, fin);
  else
    hipLaunchKernelGGL((sc_round_kernel<D, false>), dim3(grid), dim3(256), 0, c.stream, a, size, tp, partials, fin);
}

// ------------------------------------------------------------------ resident tail (dev.hpp: k_sc_tail_*)
// G workgroups (a power of two, 1 for short tables), each with a contiguous slice of every live table in LDS, run ALL
// remaining rounds:
//   evaluate: one (term, X, pair) item per lane, segmented butterfly over the pairs of a (term, X) group;
//   G > 1: the D sums of the slice go to device memory, a ticket is drawn, the workgroup with the last ticket adds the
//          G partials (finish_round's hand-off protocol);
//   the message goes to pinned memory as self-validating chunks (dev.hpp TailChunk);
//   every workgroup polls the host's mailbox for the challenge (bounded: ~2 s of the 100 MHz wall clock, or
//   SC_TAIL_ABORT) and binds its slice with it, LDS to LDS;
//   a slice that is down to one pair is bound into device memory instead, and the workgroup that arrives last
//   collects the G entries of every table and goes on alone.
// The dependent chain per round is 1 (bind) + nfac (+1 eq) + 1 (coefficient) multiplications; a workgroup of G > 1
// runs 4 waves (one per SIMD: a second wave on a SIMD doubles the time of every multiplication of the chain).
struct ScTailArgs {
  ScRound rd;
  uint32_t n0, first_bind, degree, num_out, seq0, red_off;
  uint32_t G, lgG, cap;    // workgroups; LDS carve-up: cur[T * cap], nxt[T * cap / 2], red[]
  uint32_t ticket_base;    // value of *ticket before this launch
  uint32_t* ticket;
  Fr* part;                // device scratch: [2][G][8] partial sums of the even / odd rounds, then hand[T][G]
  TailChunk* bcast;        // device: the challenge, relayed by the workgroup that talks to the host (G > 1)
  uint32_t* flag;
  TailChunk* msg_host;
  Fr* out_host;
  const TailMbox* mbox;
  uint64_t poll_ticks;  // wall_clock64 ticks the kernel waits for one challenge before it leaves
  uint64_t* trace;      // development (LH_SC_TAIL_TRACE): [round][8] wall-clock stamps of workgroup 0 / the last workgroup
};
constexpr uint32_t TAIL_LDS_BYTES = 144 * 1024;      // of the CU's 160 KB (opt-in: hipFuncAttributeMaxDynamicSharedMemorySize)
constexpr uint32_t TAIL_THREADS = 512;               // single workgroup
constexpr uint32_t TAIL_THREADS_MULTI = 256;         // G > 1
constexpr uint32_t TAIL_MAX_ITEMS = 2048;            // (term, X, pair) items of a workgroup's first round: ~2 us per 512 on one CU
constexpr uint32_t TAIL_MAX_G = 64;

__global__ __launch_bounds__(TAIL_THREADS) void sc_tail_kernel(ScTailArgs a) {
  extern __shared__ __align__(16) unsigned char tail_lds_raw[];
  __shared__ Fr r_sh;
  __shared__ uint32_t stop_sh, last_sh;
  Fr* lds = (Fr*)tail_lds_raw;
  const ScRound& rd = a.rd;
  const uint32_t T = rd.num_tables, D = a.degree, tid = threadIdx.x, lane = tid & 63u, nthr = blockDim.x;
  const uint32_t G = a.G, wg = blockIdx.x;
  Fr* cur = lds;                          // T * n entries
  Fr* nxt = lds + T * a.cap;              // T * n / 2 entries
  Fr* red = lds + a.red_off;
  Fr* hand = a.part + 2 * G * 8;
  bool multi = G > 1;
  uint32_t n = a.n0 >> a.lgG, lg = 0;     // this workgroup's slice: entries [wg * n, (wg + 1) * n) of every table
  while ((1u << lg) < n) lg++;
  for (uint32_t e = tid; e < T * n; e += nthr) {
    const uint32_t t = e >> lg, k = e & (n - 1);
    const size_t gk = (size_t)wg * n + k;
    Fr v;
    if (a.first_bind) {
      const Fr* p = rd.in[t] + 2 * gk;
      Fr e0 = p[0], e1 = p[1];
      v = add(mul(sub(e1, e0), rd.r), e0);
    } else {
      v = rd.in[t][gk];
    }
    cur[e] = v;
  }
  if (tid == 0) stop_sh = 0;
  __syncthreads();
  uint32_t rounds = lg + a.lgG, batch = 0;
  const uint32_t groups = rd.num_terms * D;
  for (uint32_t i = 0; i < rounds; i++, lg--) {
    const uint32_t P = n >> 1, lgP = lg - 1, items = groups << lgP;
    const uint32_t seg = P < 64 ? P : 64, chunks = P < 64 ? 1 : P >> 6;
    const bool tr0 = a.trace && tid == 0 && (wg == 0 || !multi);
    if (tr0) a.trace[i * 8 + 0] = wall_clock64();
    for (uint32_t base = tid & ~63u; base < items; base += nthr) {
      const uint32_t it = base + lane;
      const uint32_t pl = it & (P - 1), g = it >> lgP;
      Fr acc = Fr::zero();
      if (it < items) {
        const uint32_t x = g % D, m = g / D;
        const int nf = rd.nfac[m];
        for (int k = 0; k <= nf; k++) {
          int t;
          if (k < nf) t = rd.fac[m][k];
          else if (rd.global_eq >= 0) t = rd.global_eq;
          else break;
          const Fr* pv = cur + ((uint32_t)t << lg) + 2 * pl;
          Fr v0 = pv[0], v1 = pv[1];
          Fr val = v1;
          if (x > 0) {
            Fr step = sub(v1, v0);
            for (uint32_t j = 0; j < x; j++) val = add(val, step);
          }
          acc = k == 0 ? val : mul(acc, val);
        }
      }
      for (uint32_t off = 1; off < seg; off <<= 1) acc = add(acc, shfl_xor_fr(acc, (int)off));
      if (it < items && (lane & (seg - 1)) == 0) red[g * chunks + (pl >> 6)] = acc;
    }
    __syncthreads();
    Fr gs = Fr::zero();
    if (tid < groups) {
      for (uint32_t k = 0; k < chunks; k++) gs = add(gs, red[tid * chunks + k]);
      const uint32_t m = tid / D;
      if (!rd.coeff_is_one[m]) gs = mul(gs, rd.coeff[m]);
    }
    __syncthreads();
    if (tid < groups) red[tid] = gs;
    __syncthreads();
    if (tr0) a.trace[i * 8 + 1] = wall_clock64();
    const uint32_t seq = a.seq0 + i;
    if (!multi) {
      if (tid < D) {
        Fr s = Fr::zero();
        for (uint32_t m = 0; m < rd.num_terms; m++) s = add(s, red[m * D + tid]);
        tail_send(a.msg_host, tid, s, seq);
      }
    } else {
      // partial sums of this slice -> device memory, ticket; the last workgroup adds the G partials (one wave per X,
      // butterfly over the workgroups) and sends the message
      Fr* part = a.part + (size_t)(i & 1u) * G * 8;
      if (tid < 64) {
        if (tid < D) {
          Fr s = Fr::zero();
          for (uint32_t m = 0; m < rd.num_terms; m++) s = add(s, red[m * D + tid]);
          part[wg * 8 + tid] = s;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) last_sh = tail_ticket(a.ticket, a.ticket_base + batch * G + G - 1) ? 1u : 0u;
      }
      batch++;
      __syncthreads();
      if (tr0) a.trace[i * 8 + 2] = wall_clock64();
      if (last_sh && a.trace && tid == 0) a.trace[i * 8 + 3] = wall_clock64();
      if (last_sh) {
        for (uint32_t x = tid >> 6; x < D; x += nthr >> 6) {
          Fr v = lane < G ? part[lane * 8 + x] : Fr::zero();
          for (uint32_t off = 1; off < 64 && off < G; off <<= 1) v = add(v, shfl_xor_fr(v, (int)off));
          if (lane == 0) tail_send(a.msg_host, x, v, seq);
        }
      }
    }
    if (a.trace && tid == 0 && (multi ? last_sh != 0 : true)) a.trace[i * 8 + 4] = wall_clock64();
    if (tid < 64) {
      // lanes 0..2 fetch one chunk of the mailbox each; the challenge is there when all three carry this round's seq.
      // G > 1: only the workgroup that sent the message asks the host (dozens of pollers queue up on the PCIe link:
      // measured 4 us of skew between workgroups); it relays the chunks through device memory to the others.
      const bool relay = multi && last_sh;
      const TailChunk* box = multi && !last_sh ? a.bcast : a.mbox->c;
      const uint64_t t0 = wall_clock64();
      uint32_t stop = 0;
      u32x4 v = {0u, 0u, 0u, 0u};
      for (;;) {
        if (lane < 3) v = load_sys_x4(&box[lane]);
        const uint64_t ok = __ballot(lane >= 3 || v.x == seq);
        if (ok == ~0ull) break;
        const uint64_t ab = __ballot(lane < 3 && v.x == SC_TAIL_ABORT);
        if (ab || wall_clock64() - t0 > a.poll_ticks) {
          stop = 1;
          break;
        }
      }
      if (relay && lane < 3) store_sys_x4((void*)&a.bcast[lane], stop ? u32x4{SC_TAIL_ABORT, 0u, 0u, 0u} : v);
      if (lane < 3) {
        r_sh.l[3 * lane] = v.y, r_sh.l[3 * lane + 1] = v.z;
        if (lane < 2) r_sh.l[3 * lane + 2] = v.w;
      }
      if (lane == 0) stop_sh = stop;
    }
    __syncthreads();
    if (stop_sh) return;  // the host gave up (or went away): leave without publishing anything further
    if (tr0) a.trace[i * 8 + 5] = wall_clock64();
    const Fr r = r_sh;
    if (multi && P == 1) {
      // hand-over: one bound entry per table and workgroup -> device memory; the last arrival goes on alone with
      // tables of G entries
      if (tid < 64) {
        for (uint32_t t = tid; t < T; t += 64) {
          const Fr v0 = cur[2 * t], v1 = cur[2 * t + 1];
          hand[(size_t)t * G + wg] = add(mul(sub(v1, v0), r), v0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) last_sh = tail_ticket(a.ticket, a.ticket_base + batch * G + G - 1) ? 1u : 0u;
      }
      __syncthreads();
      if (!last_sh) return;
      cur = lds, nxt = lds + T * a.cap;  // (the slice may sit in the smaller of the two regions by now)
      for (uint32_t e = tid; e < T * G; e += nthr) cur[e] = hand[e];
      __syncthreads();
      multi = false;
      n = G;
      lg = a.lgG + 1;  // (the loop header takes one off)
      continue;
    }
    for (uint32_t e = tid; e < (T << lgP); e += nthr) {
      const uint32_t t = e >> lgP, k = e & (P - 1);
      const Fr* pv = cur + (t << lg) + 2 * k;
      Fr v0 = pv[0], v1 = pv[1];
      nxt[e] = add(mul(sub(v1, v0), r), v0);
    }
    __syncthreads();
    Fr* tmp = cur;
    cur = nxt;
    nxt = tmp;
    n = P;
  }
  if (tid < a.num_out) {
    a.out_host[tid] = cur[tid];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  }
  __syncthreads();
  if (tid == 0) publish_flag(a.flag, a.seq0 + rounds);
}

static uint32_t tail_red_entries(const ScRound& rd, int degree, size_t n) {
  const size_t groups = (size_t)rd.num_terms * degree, P = n >> 1;
  return (uint32_t)std::max(groups, groups * (P < 64 ? 1 : P >> 6));
}

// workgroups of a resident tail over tables of n0 entries: slices of ~32 entries (about one multiplication per lane and
// round), at most TAIL_MAX_G
static uint32_t tail_workgroups(size_t n0) {
  static const int forced = [] {
    const char* e = getenv("LH_SC_TAIL_G");  // development: 1 = always a single workgroup
    return e ? atoi(e) : 0;
  }();
  uint32_t G = 1;
  while (G < TAIL_MAX_G && (size_t)G * 64 <= n0) G <<= 1;
  if (forced > 0) {
    G = 1;
    while (G < (uint32_t)forced && (size_t)G * 4 <= n0) G <<= 1;
  }
  return G;
}

static bool tail_fits(const ScRound& rd, int degree, size_t n0, uint32_t G) {
  const size_t s0 = n0 / G, cap = std::max<size_t>(s0, G);
  const size_t fr = (size_t)rd.num_tables * (cap + cap / 2) + tail_red_entries(rd, degree, cap);
  if (fr * sizeof(Fr) > TAIL_LDS_BYTES) return false;
  return cap <= 2 || (size_t)rd.num_terms * degree * (cap >> 1) <= TAIL_MAX_ITEMS;
}

size_t k_sc_tail_capacity(const Ctx& c, const ScRound& rd, int degree) {
  const bool enabled = c.opt.sc_tail != 0;  // 0: one launch per round all the way down
  const size_t max_len = (size_t)c.opt.sc_tail_max_len;
  if (!enabled || degree < 1 || degree > SC_TAIL_MAX_DEGREE) return 0;
  if ((size_t)rd.num_terms * degree > 256 || rd.num_tables == 0) return 0;
  size_t best = 0;
  for (size_t n0 = 2; n0 <= max_len; n0 <<= 1) {
    if (!tail_fits(rd, degree, n0, tail_workgroups(n0))) break;
    best = n0;
  }
  return best;
}

void k_sc_tail_launch(Ctx& c, const ScRound& rd, int degree, size_t n0, bool first_bind, size_t num_out, uint32_t seq0,
                      TailChunk* msg_host, Fr* out_host) {
  const uint32_t G = tail_workgroups(n0);
  LH_REQUIRE(n0 >= 2 && (n0 & (n0 - 1)) == 0 && n0 <= k_sc_tail_capacity(c, rd, degree) && tail_fits(rd, degree, n0, G),
             LH_ERR_ARG, "sum-check tail: tables do not fit");
  LH_REQUIRE(num_out <= 256, LH_ERR_ARG, "sum-check tail: too many outputs");
  ScTailArgs a;
  a.rd = rd;
  a.n0 = (uint32_t)n0;
  a.first_bind = first_bind ? 1 : 0;
  a.degree = (uint32_t)degree;
  a.num_out = (uint32_t)num_out;
  a.seq0 = seq0;
  a.G = G;
  a.lgG = 0;
  while ((1u << a.lgG) < G) a.lgG++;
  a.cap = (uint32_t)std::max<size_t>(n0 / G, G);
  a.red_off = (uint32_t)((size_t)rd.num_tables * (a.cap + a.cap / 2));
  a.ticket = c.ticket;
  a.ticket_base = c.ticket_base;
  a.part = nullptr;
  a.bcast = nullptr;
  if (G > 1) {
    // one batch of G tickets per distributed round, one for the hand-over
    uint32_t rounds_a = 0;
    while (((size_t)G << (rounds_a + 1)) <= n0) rounds_a++;
    c.ticket_base += (rounds_a + 1) * G;
    a.part = c.arena.alloc_n<Fr>((size_t)2 * G * 8 + (size_t)rd.num_tables * G);
    // (the ctx's own relay chunks: every round's sequence number is new, so they are never cleared between tails - only
    // after an aborted one, whose abort marker would stop the next tail: k_sc_tail_resync)
    a.bcast = (TailChunk*)(c.ticket + 32);
  }
  a.flag = c.flag;
  a.msg_host = msg_host;
  a.out_host = out_host;
  a.mbox = c.mbox();
  // bounded wait per challenge: LH_SC_TAIL_TIMEOUT_MS (default 2000) in ticks of the device's constant-rate counter
  // (hipDeviceAttributeWallClockRate, 100 MHz on gfx950).  When it expires the host resumes with launched rounds.
  const char* tmo = getenv("LH_SC_TAIL_TIMEOUT_MS");
  const double ms = tmo && *tmo ? atof(tmo) : 2000.0;
  a.poll_ticks = (uint64_t)(ms * (double)c.wall_clock_khz);
  static const bool trace_on = getenv("LH_SC_TAIL_TRACE") != nullptr;
  a.trace = nullptr;
  if (trace_on) {
    a.trace = (uint64_t*)c.arena.alloc(32 * 8 * sizeof(uint64_t));
    LH_HIP(hipMemsetAsync(a.trace, 0, 32 * 8 * sizeof(uint64_t), c.stream));
    c.tail_trace = a.trace;
  }
  c.mbox_send(Fr::zero(), 0u);
  const size_t lds = ((size_t)a.red_off + tail_red_entries(rd, degree, a.cap)) * sizeof(Fr);
  c.opt_in_lds((const void*)sc_tail_kernel, (int)TAIL_LDS_BYTES + 2048);
  hipLaunchKernelGGL(sc_tail_kernel, dim3(G), dim3(G > 1 ? TAIL_THREADS_MULTI : TAIL_THREADS), lds, c.stream, a);
  LH_HIP(hipGetLastError());
}

void k_sc_tail_resync(Ctx& c) {
  LH_HIP(hipMemsetAsync(c.ticket + 32, 0, 4 * sizeof(TailChunk), c.stream));  // the relay chunks (an abort marker may sit there)
  uint32_t v = 0;
  c.d2h(&v, c.ticket, sizeof(v));
  c.ticket_base = v;
}

static size_t sc_lds_max_items() {
  static const size_t lds_max = [] {
    const char* e = getenv("LH_SC_LDS_MAX_ITEMS");  // tuning knob: (pairs * terms) up to which the LDS kernel is used
    return e ? (size_t)atoll(e) : ((size_t)1 << 16);
  }();
  return lds_max;
}
bool k_sc_round_streams(const ScRound& rd, int degree, size_t size) {
  return !(rd.num_terms * (uint32_t)degree <= 256 && size * rd.num_terms <= sc_lds_max_items());
}

// Every level of a factored eq table (host.hpp EqFactoring: E_{j+1}[b] = E_j[2b] + E_j[2b+1]) in a few launches instead of
// one per round.  A wave takes a tile of 512 entries as four rows of 128: a lane loads one pair per row (64 contiguous
// bytes per lane: the plain bind kernel's access pattern), adds it (level 1), six rounds of lane shuffles make levels 2..7
// of every row, the four row totals levels 8 and 9 - one launch makes 9 levels, the next one starts from the last level of
// the first.  (One launch per round, in front of the round's kernel: 76 launches and ~1.3 ms of a 2^24 AND proof.)
struct EqLevelsPack {
  const Fr* in;
  Fr* out[9];
};
__global__ __launch_bounds__(256) void eq_levels_kernel(EqLevelsPack pk, size_t n_in, int nlev) {
  const int lane = threadIdx.x & 63;
  const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6, tiles = (n_in + 511) / 512;
  for (size_t tile = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; tile < tiles; tile += waves) {
    Fr v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const size_t i = tile * 512 + (size_t)r * 128 + 2 * (size_t)lane;
      v[r] = i + 1 < n_in ? add(pk.in[i], pk.in[i + 1]) : Fr::zero();
      if (nlev > 0 && (i >> 1) < (n_in >> 1)) pk.out[0][i >> 1] = v[r];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {  // levels 2..7: pairs, quads, ... of lanes inside a row
      const int lv = 1 + k;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        Fr o;
#pragma unroll
        for (int q = 0; q < 8; q++) o.l[q] = __shfl_xor(v[r].l[q], 1 << k, 64);
        v[r] = add(v[r], o);
        if (lv < nlev && (lane & ((2 << k) - 1)) == 0) {
          const size_t idx = (tile * 256 + (size_t)r * 64 + (size_t)lane) >> (k + 1);
          if (idx < (n_in >> (lv + 1))) pk.out[lv][idx] = v[r];
        }
      }
    }
    if (lane == 0) {  // the row totals: levels 8 and 9
      const Fr w0 = add(v[0], v[1]), w1 = add(v[2], v[3]);
      if (nlev > 7) {
        if (tile * 2 < (n_in >> 8)) pk.out[7][tile * 2] = w0;
        if (tile * 2 + 1 < (n_in >> 8)) pk.out[7][tile * 2 + 1] = w1;
      }
      if (nlev > 8 && tile < (n_in >> 9)) pk.out[8][tile] = add(w0, w1);
    }
  }
}
// levels[k] (k < nlev) = the pair sums of levels[k - 1], levels[-1] = `in` of n_in entries (a power of two)
void k_eq_levels(Ctx& c, const Fr* in, size_t n_in, Fr* const* levels, size_t nlev) {
  size_t done = 0;
  while (done < nlev && n_in >= 2) {
    const size_t k = std::min<size_t>(9, nlev - done);
    EqLevelsPack pk;
    pk.in = in;
    for (size_t i = 0; i < 9; i++) pk.out[i] = i < k ? levels[done + i] : nullptr;
    const size_t waves = (n_in + 511) / 512;
    hipLaunchKernelGGL(eq_levels_kernel, dim3((unsigned)std::min<size_t>((waves + 3) / 4, 8192)), dim3(256), 0, c.stream, pk,
                       n_in, (int)k);
    in = levels[done + k - 1];
    n_in >>= k;
    done += k;
  }
}

// workgroups per CU of the entry-per-lane round kernels (grid-stride loops; tuning knob LH_SC_ENTRY_BLOCKS)
static size_t sc_entry_blocks_per_cu() {
  static const size_t v = [] {
    const char* e = getenv("LH_SC_ENTRY_BLOCKS");
    return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)8;
  }();
  return v;
}

// grid of an entry-per-lane round kernel over `entries` bound entries: every workgroup ends with block reductions, a ticket
// and 8 lanes per partial sum for the finishing workgroup to collect, so a launch of few entries gives a lane
// LH_SC_ENTRIES_PER_LANE of them (default 4) rather than one - but never fewer than two workgroups per CU
static size_t sc_entry_grid(const Ctx& c, size_t entries) {
  static const size_t per_lane = [] {
    const char* e = getenv("LH_SC_ENTRIES_PER_LANE");
    return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)4;
  }();
  const size_t full = (entries + 255) / 256, cap = (size_t)c.num_cus * sc_entry_blocks_per_cu();
  size_t g = (entries + 256 * per_lane - 1) / (256 * per_lane);
  g = std::max(g, std::min(full, (size_t)c.num_cus * 2));
  return std::max<size_t>(1, std::min(std::min(g, full), cap));
}

// ------------------------------------------------------------------ batch-opening rounds with factored eq tables
// expression sum_m eq_m(x) * poly_m(x) (pcs/multilinear.rs:182-190): per term and pair one bind (2 multiplications),
// two products with the term's eq-level entry; the eq tables are neither read in full nor bound (sumcheck.cpp).
// (the number of terms is a template parameter: an accumulator array indexed by a run-time term count lives in scratch
// memory - 400 B per lane of spills doubled the kernel's HBM writes)
template <int M, bool BIND>
__global__ __launch_bounds__(256) LH_E_WAVES_ATTR(LH_E_WAVES_OPEN) void sc_round_open_kernel(ScOpenRound rd, size_t size, Fr* __restrict__ partials, ScFinish fin) {
  // one lane per bound entry (see sc_round_e2_kernel): q_m(0) = sum_b E_m[b] v0 comes from the even lanes, q_m(1) from the
  // odd ones - no exchange between lanes at all
  __shared__ Fr lds[4];
  constexpr int NQ = 2 * M;
  Fr acc[M];
#pragma unroll
  for (int m = 0; m < M; m++) acc[m] = Fr::zero();
  const size_t items = 2 * size;
  const bool odd = threadIdx.x & 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int m = 0; m < M; m++) {
      const Fr v = load_entry<BIND>(rd.in[m], rd.out[m], i, rd.r, true);
      acc[m] = add(acc[m], mul(rd.eq_level[m][i >> 1], v));
    }
  }
#pragma unroll
  for (int m = 0; m < M; m++) {
    Fr q0, q1;
    reduce_by_parity(acc[m], odd, lds, q0, q1);
    if (threadIdx.x == 0) {
      fin_put(fin, partials, (size_t)blockIdx.x * NQ + 2 * m, q0);
      fin_put(fin, partials, (size_t)blockIdx.x * NQ + 2 * m + 1, q1);
    }
  }
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) publish_round(fin, NQ);
    return;
  }
  finish_round<NQ>(fin, partials, lds);
}

template <int M>
static void launch_open(Ctx& c, const ScOpenRound& rd, bool bind, size_t size, unsigned g, Fr* partials, const ScFinish& fin) {
  if (bind)
    hipLaunchKernelGGL((sc_round_open_kernel<M, true>), dim3(g), dim3(256), 0, c.stream, rd, size, partials, fin);
  else
    hipLaunchKernelGGL((sc_round_open_kernel<M, false>), dim3(g), dim3(256), 0, c.stream, rd, size, partials, fin);
}

void k_sc_round_open(Ctx& c, const ScOpenRound& rd, bool bind, size_t size, Fr* out_host) {
  LH_REQUIRE(rd.num_terms >= 1 && rd.num_terms <= (uint32_t)SC_OPEN_MAX_TERMS && size >= 1, LH_ERR_ARG, "sc_round_open: bad shape");
  const uint32_t seq = c.next_seq();
  ArenaScope scope(c.arena);
  out_host = c.round_out(out_host);  // (sharded rounds: the sums stay on the device, sumcheck.cpp)
  size_t g = sc_entry_grid(c, 2 * size);
  const int nq = 2 * (int)rd.num_terms;
  Fr* partials = g == 1 ? out_host : c.arena.alloc_n<Fr>(g * nq);
  const ScFinish fin = c.finish_for((uint32_t)g, out_host, seq, bind ? 64.0 * (double)size * rd.num_terms : 0.0);
  {
    // algorithmic bytes: 96 B per bound entry of the polys (192 B per pair and term) + the eq-level entry
    ProfScope ps(c, bind ? "sc_round_open<bind>" : "sc_round_open<first>", ((bind ? 192.0 : 64.0) + 32.0) * (double)size * rd.num_terms,
                 (bind ? 4.0 : 2.0) * (double)size * rd.num_terms, (double)size);
    switch (rd.num_terms) {
      case 1: launch_open<1>(c, rd, bind, size, (unsigned)g, partials, fin); break;
      case 2: launch_open<2>(c, rd, bind, size, (unsigned)g, partials, fin); break;
      case 3: launch_open<3>(c, rd, bind, size, (unsigned)g, partials, fin); break;
      case 4: launch_open<4>(c, rd, bind, size, (unsigned)g, partials, fin); break;
      case 5: launch_open<5>(c, rd, bind, size, (unsigned)g, partials, fin); break;
      default: launch_open<6>(c, rd, bind, size, (unsigned)g, partials, fin); break;
    }
  }
  c.wait_round(seq);
}

// ------------------------------------------------------------------ grand-product layer over (A, A + 1) tree pairs
// FOLD (a BIND round): what is stored is l'_p = cs_p (l_p + k_p) and r'_p = r_p + k_p (binding is affine: the bound
// tables of l', r' ARE the folded bound tables), so that every later round is the plain product-pair shape sum_p l'_p r'_p
// and runs sc_round_pp_kernel (sumcheck.cpp divides cs out of, and takes k off, the final evaluations).  Either way the P
// products of a lane share one Montgomery reduction (ff.cuh dot): 2 P -> P + ~0.6 P products (fold round: the P
// multiplications by cs are the fold itself).
template <int P, bool BIND, bool FOLD>
__global__ __launch_bounds__(256) LH_E_WAVES_ATTR(LH_E_WAVES_RW) void sc_round_rw_kernel(ScRwRound rd, size_t size, Fr* __restrict__ partials, ScFinish fin) {
  // one lane per bound entry (see sc_round_e2_kernel): odd lanes evaluate X = 1, even lanes X = 2
  __shared__ Fr lds[4];
  Fr acc = Fr::zero();
  const size_t items = 2 * size;
  const bool odd = threadIdx.x & 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t)gridDim.x * blockDim.x) {
    Fr av[P], bv[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
      if (BIND && FOLD) {
        const Fr lf = mul(add(load_entry<true>(rd.l[p], rd.lo[p], i, rd.rchal, false), rd.k[p]), rd.cs[p]);
        const Fr rf = add(load_entry<true>(rd.r[p], rd.ro[p], i, rd.rchal, false), rd.k[p]);
        rd.lo[p][i] = lf, rd.ro[p][i] = rf;
        av[p] = at_lane_point(lf, odd), bv[p] = at_lane_point(rf, odd);  // (the lane point of an affine image is the image)
      } else {
        const Fr a = add(at_lane_point(load_entry<BIND>(rd.l[p], rd.lo[p], i, rd.rchal, true), odd), rd.k[p]);
        av[p] = mul(a, rd.cs[p]);
        bv[p] = add(at_lane_point(load_entry<BIND>(rd.r[p], rd.ro[p], i, rd.rchal, true), odd), rd.k[p]);
      }
    }
    acc = add(acc, mul(dot<FrParams, P>(av, bv), rd.eq_level[i >> 1]));
  }
  Fr q2, q1;
  reduce_by_parity(acc, odd, lds, q2, q1);
  if (threadIdx.x == 0) {
    fin_put(fin, partials, (size_t)blockIdx.x * 2, q1);
    fin_put(fin, partials, (size_t)blockIdx.x * 2 + 1, q2);
  }
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) publish_round(fin, 2);
    return;
  }
  finish_round<2>(fin, partials, lds);
}

template <int P>
static void launch_rw(Ctx& c, const ScRwRound& rd, bool bind, bool fold, size_t size, unsigned g, Fr* partials, const ScFinish& fin) {
  if (bind && fold)
    hipLaunchKernelGGL((sc_round_rw_kernel<P, true, true>), dim3(g), dim3(256), 0, c.stream, rd, size, partials, fin);
  else if (bind)
    hipLaunchKernelGGL((sc_round_rw_kernel<P, true, false>), dim3(g), dim3(256), 0, c.stream, rd, size, partials, fin);
  else
    hipLaunchKernelGGL((sc_round_rw_kernel<P, false, false>), dim3(g), dim3(256), 0, c.stream, rd, size, partials, fin);
}

void k_sc_round_rw(Ctx& c, const ScRwRound& rd, bool bind, size_t size, Fr* out_host, bool fold) {
  LH_REQUIRE(!fold || bind, LH_ERR_ARG, "sc_round_rw: folding is part of a binding round");
  LH_REQUIRE(rd.num_pairs >= 1 && rd.num_pairs <= (uint32_t)SC_RW_MAX_PAIRS && size >= 1 && rd.eq_level, LH_ERR_ARG,
             "sc_round_rw: bad shape");
  const uint32_t seq = c.next_seq();
  ArenaScope scope(c.arena);
  out_host = c.round_out(out_host);
  size_t g = sc_entry_grid(c, 2 * size);
  Fr* partials = g == 1 ? out_host : c.arena.alloc_n<Fr>(g * 2);
  const ScFinish fin = c.finish_for((uint32_t)g, out_host, seq, bind ? 64.0 * (double)size * 2.0 * rd.num_pairs : 0.0);
  {
    // algorithmic bytes: 96 B per bound entry of the 2 P tables read (192 B per pair and table) + the eq-level entry;
    // products per pair: 4 per tree pair (+ 4 binds), 2 for the eq factor
    const double P = (double)rd.num_pairs;
    ProfScope ps(c, bind ? "sc_round_rw<bind>" : "sc_round_rw<first>", ((bind ? 192.0 : 64.0) * 2.0 * P + 32.0) * (double)size,
                 ((bind ? 8.0 : 4.0) * P + 2.0) * (double)size, (double)size);
    switch (rd.num_pairs) {
      case 1: launch_rw<1>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
      case 2: launch_rw<2>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
      case 3: launch_rw<3>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
      case 4: launch_rw<4>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
      case 5: launch_rw<5>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
      case 6: launch_rw<6>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
      case 7: launch_rw<7>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
      default: launch_rw<8>(c, rd, bind, fold, size, (unsigned)g, partials, fin); break;
    }
  }
  c.wait_round(seq);
}

void k_sc_round(Ctx& c, const ScRound& rd, int degree, bool bind, size_t size, Fr* evals_host) {
  c.last_round_folded = false;  // (set below when the product-pair kernel ran a folding round: ScRound::pp == 2)
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
  // count the tables the expression touches
  size_t tabs = 0;
  for (int t = 0; t < SC_MAX_TABLES; t++) tabs += seen[t] ? 1 : 0;
  double nfac = 0, ncoef = 0;
  for (uint32_t m = 0; m < rd.num_terms; m++) nfac += rd.nfac[m], ncoef += rd.coeff_is_one[m] ? 0 : 2;
  const double muls_pair = std::max(0.0, nfac - rd.num_terms) * degree + ncoef + (rd.global_eq >= 0 || rd.eq_level ? degree : 0) +
                           (bind ? 2.0 * (nfac + (rd.global_eq >= 0 ? 1 : 0)) : 0.0);
  // algorithmic bytes: fused round = bind bytes only, 96 B per bound entry = 192 B per pair
  // and table; the unfused first round reads 64 B per pair and table.
  const double bytes = (bind ? 192.0 : 64.0) * (double)size * (double)tabs + (rd.eq_level ? 32.0 * (double)size : 0.0);
  const uint32_t seq = c.next_seq();
  ArenaScope scope(c.arena);
  // sharded rounds (sumcheck.cpp): the D sums stay on the device (all-gather and sum-and-publish follow on the stream)
  evals_host = c.round_out(evals_host);
  // (bytes a binding round stores: half of what it moves)
  auto finish = [&](size_t grid, bool streams = false) {
    return c.finish_for((uint32_t)grid, evals_host, seq, streams && bind ? bytes / 3.0 : 0.0);
  };

  // pairs per workgroup of the LDS-staged kernel
  uint32_t P = (uint32_t)std::min<size_t>(size, 64);
  P = std::min<uint32_t>(P, LDS_VALS / (2 * (uint32_t)tabs));
  P = std::min<uint32_t>(P, LDS_ITEMS / (rd.num_terms * degree));
  const bool use_lds = P >= 1 && !rd.eq_level && !k_sc_round_streams(rd, degree, size);
  if (use_lds) {
    // fill a workgroup's 256 threads in the item phase when there are pairs enough
    // ... and keep the bind phase at one entry per thread
    uint32_t want = (256 + rd.num_terms * degree - 1) / (rd.num_terms * degree);
    want = std::min<uint32_t>(want, std::max<uint32_t>(1, 256 / (2 * (uint32_t)tabs)));
    if (P > want && want >= 1) P = std::max<uint32_t>(want, 1);
    ScLdsArgs g;
    g.a = a;
    g.num_used = 0;
    for (int t = 0; t < SC_MAX_TABLES; t++)
      if (seen[t]) {
        g.slot_of[t] = (uint8_t)g.num_used;
        g.used[g.num_used++] = (uint8_t)t;
      }
    g.P = P;
    g.vals_entries = std::max<uint32_t>(g.num_used * P * 2, 8);
    size_t grid = (size + P - 1) / P;
    Fr* partials = grid == 1 ? evals_host : c.arena.alloc_n<Fr>(grid * degree);
    const ScFinish kflag = finish(grid);
    {
      char name[40];
      snprintf(name, sizeof name, "sc_round<%d,%s>/lds", degree, bind ? "bind" : "first");
      ProfScope ps(c, name, bytes, muls_pair * (double)size, (double)size);
      switch (degree) {
        case 1: launch_lds<1>(c, g, bind, size, (unsigned)grid, partials, kflag); break;
        case 2: launch_lds<2>(c, g, bind, size, (unsigned)grid, partials, kflag); break;
        case 3: launch_lds<3>(c, g, bind, size, (unsigned)grid, partials, kflag); break;
        case 4: launch_lds<4>(c, g, bind, size, (unsigned)grid, partials, kflag); break;
        case 5: launch_lds<5>(c, g, bind, size, (unsigned)grid, partials, kflag); break;
        default: launch_lds<6>(c, g, bind, size, (unsigned)grid, partials, kflag); break;
      }
    }
    c.wait_round(seq);
    return;
  }

  // large rounds: one thread per pair keeps everything in registers; in between, one thread per
  // (pair, term) so that a launch that cannot fill the chip is not also a long dependent chain
  static const size_t tp_max = [] {
    const char* e = getenv("LH_SC_TP_MAX_ITEMS");
    return e ? (size_t)atoll(e) : ((size_t)1 << 17);
  }();
  const uint32_t tp = (rd.num_terms > 1 && size * rd.num_terms <= tp_max) ? rd.num_terms : 1u;
  if (degree == 2 && rd.eq_level && tp == 1) {
    // the factored degree-2 rounds (GKR layers, Surge over one table): one lane per bound entry
    size_t g2 = sc_entry_grid(c, 2 * size);
    Fr* partials = g2 == 1 ? evals_host : c.arena.alloc_n<Fr>(g2 * 2);
    const ScFinish kflag = finish(g2, true);
    c.last_round_folded = rd.pp == 2 && bind;
    if (rd.pp) {
      // products per pair: binds, one per coefficient still applied on the way, ~0.62 per term for the shared reductions
      // (two points), the eq entry
      const double mp = (bind ? 2.0 * nfac : 0.0) + ncoef + 2.0 * 0.62 * rd.num_terms + 2.0;
      ProfScope ps(c, bind ? "sc_round_pp<bind>" : "sc_round_pp<first>", bytes, mp * (double)size, (double)size);
      if (bind) hipLaunchKernelGGL((sc_round_pp_kernel<true>), dim3((unsigned)g2), dim3(256), 0, c.stream, a, size, partials, kflag);
      else hipLaunchKernelGGL((sc_round_pp_kernel<false>), dim3((unsigned)g2), dim3(256), 0, c.stream, a, size, partials, kflag);
    } else {
      ProfScope ps(c, bind ? "sc_round<2,bind>" : "sc_round<2,first>", bytes, muls_pair * (double)size, (double)size);
      if (bind) hipLaunchKernelGGL((sc_round_e2_kernel<true>), dim3((unsigned)g2), dim3(256), 0, c.stream, a, size, partials, kflag);
      else hipLaunchKernelGGL((sc_round_e2_kernel<false>), dim3((unsigned)g2), dim3(256), 0, c.stream, a, size, partials, kflag);
    }
    c.wait_round(seq);
    return;
  }
  size_t g = (size * tp + 255) / 256;
  size_t cap = (size_t)c.num_cus * 4;
  if (g > cap) g = cap;
  Fr* partials = g == 1 ? evals_host : c.arena.alloc_n<Fr>(g * degree);
  const ScFinish kflag = finish(g, true);
  {
    char name[40];
    snprintf(name, sizeof name, "sc_round<%d,%s>%s", degree, bind ? "bind" : "first", tp > 1 ? "/tp" : "");
    ProfScope ps(c, name, bytes, muls_pair * (double)size, (double)size);
    switch (degree) {
      case 1: launch_round<1>(c, a, bind, size, tp, (unsigned)g, partials, kflag); break;
      case 2: launch_round<2>(c, a, bind, size, tp, (unsigned)g, partials, kflag); break;
      case 3: launch_round<3>(c, a, bind, size, tp, (unsigned)g, partials, kflag); break;
      case 4: launch_round<4>(c, a, bind, size, tp, (unsigned)g, partials, kflag); break;
      case 5: launch_round<5>(c, a, bind, size, tp, (unsigned)g, partials, kflag); break;
      default: launch_round<6>(c, a, bind, size, tp, (unsigned)g, partials, kflag); break;
    }
  }
  c.wait_round(seq);
}

}  // namespace lh
