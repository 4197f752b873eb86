// Resident grand-product layers: the layers of a product-tree argument whose tables fit on-chip run in ONE launch.
//
// Replaces, for the layers near the root of the memory-checking trees (every tree level of <= 2^15 nodes: 14 of the 23
// layers of a 2^24-lookup proof, all of them for small proofs), the per-layer sequence
//   eq_xy kernels -> [launched small rounds] -> sc_tail launch -> final evaluations
// i.e. the round loop piop/sum_check/classic.rs:208-240 inside the layer loop piop/gkr/fractional_sum_check.rs:146-181
// (product-only form, oracle/pyref/gkr.py::prove_grand_product).  The kernel keeps the LAYER loop inside:
//   per layer  wait for the host's layer message (batching coefficients c_k = lambda^k and the point y of the layer's eq
//              factor: the previous layer's challenges and mu) -> load this workgroup's slice of every tree level, the
//              coefficient folded into the left half (l'_k = c_k l_k) -> build the slice of the factored eq table in
//              registers (one wave: doubling by lane shuffles for the slice-local variables, a lane product for the
//              workgroup's own bits) -> rounds -> final evaluations to the host;
//   per round  q(X) = sum_b E_j[b] sum_k l'_k(X, b) r_k(X, b) at X = 1, 2 (the host rebuilds the reference's message
//              p(0..3) = S_j eq(y_j, X) q(X), host.hpp EqFactoring - identical bytes): one lane per (pair, X, four
//              trees), the four products in ONE Montgomery reduction (ff.cuh dot_scan), lane shuffles over the tree
//              groups, one product with the eq entry, a butterfly over the pairs; G > 1: partial sums and a ticket
//              through device memory, the last arrival sends; challenge from the host's mailbox (relayed through
//              device memory to the other workgroups); bind in place.
// Against the generic resident tail (kernels_sumcheck.hip sc_tail_kernel: (term, X, pair) items of two products each at
// three points, an eq table bound like any other) a round evaluates 2 x 4.5 instead of 3 x 32 products per pair of a
// 16-tree layer, and nothing is launched, staged or drained between layers.
// Two stages per layer: g workgroups with 2^s_log entries of every table each, then - the slices down to one pair -
// the bound entries and the workgroups' eq scalars go through device memory to the workgroup that arrives last, which
// finishes the layer alone (g <= 128 entries per table) and polls the host for the next layer's message.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "dev.hpp"
#include "reduce.cuh"
#include "resident.cuh"

namespace lh {

struct GkrResArgs {
  const GkrLayerDev* layers;  // device array, layers in the order they are proved (g never decreases)
  uint32_t num_layers;
  uint32_t cap;               // LDS carve-up: tab[2 * max_trees * cap], E[cap / 2], red[cap], cy[GKR_MAX_TREES + GKR_MAX_VARS]
  uint32_t max_trees;
  uint32_t ticket_base;       // value of *ticket before this launch
  uint32_t* ticket;
  Fr* part;                   // device: [2][G][2] partial sums of the even / odd rounds
  Fr* hand;                   // device: [2 * max_trees + 1][G] hand-over (bound entries, eq scalars)
  TailChunk* relay;           // device: chunks 0..2 the round challenge, 4.. the layer message, relayed by the polling workgroup
  const TailChunk* mbox_round;  // host (dev.hpp TailMbox)
  const TailChunk* mbox_layer;  // host: GKR_MSG_CHUNKS chunks
  TailChunk* msg_host;        // device -> host: q(1), q(2) as 6 chunks
  Fr* out_host;               // final evaluations (l'_k, r_k per tree), then the flag
  uint32_t* flag;
  uint64_t poll_ticks;
  uint64_t start_ticks;  // how long the workgroups wait for each other at the start before they give the launch up
  uint32_t* start_word;  // device: the start verdict of the launch in progress, launch id << 1 | failed
  uint32_t start_id;     // this launch's id (the first layer's sequence number: unique per launch on a ctx)
  uint64_t* trace;  // development (LH_GKR_TRACE): per layer 8 stamps, per round 8 stamps, of workgroup 0 / the sender
};
constexpr uint32_t GKR_TRACE_ROUNDS = 160;

__device__ __forceinline__ Fr shfl_fr(const Fr& v, int src) {
  Fr o;
#pragma unroll
  for (int i = 0; i < 8; i++) o.l[i] = __shfl(v.l[i], src, 64);
  return o;
}

__global__ __launch_bounds__(GKR_THREADS) void gkr_resident_kernel(GkrResArgs a) {
  extern __shared__ __align__(16) unsigned char gkr_lds_raw[];
  __shared__ Fr r_sh;
  __shared__ uint32_t stop_sh, last_sh;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, wg = blockIdx.x;
  Fr* tab = (Fr*)gkr_lds_raw;                     // table t at t * stride
  Fr* E = tab + (size_t)2 * a.max_trees * a.cap;  // the current level of the factored eq table: one entry per pair
  Fr* red = E + a.cap / 2 + 1;                    // [2][P] products of a round, hand-over staging
  Fr* cy = red + a.cap + 2;                       // coefficients c_k, then the point y
  // the chains of a resident round are latency-bound: ahead of whatever else shares the SIMDs (a helper ctx's MSM)
  __builtin_amdgcn_s_setprio(3);
  bool i_poll = wg == 0;   // this workgroup finished the previous layer: it asks the host for the next layer's message
  uint32_t tbase = a.ticket_base;
  if (tid == 0) stop_sh = 0;
  __syncthreads();
  if (gridDim.x > 1) {
    // check-in: the layers below hand work from workgroup to workgroup and only end when ALL of them run.  On a GPU shared
    // with other processes' resident kernels (several ranks on one device: the tests) a launch may stay partly
    // un-dispatched for as long as the others wait for THEIR missing workgroups; so everybody signs in first, and if the
    // roll is not complete within `start_ticks` the kernel leaves before the transcript has seen anything of it
    // (GKR_START_FAILED: the host takes the launched path for these layers instead).
    // The verdict is ONE word every workgroup agrees on (a.start_word = launch id << 1 | failed): the last arrival tries to
    // set "go", a workgroup whose patience runs out (or that sees the host's abort) tries to set "failed", a compare-and-
    // swap lets exactly one of them win, and everybody - late arrivals included - acts on what the word says.  (Deciding
    // from the ticket count and an abort marker, as round 4 did, let a workgroup that saw the roll complete run on while a
    // neighbour that had timed out a microsecond earlier was still writing its marker: messages of a launch that was
    // about to die could reach the transcript.)
    if (tid == 0) {
      const uint32_t id = (a.start_id & 0x3fffffffu) | 0x40000000u;  // (never 0: the word's initial state is "undecided")
      auto decide = [&](uint32_t failed) {  // returns the word as decided (by us or by somebody before us)
        uint32_t cur = __hip_atomic_load(a.start_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((cur >> 1) != id) {
          if (__hip_atomic_compare_exchange_strong(a.start_word, &cur, (id << 1) | failed, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT))
            return (id << 1) | failed;
        }
        return cur;
      };
      const uint32_t mine = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tbase;
      uint32_t word = 0;
      if (mine == gridDim.x - 1) {
        word = decide(0);  // the roll is complete
      } else {
        const uint64_t t0 = wall_clock64();
        for (;;) {
          word = __hip_atomic_load(a.start_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if ((word >> 1) == id) break;
          if (wall_clock64() - t0 > a.start_ticks || load_sys_x4(&a.relay[0]).x == SC_TAIL_ABORT) {
            word = decide(1);
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      const uint32_t failed = word & 1u;
      if (failed) {  // (every workgroup that leaves says so: idempotent)
        store_sys_x4((void*)&a.relay[0], u32x4{SC_TAIL_ABORT, 0u, 0u, 0u});
        store_sys_x4((void*)&a.relay[4], u32x4{SC_TAIL_ABORT, 0u, 0u, 0u});
        store_sys_x4((void*)&a.msg_host[0], u32x4{GKR_START_FAILED, 0u, 0u, 0u});
      }
      stop_sh = failed;
    }
    __syncthreads();
    if (stop_sh) return;
    tbase += gridDim.x;
  }
  for (uint32_t li = 0; li < a.num_layers; li++) {
    const GkrLayerDev& L = a.layers[li];
    const uint32_t h = L.h, B = L.B, g = L.g, slog = L.s_log, T = 2 * B, seq0 = L.seq, flags = L.flags;
    const uint32_t my_tbase = tbase;
    tbase += g > 1 ? (slog + 1) * g : 0;
    if (wg >= g) continue;  // (not one of this layer's workgroups; g never decreases, so it is not the poller either)
    // ---- the layer message: 3 chunks per field element, c_0 .. c_{B-1}, y_0 .. y_{h-1}
    if (flags & GKR_F_NOMSG) {  // (the tail of a sum-check: what a message would bring came with the descriptor)
      if (tid < B) cy[tid] = L.coef[tid];
      else if (tid >= GKR_MAX_TREES && tid < GKR_MAX_TREES + B) cy[tid] = L.koff[tid - GKR_MAX_TREES];
      __syncthreads();
    } else {
      const uint32_t nch = 3 * (B + h);
      const TailChunk* src = i_poll ? a.mbox_layer : a.relay + 4;
      u32x4 v = {0u, 0u, 0u, 0u};
      const uint64_t t0 = wall_clock64();
      for (;;) {
        bool ok = true;
        if (tid < nch) {
          v = load_sys_x4(&src[tid]);
          ok = v.x == seq0;
        }
        if (__syncthreads_and(ok)) break;
        const bool stop = (tid == 0 && (v.x == SC_TAIL_ABORT || wall_clock64() - t0 > a.poll_ticks));
        if (__syncthreads_or(stop)) {
          // (whoever gives up says so - the poller or a workgroup waiting for the relay: its peers and the host learn of
          // it from the markers, not from timeouts of their own)
          if (tid == 0) {
            store_sys_x4((void*)&a.relay[4], u32x4{SC_TAIL_ABORT, 0u, 0u, 0u});
            store_sys_x4((void*)&a.relay[0], u32x4{SC_TAIL_ABORT, 0u, 0u, 0u});
          }
          return;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (tid < nch) {
        if (i_poll) store_sys_x4((void*)&a.relay[4 + tid], v);
        const uint32_t f = tid / 3, j = tid % 3;
        cy[f].l[3 * j] = v.y, cy[f].l[3 * j + 1] = v.z;
        if (j < 2) cy[f].l[3 * j + 2] = v.w;
      }
      __syncthreads();
    }
    const bool tr0 = a.trace && tid == 0 && wg == 0;
    uint64_t* trl = a.trace ? a.trace + (size_t)li * 8 : nullptr;
    if (tr0) trl[0] = wall_clock64();  // layer message seen
    const Fr* yv = cy + B;
    // ---- stage 0: this workgroup's slice [wg * s, (wg + 1) * s) of every tree level; wave 3 builds the eq slice meanwhile
    uint32_t s = 1u << slog, stride = s;
    const size_t N = (size_t)1 << h;
    if (wave == 3 && (flags & GKR_F_EQ)) {
      for (uint32_t b = lane; b < (s >> 1); b += 64) E[b] = L.eq_level[(size_t)wg * (s >> 1) + b];
    } else if (wave == 3) {
      // slice-local variables 1 .. slog-1 by doubling (lane b ends with prod_i eq(y_i, bit_{i-1}(b))), the variables
      // slog .. h-1 are this workgroup's index bits
      Fr low = Fr::one();
      for (uint32_t i = slog; i-- > 1;) {
        const Fr parent = shfl_fr(low, (int)(lane >> 1));
        const Fr t = mul(parent, yv[i]);
        low = (lane & 1u) ? t : sub(parent, t);
      }
      Fr hi = Fr::one();
      if (lane < h - slog) {
        const Fr yi = yv[slog + lane];
        hi = ((wg >> lane) & 1u) ? yi : sub(Fr::one(), yi);
      }
      for (uint32_t off = 1; off < h - slog; off <<= 1) hi = mul(hi, shfl_xor_fr(hi, (int)off));
      hi = shfl_fr(hi, 0);
      if (lane < (s >> 1)) E[lane] = h - slog ? mul(hi, low) : low;
    } else {
      for (uint32_t e = tid; e < T * s; e += 192) {
        const uint32_t t = e >> slog, idx = e & (s - 1), k = t >> 1;
        const size_t gi = (size_t)wg * s + idx;
        const Fr* src = (flags & GKR_F_SPLIT) ? ((t & 1u) ? L.rv[k] : L.lv[k]) : L.lv[k] + ((t & 1u) ? N : 0);
        Fr v;
        if (flags & GKR_F_BIND) {
          const Fr e0 = src[2 * gi], e1 = src[2 * gi + 1];
          v = add(mul(sub(e1, e0), L.r_prev), e0);
        } else {
          v = src[gi];
        }
        if (flags & GKR_F_KOFF) v = add(v, cy[GKR_MAX_TREES + k]);
        if (!(t & 1u)) v = mul(v, cy[k]);
        tab[t * stride + idx] = v;
      }
    }
    __syncthreads();
    if (tr0) trl[1] = wall_clock64();  // tables loaded and folded, eq slice built
    bool multi = g > 1;
    const uint32_t clog = B <= 4 ? 0u : B <= 8 ? 1u : 2u, Cn = 1u << clog;
    uint32_t round = 0, batch = 0;
    bool finished = false, left = false;
    while (!finished) {
      const uint32_t P = s >> 1;
      // ---- evaluate: item (pair p, point xi, tree group c) = sum over the group's four trees of l'_k(X) r_k(X)
      const uint32_t items = (P * 2) << clog;
      for (uint32_t base = wave * 64; base < items; base += GKR_THREADS) {
        const uint32_t it = base + lane;
        const bool valid = it < items;
        const uint32_t c = it & (Cn - 1), xi = (it >> clog) & 1u, p = it >> (clog + 1);
        Fr av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t k = 4 * c + q;
          if (valid && k < B) {
            const Fr* pl = tab + (2 * k) * stride + 2 * p;
            const Fr* pr = tab + (2 * k + 1) * stride + 2 * p;
            const Fr l0 = pl[0], l1 = pl[1], r0 = pr[0], r1 = pr[1];
            av[q] = xi ? sub(dbl(l1), l0) : l1;
            bv[q] = xi ? sub(dbl(r1), r0) : r1;
          } else {
            av[q] = Fr::zero(), bv[q] = Fr::zero();
          }
        }
        Fr t = dot<FrParams, 4>(av, bv);
        if (clog >= 1) t = add(t, shfl_xor_fr(t, 1));
        if (clog >= 2) t = add(t, shfl_xor_fr(t, 2));
        if (valid && c == 0) red[xi * P + p] = mul(t, E[p]);
      }
      __syncthreads();
      const uint32_t seq = seq0 + 1 + round;
      // (trace rows are indexed by the sequence number's offset from the first layer's: one row per round of the launch)
      uint64_t* trq = a.trace ? a.trace + ((size_t)GKR_MAX_VARS + (seq - a.layers[0].seq)) * 8 : nullptr;
      if (a.trace && seq - a.layers[0].seq >= GKR_TRACE_ROUNDS) trq = nullptr;
      if (trq && tid == 0 && wg == 0) trq[0] = wall_clock64();  // evaluated
      if (wave == 0) {
        // lanes 0..31 sum the pairs at X = 1, lanes 32..63 at X = 2
        const uint32_t half = lane >> 5, l5 = lane & 31u;
        Fr v = Fr::zero();
        for (uint32_t p = l5; p < P; p += 32) v = add(v, red[half * P + p]);
        for (uint32_t off = 1; off < 32 && off < P; off <<= 1) v = add(v, shfl_xor_fr(v, (int)off));
        uint32_t last = 1;
        if (!multi) {
          if (l5 == 0) tail_send(a.msg_host, half, v, seq);
        } else {
          Fr* part = a.part + (size_t)(round & 1u) * g * 2;
          if (l5 == 0) part[wg * 2 + half] = v;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          last = 0;
          if (lane == 0) last = tail_ticket(a.ticket, my_tbase + batch * g + g - 1) ? 1u : 0u;
          last = __shfl(last, 0, 64);
          if (last) {
            Fr w = Fr::zero();
            for (uint32_t x = l5; x < g; x += 32) w = add(w, part[x * 2 + half]);
            for (uint32_t off = 1; off < 32 && off < g; off <<= 1) w = add(w, shfl_xor_fr(w, (int)off));
            if (l5 == 0) tail_send(a.msg_host, half, w, seq);
          }
        }
        // ---- the challenge: the workgroup that sent the message asks the host and relays it through device memory
        const bool talker = last != 0;
        if (trq && lane == 0 && wg == 0) trq[1] = wall_clock64();        // workgroup 0: ticket drawn / message sent
        if (trq && lane == 0 && talker) trq[2] = wall_clock64();         // the sender: message sent
        const TailChunk* box = talker ? a.mbox_round : a.relay;
        const uint64_t t0 = wall_clock64();
        uint32_t stop = 0;
        u32x4 v4 = {0u, 0u, 0u, 0u};
        for (;;) {
          if (lane < 3) v4 = load_sys_x4(&box[lane]);
          const uint64_t ok = __ballot(lane >= 3 || v4.x == seq);
          if (ok == ~0ull) break;
          const uint64_t ab = __ballot(lane < 3 && v4.x == SC_TAIL_ABORT);
          if (ab || wall_clock64() - t0 > a.poll_ticks) {
            stop = 1;
            break;
          }
        }
        if (talker && multi && lane < 3) store_sys_x4((void*)&a.relay[lane], stop ? u32x4{SC_TAIL_ABORT, 0u, 0u, 0u} : v4);
        if (talker && stop && lane == 0) store_sys_x4((void*)&a.relay[4], u32x4{SC_TAIL_ABORT, 0u, 0u, 0u});
        if (lane < 3) {
          r_sh.l[3 * lane] = v4.y, r_sh.l[3 * lane + 1] = v4.z;
          if (lane < 2) r_sh.l[3 * lane + 2] = v4.w;
        }
        if (lane == 0) stop_sh = stop;
        if (trq && lane == 0 && talker) trq[3] = wall_clock64();         // the sender: challenge seen
        if (trq && lane == 0 && wg == 0) trq[4] = wall_clock64();        // workgroup 0: challenge seen
      }
      if (multi) batch++;
      __syncthreads();
      if (stop_sh) return;
      const Fr r = r_sh;
      round++;
      if (multi && P == 1) {
        // ---- hand-over: one bound entry per table and this workgroup's eq scalar -> the last arrival goes on alone
        if (wave == 0) {
          if (lane < T) {
            const Fr v0 = tab[lane * stride], v1 = tab[lane * stride + 1];
            a.hand[(size_t)lane * g + wg] = add(mul(sub(v1, v0), r), v0);
          } else if (lane == T) {
            a.hand[(size_t)T * g + wg] = E[0];
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) last_sh = tail_ticket(a.ticket, my_tbase + batch * g + g - 1) ? 1u : 0u;
        }
        __syncthreads();
        if (!last_sh) {
          left = true;
          break;
        }
        for (uint32_t e = tid; e < (T + 1) * g; e += GKR_THREADS) {
          const Fr v = a.hand[e];
          if (e < T * g) tab[e] = v;  // (table t at t * g: the new stride)
          else red[e - T * g] = v;
        }
        __syncthreads();
        if (tid < (g >> 1)) E[tid] = add(red[2 * tid], red[2 * tid + 1]);
        __syncthreads();
        stride = g, s = g, multi = false;
        continue;
      }
      if (P == 1) {
        // ---- the layer's final evaluations: l'_k(x), r_k(x)
        if (tid < T) {
          const Fr v0 = tab[tid * stride], v1 = tab[tid * stride + 1];
          a.out_host[tid] = add(mul(sub(v1, v0), r), v0);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        }
        __syncthreads();
        if (tid == 0) publish_flag(a.flag, seq0 + h + 1);
        finished = true;
        break;
      }
      // ---- bind in place (every thread reads its inputs, then all write), next eq level by pair sums
      Fr outv[8];
      const uint32_t outputs = T * P, plog = 31u - (uint32_t)__clz(P);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t idx = tid + (uint32_t)i * GKR_THREADS;
        if (idx < outputs) {
          const uint32_t t = idx >> plog, e = idx & (P - 1);
          const Fr v0 = tab[t * stride + 2 * e], v1 = tab[t * stride + 2 * e + 1];
          outv[i] = add(mul(sub(v1, v0), r), v0);
        }
      }
      Fr en = Fr::zero();
      if (tid < (P >> 1)) en = add(E[2 * tid], E[2 * tid + 1]);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t idx = tid + (uint32_t)i * GKR_THREADS;
        if (idx < outputs) tab[(idx >> plog) * stride + (idx & (P - 1))] = outv[i];
      }
      if (tid < (P >> 1)) E[tid] = en;
      __syncthreads();
      if (trq && tid == 0 && wg == 0) trq[5] = wall_clock64();  // bound
      s = P;
    }
    i_poll = finished && !left;
  }
}

size_t k_gkr_resident_lds_bytes(uint32_t max_trees, uint32_t cap) {
  return ((size_t)2 * max_trees * cap + cap / 2 + 1 + cap + 2 + GKR_MAX_TREES + GKR_MAX_VARS) * sizeof(Fr);
}

// geometry of a resident layer over tables of 2^h entries: one workgroup while that is cheap (no ticket, no partial
// sums through device memory: ~5 us per round), else slices of 64 entries, 128 when that takes more than 128 workgroups
bool k_gkr_resident_geometry(uint32_t h, uint32_t* g, uint32_t* s_log) {
  if (h < 1 || h > GKR_MAX_VARS - 2) return false;
  if (h <= 7) {
    *g = 1, *s_log = h;
    return true;
  }
  uint32_t sl = 6;
  while ((1u << (h - sl)) > GKR_CAP) sl++;
  if ((1u << sl) > GKR_CAP) return false;
  *g = 1u << (h - sl), *s_log = sl;
  return true;
}

void k_gkr_resident_launch(Ctx& c, const GkrLayerDev* layers, size_t num_layers, TailChunk* msg_host, Fr* out_host) {
  LH_REQUIRE(num_layers >= 1, LH_ERR_ARG, "resident layers: none");
  uint32_t G = 1, max_trees = 1, tickets = 0, prev_g = 1;
  for (size_t i = 0; i < num_layers; i++) {
    const GkrLayerDev& L = layers[i];
    LH_REQUIRE(L.B >= 1 && L.B <= (uint32_t)GKR_MAX_TREES && L.h >= 1 && L.h <= (uint32_t)GKR_MAX_VARS - 2 && L.g >= prev_g &&
                   ((size_t)L.g << L.s_log) == ((size_t)1 << L.h) && (1u << L.s_log) <= GKR_CAP && L.g <= GKR_CAP && L.s_log >= 1,
               LH_ERR_ARG, "resident layers: bad geometry");
    prev_g = L.g;
    G = std::max(G, L.g);
    max_trees = std::max(max_trees, L.B);
    if (L.g > 1) tickets += (L.s_log + 1) * L.g;
  }
  if (G > 1) tickets += G;  // the check-in
  GkrResArgs a;
  GkrLayerDev* d_layers = (GkrLayerDev*)c.arena.alloc(num_layers * sizeof(GkrLayerDev));
  // (the descriptors travel through the ctx's pinned staging block: the caller's vector may die before the copy runs)
  GkrLayerDev* staged = (GkrLayerDev*)((char*)c.pin(65536) + 32768);
  LH_REQUIRE(num_layers * sizeof(GkrLayerDev) <= 32768, LH_ERR_ARG, "resident layers: too many");
  memcpy(staged, layers, num_layers * sizeof(GkrLayerDev));
  LH_HIP(hipMemcpyAsync(d_layers, staged, num_layers * sizeof(GkrLayerDev), hipMemcpyHostToDevice, c.stream));
  a.layers = d_layers;
  a.num_layers = (uint32_t)num_layers;
  a.cap = GKR_CAP;
  a.max_trees = max_trees;
  a.ticket = c.ticket;
  a.ticket_base = c.ticket_base;
  c.ticket_base += tickets;
  a.part = c.arena.alloc_n<Fr>((size_t)4 * G + (size_t)(2 * max_trees + 1) * G);
  a.hand = a.part + (size_t)4 * G;
  c.gkr_boxes();
  a.relay = c.gkr_relay;
  a.mbox_round = c.mbox()->c;
  a.mbox_layer = c.gkr_mbox;
  a.msg_host = msg_host;
  a.out_host = out_host;
  a.flag = c.flag;
  const char* tmo = getenv("LH_SC_TAIL_TIMEOUT_MS");
  const double ms = tmo && *tmo ? atof(tmo) : 2000.0;
  a.poll_ticks = (uint64_t)(ms * (double)c.wall_clock_khz);
  const char* smo = getenv("LH_GKR_START_TIMEOUT_MS");
  a.start_ticks = (uint64_t)((smo && *smo ? atof(smo) : 25.0) * (double)c.wall_clock_khz);
  a.start_word = c.ticket + 10;  // (word 0: tickets, word 8: the sharded rounds' device flag, words 32..: the tail's relay)
  a.start_id = layers[0].seq;    // (sequence numbers only grow: no two launches of a ctx share one)
  static const bool trace_on = getenv("LH_GKR_TRACE") != nullptr;
  a.trace = nullptr;
  if (trace_on) {
    const size_t words = ((size_t)GKR_MAX_VARS + GKR_TRACE_ROUNDS) * 8;
    a.trace = (uint64_t*)c.arena.alloc(words * sizeof(uint64_t));
    LH_HIP(hipMemsetAsync(a.trace, 0, words * sizeof(uint64_t), c.stream));
    c.tail_trace = a.trace;
  }
  c.mbox_send(Fr::zero(), 0u);
  const size_t lds = k_gkr_resident_lds_bytes(max_trees, GKR_CAP);
  c.opt_in_lds((const void*)gkr_resident_kernel, (int)k_gkr_resident_lds_bytes(GKR_MAX_TREES, GKR_CAP));
  hipLaunchKernelGGL(gkr_resident_kernel, dim3(G), dim3(GKR_THREADS), lds, c.stream, a);
  LH_HIP(hipGetLastError());
}

}  // namespace lh
