// Host side of the resident grand-product kernel (kernels_gkr.hip), shared by the sum-check loop (the tail of a layer:
// sumcheck.cpp) and the grand-product driver (the layers near the roots: gkr.cpp).
#pragma once
#include <memory>
#include "host.hpp"

namespace lh {

// ------------------------------------------------------------------ resident layers of a grand product (kernels_gkr.hip)
// Host side of the resident kernel: ONE launch serves the layers h = 1 .. H; per layer the host sends the batching
// coefficients and the layer's point, turns every round's q(1), q(2) into the reference's message p(0..3) (the eq
// factoring of host.hpp EqFactoring: p(X) = S eq(y_j, X) q(X), q(0) from the claim), squeezes the challenge and sends
// it back, and at the end of the layer unfolds the coefficients from the left factors' evaluations.
// the host's half of the rounds of one resident layer / tail: q(1), q(2) from the kernel -> the reference's message
// p(0..3) = S eq(y_j, X) q(X) (q(0) from the claim; `add_const`: a constant the kernel leaves out of every q value) ->
// transcript -> challenge back to the kernel.  y[j], inv_1my[j]: the eq point's coordinate of round j and (1 - y_j)^-1;
// S, cq = claim / S, cl = the claim on entry.  Appends the challenges to x.
// Returns false - nothing absorbed, nothing written - when `first` is set and the kernel reports that its workgroups did not
// all start (GKR_START_FAILED: a GPU shared with other resident kernels); the caller takes the launched path.
inline bool resident_rounds(Ctx& c, TailChunk* chunks, uint32_t seq, size_t rounds, const HFr* y, const HFr* inv_1my, HFr S, HFr cq,
                            HFr cl, const HFr& add_const, Transcript& tr, std::vector<HFr>& x, bool first) {
  static const HFr inv2 = HFr::from_u64(2).inv();
  const HFr one = HFr::one(), two = HFr::from_u64(2), three = HFr::from_u64(3), five = HFr::from_u64(5);
  Fr q12[2];
  for (size_t j = 0; j < rounds; j++) {
    if (first && j == 0) {
      if (!c.wait_chunks_or(chunks, 6, seq + 1, GKR_START_FAILED, q12)) return false;
    } else {
      c.wait_chunks(chunks, 6, seq + 1 + (uint32_t)j, q12);
    }
    const HFr q1 = hst(q12[0]) + add_const, q2 = hst(q12[1]) + add_const, yj = y[j];
    const HFr q0 = (cq - yj * q1) * inv_1my[j];
    const HFr q3 = (q2 - q1) * three + q0;  // the quadratic through q(0), q(1), q(2) at 3
    std::vector<HFr> ev(4);
    ev[1] = S * yj * q1;                       // eq(y_j, 1) = y_j
    ev[2] = S * (yj * three - one) * q2;       // eq(y_j, 2) = 3 y_j - 1
    ev[3] = S * (yj * five - two) * q3;        // eq(y_j, 3) = 5 y_j - 2
    ev[0] = cl - ev[1];                        // eval.rs:129
    tr.write_field_elements(ev);
    const HFr r = tr.squeeze_challenge();
    c.mbox_send(dev(r), seq + 1 + (uint32_t)j);
    c.route.v[RouteStats::TAIL_ROUNDS]++;
    // (off the critical path: the kernel binds and evaluates the next round meanwhile)
    const HFr rm1 = r - one, rm2 = r - two;
    cq = q0 * rm1 * rm2 * inv2 - q1 * r * rm2 + q2 * r * rm1 * inv2;
    S = S * ((one - yj) * (one - r) + yj * r);
    cl = S * cq;
    x.push_back(r);
  }
  return true;
}

struct GkrResident {
  Ctx& c;
  size_t H = 0;            // resident layers 1 .. H (0: none)
  bool live = false;       // the kernel is running and expects messages
  std::vector<uint32_t> seq_of;  // layer h -> sequence number of its layer message
  TailChunk* chunks = nullptr;
  Fr* out_host = nullptr;
  ArenaScope* scope = nullptr;
  std::unique_ptr<ProfScope> prof;
  explicit GkrResident(Ctx& c_) : c(c_) {}
  ~GkrResident() { stop(false); }
  // the kernel leaves (or has left): tell it, wait for it, put the ticket counter and the boxes back in order
  void stop(bool finished) {
    if (!live) return;
    live = false;
    if (!finished) {
      c.gkr_abort();
      (void)hipStreamSynchronize(c.stream);
      try {
        c.gkr_resync();
      } catch (...) {
      }
    }
    prof.reset();
  }
  void launch(const std::vector<GkrLayerDev>& layers, const char* prof_name = "gkr_resident") {
    H = layers.size();
    seq_of.assign(H + 1, 0);
    std::vector<GkrLayerDev> ls(layers);
    uint32_t seq = c.flag_seq + 1;
    double entries = 0;
    for (size_t i = 0; i < H; i++) {
      ls[i].seq = seq;
      seq_of[i + 1] = seq;
      seq += ls[i].h + 2;
      entries += (double)ls[i].B * 2.0 * (double)((size_t)1 << ls[i].h);
    }
    c.flag_seq = seq - 1;
    Fr* pin = (Fr*)c.pin((16 + 2 * SC_MAX_TABLES) * sizeof(Fr));
    chunks = (TailChunk*)pin;
    memset((void*)chunks, 0, 6 * sizeof(TailChunk));
    out_host = pin + 16;
    prof.reset(new ProfScope(c, prof_name, 32.0 * entries, 0, entries));
    k_gkr_resident_launch(c, ls.data(), H, chunks, out_host);
    traced = ls;
    live = true;
  }
  // one layer: false when the layer cannot run factored (a zero among 1 - y_j or the coefficients): the kernel is
  // stopped and the caller goes on with launched sum-checks from this layer on
  bool layer(size_t h, const std::vector<HFr>& coeff, const std::vector<HFr>& y, const HFr& claim, Transcript& tr,
             std::vector<HFr>& x, std::vector<HFr>& evals) {
    const size_t B = coeff.size();
    const HFr one = HFr::one();
    // (1 - y_j)^-1 and c_k^-1 with one inversion
    std::vector<HFr> d(h + B), pre(h + B + 1);
    bool ok = true;
    for (size_t j = 0; j < h; j++) d[j] = one - y[j];
    for (size_t k = 0; k < B; k++) d[h + k] = coeff[k];
    pre[0] = one;
    for (size_t i = 0; i < h + B; i++) {
      ok = ok && !d[i].is_zero();
      pre[i + 1] = pre[i] * d[i];
    }
    if (!ok) {
      stop(false);
      return false;
    }
    const uint32_t seq = seq_of[h];
    std::vector<HFr> msg(coeff);
    msg.insert(msg.end(), y.begin(), y.end());
    c.gkr_send_layer((const Fr*)msg.data(), msg.size(), seq);
    // (the kernel loads and folds its tables meanwhile)
    HFr inv = pre[h + B].inv();
    std::vector<HFr> dinv(h + B);
    for (size_t i = h + B; i-- > 0;) {
      dinv[i] = inv * pre[i];
      inv = inv * d[i];
    }
    x.clear();
    if (!resident_rounds(c, chunks, seq, h, y.data(), dinv.data(), one, claim, claim, HFr::zero(), tr, x, h == 1)) {
      stop(false);  // (the launch never got all its workgroups: nothing of it reached the transcript)
      return false;
    }
    c.wait_flag(seq + (uint32_t)h + 1);
    evals.resize(2 * B);
    for (size_t k = 0; k < B; k++) {
      evals[2 * k] = hst(out_host[2 * k]) * dinv[h + k];
      evals[2 * k + 1] = hst(out_host[2 * k + 1]);
    }
    if (h == H) {
      if (c.tail_trace) print_trace();
      stop(true);
    }
    return true;
  }
  // development (LH_GKR_TRACE): device wall-clock stamps of the launch, per layer and per round
  std::vector<GkrLayerDev> traced;
  void print_trace() {
    const size_t words = ((size_t)GKR_MAX_VARS + 160) * 8;
    std::vector<uint64_t> st(words);
    (void)hipStreamSynchronize(c.stream);
    c.d2h(st.data(), c.tail_trace, words * sizeof(uint64_t));
    c.tail_trace = nullptr;
    const double us = 1e3 / (double)c.wall_clock_khz;
    size_t row = 0;
    uint64_t prev_end = 0;
    for (size_t i = 0; i < traced.size(); i++) {
      const GkrLayerDev& L = traced[i];
      const uint64_t* tl = &st[i * 8];
      fprintf(stderr, "[gkr trace] layer h %u B %u g %u s %u: message->ready %.2f us (since previous layer's last bind %.2f)\n", L.h, L.B,
              L.g, 1u << L.s_log, (double)(int64_t)(tl[1] - tl[0]) * us, prev_end ? (double)(int64_t)(tl[0] - prev_end) * us : 0.0);
      uint64_t t_prev = tl[1];
      row = (size_t)(L.seq - traced[0].seq);
      for (uint32_t j = 0; j < L.h && row + 1 + j < 160; j++) {
        const uint64_t* q = &st[((size_t)GKR_MAX_VARS + row + 1 + j) * 8];
        auto rel = [&](int k) { return q[k] && t_prev ? (double)(int64_t)(q[k] - t_prev) * us : -1.0; };
        fprintf(stderr, "    round %2u: eval %.2f | wg0 ticket %.2f | sender sent %.2f challenge %.2f | wg0 challenge %.2f bound %.2f\n", j,
                rel(0), rel(1), rel(2), rel(3), rel(4), rel(5));
        if (q[5]) t_prev = q[5];
        else if (q[4]) t_prev = q[4];
        prev_end = t_prev;
      }
    }
  }
};
// The tail of ONE sum-check of the shape eq * sum_k c_k (l_k + koff_k)(r_k + koff_k) through the resident kernel
// (GKR_F_* tail mode): `L` carries tables, coefficients, offsets, the eq level of the first resident round and the
// pending bind; n0 entries per table once that bind is done.  Returns the challenges and the 2 B raw final values
// (l'_k = c_k (l_k + koff_k), r'_k = r_k + koff_k at the point).
inline bool resident_tail_run(Ctx& c, GkrLayerDev L, size_t n0, const HFr* y, const HFr* inv_1my, const HFr& S, const HFr& cq,
                              const HFr& cl, const HFr& add_const, Transcript& tr, std::vector<HFr>& x, std::vector<HFr>& finals) {
  uint32_t h = 0;
  while (((size_t)1 << h) < n0) h++;
  if (((size_t)1 << h) != n0 || !k_gkr_resident_geometry(h, &L.g, &L.s_log)) return false;
  L.h = h;
  GkrResident run(c);
  run.launch(std::vector<GkrLayerDev>{L}, "gkr_tail");
  c.route.v[RouteStats::TAILS]++;
  const uint32_t seq = run.seq_of[1];
  if (!resident_rounds(c, run.chunks, seq, h, y, inv_1my, S, cq, cl, add_const, tr, x, true)) {
    run.stop(false);
    c.route.v[RouteStats::TAILS]--;
    return false;
  }
  c.wait_flag(seq + h + 1);
  finals.resize(2 * (size_t)L.B);
  for (size_t i = 0; i < finals.size(); i++) finals[i] = hst(run.out_host[i]);
  if (c.tail_trace) run.print_trace();
  run.stop(true);
  return true;
}

}  // namespace lh
