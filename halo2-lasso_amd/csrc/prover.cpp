// Host orchestration of sum-check, GKR and multilinear-KZG over the device kernels.
// Every function cites the reference routine whose transcript schedule it reproduces; all heavy
// loops run in HIP kernels (dev.hpp), the host keeps the Fiat-Shamir state and O(n) scalars.
#include <algorithm>
#include <functional>
#include <chrono>
#include <memory>
#include <thread>
#include "host.hpp"

namespace lh {

// ------------------------------------------------------------------ small host helpers
std::vector<HFr> host_eq_xy(const std::vector<HFr>& y) {
  if (y.empty()) return {};  // MultilinearPolynomial::zero() (multilinear.rs:92-94)
  std::vector<HFr> evals{HFr::one()};
  for (size_t i = y.size(); i-- > 0;) {
    std::vector<HFr> nxt(evals.size() * 2);
    for (size_t k = 0; k < evals.size(); k++) {
      nxt[2 * k + 1] = evals[k] * y[i];
      nxt[2 * k] = evals[k] - nxt[2 * k + 1];
    }
    evals.swap(nxt);
  }
  return evals;
}

HFr host_eq_xy_eval(const HFr* x, const HFr* y, size_t n) {
  HFr acc = HFr::one();
  for (size_t i = 0; i < n; i++) acc *= (x[i] * y[i]).dbl() + HFr::one() - x[i] - y[i];
  return acc;
}

std::vector<HFr> evaluate_polys(Ctx& c, const Fr* const* d_polys, size_t count, size_t num_vars, const HFr* point) {
  std::vector<HFr> out(count);
  if (!count) return out;
  ArenaScope scope(c.arena);
  size_t n = (size_t)1 << num_vars;
  Fr* eq = c.arena.alloc_n<Fr>(n);
  k_eq_xy(c, (const Fr*)point, num_vars, eq);
  k_inner_products(c, d_polys, count, eq, n, (Fr*)out.data());
  return out;
}

// value at x of the polynomial through (i, evals[i]), i = 0..d
// (barycentric_interpolate over points 0..d, reference util/arithmetic.rs:108-136).  The weights
// 1/prod_{i != j}(j - i) depend on d only and are cached: a round costs O(d) multiplications and no
// inversion on the host (an Fr inversion is ~8 us, and there are hundreds of rounds per proof).
static const std::vector<HFr>& lagrange_weights(size_t d) {
  // built once, before any use: contexts on different host threads share the table read-only
  static const std::vector<std::vector<HFr>> cache = [] {
    std::vector<std::vector<HFr>> c(16);
    for (size_t deg = 0; deg < c.size(); deg++) {
      c[deg].resize(deg + 1);
      for (size_t j = 0; j <= deg; j++) {
        HFr de = HFr::one();
        for (size_t i = 0; i <= deg; i++)
          if (i != j) de *= HFr::from_u64(j) - HFr::from_u64(i);
        c[deg][j] = de.inv();
      }
    }
    return c;
  }();
  LH_REQUIRE(d < cache.size(), LH_ERR_ARG, "degree too large");
  return cache[d];
}

HFr interpolate_evals(const std::vector<HFr>& evals, const HFr& x) {
  const size_t d = evals.size() - 1;
  const std::vector<HFr>& w = lagrange_weights(d);
  // prefix[j] = prod_{i<j} (x - i), suffix[j] = prod_{i>j} (x - i); x in {0..d} is covered as well
  std::vector<HFr> diff(d + 1), prefix(d + 2), suffix(d + 2);
  for (size_t i = 0; i <= d; i++) diff[i] = x - HFr::from_u64(i);
  prefix[0] = HFr::one();
  for (size_t i = 0; i <= d; i++) prefix[i + 1] = prefix[i] * diff[i];
  suffix[d + 1] = HFr::one();
  for (size_t i = d + 1; i-- > 0;) suffix[i] = suffix[i + 1] * diff[i];
  HFr total = HFr::zero();
  for (size_t j = 0; j <= d; j++) total += evals[j] * w[j] * prefix[j] * suffix[j + 1];
  return total;
}

HFr horner(const std::vector<HFr>& coeffs, const HFr& x) {
  HFr acc = HFr::zero();
  for (size_t i = coeffs.size(); i-- > 0;) acc = acc * x + coeffs[i];
  return acc;
}

// ------------------------------------------------------------------ communicator helpers (sharded proving)
void comm_sum_fr(Ctx& c, HFr* v, size_t n) {
  const size_t R = (size_t)c.comm.size;
  std::vector<HFr> all(n * R);
  comm_all_gather_host(c, v, all.data(), n * sizeof(HFr));
  for (size_t i = 0; i < n; i++) {
    HFr acc = HFr::zero();
    for (size_t r = 0; r < R; r++) acc += all[r * n + i];
    v[i] = acc;
  }
}

void comm_sum_points(Ctx& c, HG1* pts, size_t n) {
  const size_t R = (size_t)c.comm.size;
  std::vector<HG1> all(n * R);
  comm_all_gather_host(c, pts, all.data(), n * sizeof(HG1));
  // (one inversion for all the sums: ~20 points per exchange at ~10 us of host time per inversion were 0.2 ms on the
  // critical path of every commit and opening)
  std::vector<host::G1Xyzz> acc(n, host::G1Xyzz::identity());
  for (size_t i = 0; i < n; i++)
    for (size_t r = 0; r < R; r++) acc[i] = host::g1_add(acc[i], host::g1_from_affine(all[r * n + i]));
  host::g1_batch_to_affine(acc.data(), n, pts);
}

// `count` local tables back to back (count * n_local entries) -> the full tables on every rank: ONE device all-gather and
// one rearranging pass.  `block`: entries of a rank that are still contiguous in the global order - 2^(shard_bit - rounds
// bound) for the residual tables of a sum-check (1 once the shard bits have reached bit 0), n_local when the shard bits
// are the top bits (tree levels and quotient remainders at the replication point: a plain concatenation).
void comm_gather_tables(Ctx& c, const Fr* local_block, size_t count, size_t n_local, size_t block, Fr* const* out) {
  const size_t R = (size_t)c.comm.size;
  ArenaScope scope(c.arena);
  Fr* gathered = c.arena.alloc_n<Fr>(count * n_local * R);
  comm_all_gather_dev(c, local_block, gathered, count * n_local * sizeof(Fr));
  k_gather_interleave(c, gathered, count, n_local, R, block, out);
}

// out[s * n_local + i] = (rank s).local[i]   (the shard bits are the top bits): exactly an all-gather
void comm_gather_concat(Ctx& c, const Fr* local, size_t n_local, Fr* out) {
  comm_all_gather_dev(c, local, out, n_local * sizeof(Fr));
}

static size_t log2_exact(size_t v) {
  size_t l = 0;
  while (((size_t)1 << l) < v) l++;
  return l;
}

// local shard of eq_xy(y[first..num_vars)): drop the shard coordinates, scale by eq_shard(y_shard)[rank]
void eq_xy_shard(Ctx& c, const Shard& sh, const HFr* y, size_t num_vars, size_t first, Fr* out_local) {
  std::vector<HFr> yl;
  HFr scale = HFr::one();
  for (size_t i = first; i < num_vars; i++) {
    if (i >= sh.j && i < sh.j + sh.rho) {
      bool bit = (sh.rank >> (i - sh.j)) & 1;
      scale *= bit ? y[i] : HFr::one() - y[i];
    } else {
      yl.push_back(y[i]);
    }
  }
  const size_t n_local = (size_t)1 << yl.size();
  k_eq_xy(c, (const Fr*)yl.data(), yl.size(), out_local);
  if (sh.rho) k_scale(c, out_local, dev(scale), n_local, out_local);
}

// evaluations of tables at a point; `sharded`: the tables are this rank's shards of num_vars-variable tables - partial
// inner products against the local shard of eq(point), summed over the ranks
std::vector<HFr> evaluate_polys(Ctx& c, const Fr* const* d_polys, size_t count, size_t num_vars, const HFr* point, bool sharded) {
  if (!sharded) return evaluate_polys(c, d_polys, count, num_vars, point);
  std::vector<HFr> out(count);
  if (!count) return out;
  const Shard sh(c);
  ArenaScope scope(c.arena);
  const size_t n_local = (size_t)1 << (num_vars - sh.rho);
  Fr* eq = c.arena.alloc_n<Fr>(n_local);
  eq_xy_shard(c, sh, point, num_vars, 0, eq);
  k_inner_products(c, d_polys, count, eq, n_local, (Fr*)out.data());
  comm_sum_fr(c, out.data(), count);
  return out;
}

// ------------------------------------------------------------------ shared eq tables of point tails (host.hpp)
// `sharded`: this rank's shard of the table (the shard coordinates dropped, the rank's factor multiplied in) - its entries
// then weigh the local shards of n-variable columns, and the partial sums of the ranks add up
const Fr* eq_half_lookup(Ctx& c, const HFr* y, size_t num_vars, bool sharded) {
  if (num_vars < 2) return nullptr;
  const size_t bytes = (num_vars - 1) * sizeof(HFr);
  for (const Ctx::EqHalfEntry& e : c.eq_half_cache)
    if (e.sharded == sharded && e.key.size() == bytes && memcmp(e.key.data(), y + 1, bytes) == 0) return e.table;
  return nullptr;
}
const Fr* eq_half_get(Ctx& c, const HFr* y, size_t num_vars, bool sharded) {
  if (const Fr* t = eq_half_lookup(c, y, num_vars, sharded)) return t;
  LH_REQUIRE(num_vars >= 2, LH_ERR_ARG, "eq_half: needs two variables");
  Fr* t;
  if (sharded) {
    const Shard sh(c);
    t = c.arena.alloc_n<Fr>((size_t)1 << (num_vars - sh.rho - 1));
    eq_xy_shard(c, sh, y, num_vars, 1, t);  // (shard_bit >= 1: coordinate 0 is never a shard coordinate)
  } else {
    t = c.arena.alloc_n<Fr>((size_t)1 << (num_vars - 1));
    k_eq_xy(c, (const Fr*)(y + 1), num_vars - 1, t);
  }
  Ctx::EqHalfEntry e;
  e.key.assign((const uint8_t*)(y + 1), (const uint8_t*)(y + 1) + (num_vars - 1) * sizeof(HFr));
  e.table = t;
  e.sharded = sharded;
  c.eq_half_cache.push_back(std::move(e));
  return t;
}

// ------------------------------------------------------------------ resident layers of a grand product (kernels_gkr.hip)
// Host side of the resident kernel: ONE launch serves the layers h = 1 .. H; per layer the host sends the batching
// coefficients and the layer's point, turns every round's q(1), q(2) into the reference's message p(0..3) (the eq
// factoring of host.hpp EqFactoring: p(X) = S eq(y_j, X) q(X), q(0) from the claim), squeezes the challenge and sends
// it back, and at the end of the layer unfolds the coefficients from the left factors' evaluations.
namespace {
// the host's half of the rounds of one resident layer / tail: q(1), q(2) from the kernel -> the reference's message
// p(0..3) = S eq(y_j, X) q(X) (q(0) from the claim; `add_const`: a constant the kernel leaves out of every q value) ->
// transcript -> challenge back to the kernel.  y[j], inv_1my[j]: the eq point's coordinate of round j and (1 - y_j)^-1;
// S, cq = claim / S, cl = the claim on entry.  Appends the challenges to x.
// Returns false - nothing absorbed, nothing written - when `first` is set and the kernel reports that its workgroups did not
// all start (GKR_START_FAILED: a GPU shared with other resident kernels); the caller takes the launched path.
static bool resident_rounds(Ctx& c, TailChunk* chunks, uint32_t seq, size_t rounds, const HFr* y, const HFr* inv_1my, HFr S, HFr cq,
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
static bool resident_tail_run(Ctx& c, GkrLayerDev L, size_t n0, const HFr* y, const HFr* inv_1my, const HFr& S, const HFr& cq,
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
}  // namespace

// ------------------------------------------------------------------ the round loop of ClassicSumCheck::prove
// (classic.rs:208-240) shared by the sum-of-products and the general-expression front ends.
// `cur`: current tables (polys first), `used[i]`: the round kernel binds/stores table i itself,
// `round_fn(in, out, r_prev, bind, size, evals_host)`: launches the round kernel and waits for the
// D sums at X = 1..D.
SumCheckResult sum_check_loop(Ctx& c, int prover_kind, size_t num_vars, int degree, std::vector<const Fr*> cur,
                                     const std::vector<char>& used, size_t num_polys, const HFr& sum, Transcript& tr,
                                     bool sharded, const RoundFn& round_fn, const ScRound* tail_rd, EqFactoring* ef) {
  const size_t T = cur.size();
  bool ef_on = ef != nullptr;
  const size_t tail_cap = tail_rd ? k_sc_tail_capacity(c, *tail_rd, degree) : 0;
  const size_t rho = sharded ? log2_exact((size_t)c.comm.size) : 0, j = c.shard_bit;
  size_t len = (size_t)1 << (num_vars - rho);  // current length of every (local) table
  // sharded: the round before which the residual tables are exchanged and the sum-check goes on replicated - as soon as
  // they are small enough for one all-gather (Ctx::shard_exchange_log), at the latest when the shard bits reach bit 0
  size_t x_round = 0;
  if (sharded) {
    x_round = j;
    for (size_t r = 1; r < j; r++)
      if ((T << (num_vars - r)) <= ((size_t)1 << c.opt.shard_exchange_log)) {
        x_round = r;
        break;
      }
  }
  // ping-pong targets of the binds: A holds len/2, B holds len/4
  std::vector<Fr*> bufA(T), bufB(T);
  auto alloc_bufs = [&](size_t l) {
    for (size_t i = 0; i < T; i++) {
      bufA[i] = c.arena.alloc_n<Fr>(std::max<size_t>(l >> 1, 1));
      bufB[i] = c.arena.alloc_n<Fr>(std::max<size_t>(l >> 2, 1));
    }
  };
  alloc_bufs(len);
  int flip = 0;  // next bind target: 0 -> A, 1 -> B
  Fr* evals_host = (Fr*)c.pin((16 + SC_MAX_TABLES) * sizeof(Fr));
  static const HFr inv2 = HFr::from_u64(2).inv();

  SumCheckResult res;
  HFr claim = sum;
  HFr r_prev = HFr::zero();
  bool sh = sharded;
  // one round message: device sums at X = 1..degree -> transcript -> challenge.  The new claim p(r) is not needed before
  // the NEXT message is assembled, so its interpolation is deferred until then: the challenge goes back to the device
  // (resident tail: mailbox; launched rounds: the next launch) without waiting for it - ~1 us less on the critical path of
  // every one of a proof's ~300 rounds.
  std::vector<HFr> pending;  // the last message (evaluations or coefficients), whose value at `pending_r` is the next claim
  HFr pending_r;
  bool pending_coeffs = false;
  auto resolve_claim = [&] {
    if (pending.empty()) return;
    claim = pending_coeffs ? horner(pending, pending_r) : interpolate_evals(pending, pending_r);
    pending.clear();
  };
  auto message = [&](const Fr* sums) {
    resolve_claim();
    std::vector<HFr> ev(degree + 1);
    for (int x = 1; x <= degree; x++) ev[x] = hst(sums[x - 1]);
    ev[0] = claim - ev[1];  // eval.rs:129
    HFr r;
    if (prover_kind == LH_SC_COEFFICIENTS) {
      // coeff.rs:136-149: c0 = p(0), c2 = leading coefficient, c1 = claim - (2 c0 + c2)
      std::vector<HFr> co(3);
      co[0] = ev[0];
      co[2] = (ev[2] - ev[1].dbl() + ev[0]) * inv2;
      co[1] = claim - (co[0].dbl() + co[2]);
      tr.write_field_elements(co);
      r = tr.squeeze_challenge();
      pending = std::move(co), pending_coeffs = true;
    } else {
      tr.write_field_elements(ev);
      r = tr.squeeze_challenge();
      pending = std::move(ev), pending_coeffs = false;
    }
    pending_r = r;
    res.challenges.push_back(r);
    return r;
  };
  bool factored_round = false;
  Fr *d_part = nullptr, *d_all = nullptr;  // sharded rounds: this rank's D sums, every rank's
  bool tail_ok = true;  // cleared when a resident tail ended early: the remaining rounds are launched one by one
  for (size_t round = 0; round < num_vars; round++) {
    bool bind = round > 0;
    if (sh && round == x_round) {
      // bind once more, exchange, go on replicated (the bound tables go into one block: a single all-gather moves them).
      // A factored eq table does not travel: bound through round - 1 it is S_round * eq(y[round..]) on every rank.
      len >>= 1;
      std::vector<size_t> live;
      for (size_t i = 0; i < T; i++) {
        bool factored = false;
        if (ef_on)
          for (const EqFactoring::One& one : ef->eqs) factored = factored || one.table == i;
        if (!factored) live.push_back(i);
      }
      const size_t L = live.size();
      Fr* block = c.arena.alloc_n<Fr>(L * len);
      std::vector<const Fr*> src(L);
      std::vector<Fr*> dst(L), rep(L);
      for (size_t k = 0; k < L; k++) src[k] = cur[live[k]], dst[k] = block + k * len;
      k_fix_var_multi(c, src.data(), dst.data(), L, len << 1, dev(r_prev));
      const size_t full = len << rho;
      for (size_t k = 0; k < L; k++) rep[k] = c.arena.alloc_n<Fr>(full);
      comm_gather_tables(c, block, L, len, (size_t)1 << (j - round), rep.data());
      c.route.v[RouteStats::SHARD_EXCHANGES]++;
      for (size_t k = 0; k < L; k++) cur[live[k]] = rep[k];
      if (ef_on && tail_ok && ef->resident_tail && !ef->per_term && full >= 4 && full <= ((size_t)GKR_CAP * GKR_CAP)) {
        // replicated from here on, and small enough for the resident kernel: its eq level of THIS round (the eq table over
        // the variables after it - no shard coordinate is left among them) is built on every rank, and the rest of the
        // sum-check runs inside the kernel, factored, instead of as launched rounds over materialised eq tables
        EqFactoring::One& one = ef->eqs[0];
        Fr* lvl = c.arena.alloc_n<Fr>(full >> 1);
        k_eq_xy(c, (const Fr*)(one.y + round + 1), num_vars - round - 1, lvl);  // (full = 2^(num_vars - round) >= 4)
        one.level[round] = lvl;
        resolve_claim();
        if (ef->resident_tail(cur, false, r_prev, full, round, claim, tr, res)) return res;
      }
      if (ef_on) {
        for (EqFactoring::One& one : ef->eqs) {
          Fr* tab = c.arena.alloc_n<Fr>(full);
          k_eq_xy(c, (const Fr*)(one.y + round), num_vars - round, tab);
          k_scale(c, tab, dev(one.S), full, tab);
          cur[one.table] = tab;
        }
        ef_on = false;
      }
      len = full;
      alloc_bufs(len);
      flip = 0;
      sh = false;
      bind = false;
    }
    if (ef_on && !sh && tail_ok && ef->resident_tail) {
      // the factored rounds go on INSIDE the resident kernel once the tables fit it (kernels_gkr.hip tail mode): no eq
      // table is materialised, no round is launched any more
      const size_t n0 = bind ? len >> 1 : len;
      if (n0 >= 2 && n0 <= ((size_t)GKR_CAP * GKR_CAP)) {
        resolve_claim();
        if (ef->resident_tail(cur, bind, r_prev, n0, round, claim, tr, res)) return res;
      }
    }
    const bool tail_now = !sh && tail_ok && tail_cap && (bind ? len >> 1 : len) <= tail_cap;
    // (a sharded sum-check whose tail can run in the resident kernel stays factored until its exchange: the few small
    // rounds before it run the factored kernels below their best size rather than lose the factoring - and with it the
    // resident tail - to materialised eq tables)
    const bool keep_factored = sh && tail_ok && ef_on && ef->resident_tail && !ef->per_term;
    if (ef_on && !keep_factored && (tail_now || !ef->streams(bind, bind ? len >> 2 : len >> 1))) {
      // the rounds leave the streaming kernel: materialise every factored eq table in the form the standard path
      // expects (the tables of the previous round, pending their bind with r_prev): S_{round-1} * E_{round-2}
      LH_REQUIRE(round >= 2 && bind, LH_ERR_ARG, "sum-check: eq factoring ended before it began");
      for (EqFactoring::One& one : ef->eqs) {
        Fr* tab = c.arena.alloc_n<Fr>(len);
        k_scale(c, one.level[round - 2], dev(one.S_prev), len, tab);
        cur[one.table] = tab;
      }
      ef_on = false;
    }
    if (tail_now) {
      // the rest of the sum-check runs resident on one CU (dev.hpp: k_sc_tail_*): same messages, same order
      const size_t n0 = bind ? len >> 1 : len, rounds = num_vars - round;
      LH_REQUIRE(((size_t)1 << rounds) == n0 && num_polys <= T, LH_ERR_ARG, "sum-check: internal size mismatch");
      ScRound rd = *tail_rd;
      for (size_t i = 0; i < T; i++) rd.in[i] = cur[i], rd.out[i] = nullptr;
      rd.r = dev(r_prev);
      const uint32_t seq0 = c.flag_seq + 1;
      c.flag_seq += (uint32_t)rounds + 1;
      static const bool tail_debug = getenv("LH_SC_DEBUG") != nullptr;
      const auto t_tail = std::chrono::steady_clock::now();
      ProfScope ps(c, "sc_tail", 0, 0, (double)n0);
      struct Guard {  // never leave the kernel polling: tell it to go, then wait until it is gone
        Ctx& c;
        bool done = false;
        ~Guard() {
          if (done) return;
          c.mbox_abort();
          (void)hipStreamSynchronize(c.stream);
          // the workgroups left having drawn fewer tickets than the launch reserved: later launches must count from
          // where the device counter really is
          try {
            k_sc_tail_resync(c);
          } catch (...) {
          }
        }
      } guard{c};
      // the round messages arrive as 3 * degree self-validating chunks in the first 512 bytes of the pinned block
      TailChunk* chunks = (TailChunk*)evals_host;
      memset((void*)chunks, 0, 3 * SC_TAIL_MAX_DEGREE * sizeof(TailChunk));
      k_sc_tail_launch(c, rd, degree, n0, bind, num_polys, seq0, chunks, evals_host + 16);
      c.route.v[RouteStats::TAILS]++;
      double host_us = 0;
      size_t absorbed = 0;  // tail rounds whose message is in the transcript and whose challenge is known
      bool gave_up = false;
      // The kernel waits a bounded time for each challenge (LH_SC_TAIL_TIMEOUT_MS, default 2 s): a host thread stalled
      // past that (debugger, SIGSTOP, a slow transcript callback) finds the kernel gone.  The entry tables are untouched
      // and the challenges squeezed so far are known, so the sum-check is resumed on the per-round path.
      Fr sums[SC_TAIL_MAX_DEGREE];
      auto wait = [&](uint32_t seq, bool msg) {
        try {
          if (msg) c.wait_chunks(chunks, 3 * (size_t)degree, seq, sums);
          else c.wait_flag(seq);
          return true;
        } catch (const Error& e) {
          if (e.code != LH_ERR_DEVICE || hipStreamQuery(c.stream) != hipSuccess) throw;
          return false;
        }
      };
      for (size_t i = 0; i < rounds && !gave_up; i++) {
        if (!wait(seq0 + (uint32_t)i, true)) {
          gave_up = true;
          break;
        }
        const auto t_h = std::chrono::steady_clock::now();
        const HFr r = message(sums);
        c.mbox_send(dev(r), seq0 + (uint32_t)i);
        absorbed = i + 1;
        c.route.v[RouteStats::TAIL_ROUNDS]++;
        if (tail_debug) host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_h).count();
      }
      if (!gave_up && !wait(seq0 + (uint32_t)rounds, false)) gave_up = true;
      guard.done = true;
      if (!gave_up && c.tail_trace) {
        // development: per round, in us since the round began on workgroup 0: evaluated, ticket drawn, (last workgroup,
        // absolute) partials visible, message sent, challenge seen
        std::vector<uint64_t> st(rounds * 8);
        c.d2h(st.data(), c.tail_trace, st.size() * sizeof(uint64_t));
        c.tail_trace = nullptr;
        const double tick_us = 1e3 / (double)c.wall_clock_khz;
        fprintf(stderr, "[sc_tail trace] T %zu degree %d n0 %zu\n", T, degree, n0);
        for (size_t i = 0; i < rounds; i++) {
          const uint64_t* s8 = &st[i * 8];
          auto rel = [&](int k) { return s8[k] ? (double)(int64_t)(s8[k] - s8[0]) * tick_us : -1.0; };
          fprintf(stderr, "  round %2zu: eval %.2f ticket %.2f last-sees %.2f sent %.2f challenge %.2f | next round starts %.2f\n", i,
                  rel(1), rel(2), rel(3), rel(4), rel(5),
                  i + 1 < rounds && st[(i + 1) * 8] ? (double)(int64_t)(st[(i + 1) * 8] - s8[0]) * tick_us : -1.0);
        }
      }
      if (!gave_up) {
        if (tail_debug)
          fprintf(stderr, "[sc_tail] T %zu terms %u degree %d n0 %zu rounds %zu: %.1f us (host side %.1f us)\n", T,
                  tail_rd->num_terms, degree, n0, rounds,
                  std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_tail).count(), host_us);
        res.evals.resize(num_polys);
        memcpy(res.evals.data(), evals_host + 16, num_polys * sizeof(Fr));
        return res;
      }
      // resume: replay the binds of the `absorbed` rounds with their known challenges (no evaluation, no transcript
      // traffic), then go on with launched rounds from round + absorbed
      if (tail_debug) fprintf(stderr, "[sc_tail] ended early after %zu of %zu rounds: resuming with launched rounds\n", absorbed, rounds);
      tail_ok = false;
      k_sc_tail_resync(c);
      const size_t first_ch = res.challenges.size() - absorbed;
      for (size_t i = 0; i < absorbed; i++) {
        if (bind) {
          std::vector<Fr*>& dst = flip ? bufB : bufA;
          k_fix_var_multi(c, cur.data(), dst.data(), T, len, dev(r_prev));
          for (size_t t = 0; t < T; t++) cur[t] = dst[t];
          len >>= 1;
          flip ^= 1;
        }
        r_prev = res.challenges[first_ch + i];
        bind = true;
      }
      round += absorbed;
      if (round == num_vars) break;  // every message was absorbed: only the final bind is left
    }
    const size_t size = bind ? len >> 2 : len >> 1;
    std::vector<Fr*>& dst = flip ? bufB : bufA;
    factored_round = false;
    if (bind) {
      // tables no term touches are still bound (ProverState::next_round binds every poly)
      for (size_t i = 0; i < T; i++)
        if (!used[i]) k_fix_var(c, cur[i], len, dev(r_prev), dst[i]);
    }
    if (sh) {
      // [round kernel -> all-gather of the partial sums -> sum and publish], all on the ctx's stream.  With factored eq
      // tables the partial sums are those of q (this rank's eq-level entries carry its factor of the shard coordinates).
      const size_t R = (size_t)c.comm.size;
      const bool lanes = c.opt.comm_round != 0;  // ONE collective per round: all-reduce of u64 lanes (comm.cpp comm_sum_lanes)
      if (!d_part) {
        d_part = c.arena.alloc_n<Fr>(16);
        d_all = c.arena.alloc_n<Fr>(std::max<size_t>(16 * R, 64));  // (the lanes variant: 2 x 16 sums of 8 u64 = 2 KB)
      }
      // (the device buffers hold 16 sums: degree <= 6, at most SC_OPEN_MAX_TERMS = 6 factored terms)
      const size_t nvals = !ef_on ? (size_t)degree : ef->per_term ? 2 * ef->eqs.size() : (size_t)degree - 1;
      LH_REQUIRE(nvals >= 1 && nvals <= 16, LH_ERR_ARG, "sharded sum-check: too many partial sums per round");
      c.sc_redirect = d_part;
      uint64_t* d_lanes = (uint64_t*)d_all;  // [0, 128): this rank's lanes, [128, 256): the sums (comm_round 2)
      if (lanes) c.sc_wide = d_lanes, c.sc_tag = comm_next_tag(c);
      try {
        if (ef_on) {
          ef->add_const = HFr::zero();
          ef->round(cur.data(), dst.data(), dev(r_prev), bind, size, round, (int)nvals, evals_host);
          factored_round = true;
        } else {
          round_fn(cur.data(), dst.data(), dev(r_prev), bind, size, evals_host);
        }
      } catch (...) {
        c.sc_redirect = nullptr, c.sc_wide = nullptr;
        throw;
      }
      c.sc_redirect = nullptr, c.sc_wide = nullptr;
      c.route.v[RouteStats::SHARDED_ROUNDS]++;
      c.route.v[factored_round ? RouteStats::EF_ROUNDS : RouteStats::STD_ROUNDS]++;
      if (lanes) {
        comm_sum_lanes(c, d_lanes, d_lanes + 128, nvals, evals_host);
      } else {
        const uint32_t seq = c.next_seq();
        comm_sum_publish(c, d_part, d_all, nvals, evals_host, seq);
        c.wait_flag(seq);
      }
      if (factored_round && !ef->per_term)  // (constants the factored kernel leaves to the host: the eq level sums to one)
        for (size_t x = 0; x < nvals; x++) evals_host[x] = dev(hst(evals_host[x]) + ef->add_const);
    } else if (ef_on) {
      // global-eq shape, round 0: one point more (q at 1..D determines q(0) too), so that the claim can be CHECKED instead
      // of trusted: with a claim that is not the true sum the reference still sends the true p(1..D), and so must we
      const bool check_claim = !ef->per_term && round == 0 && !ef->trusted_claim;
      ef->add_const = HFr::zero();
      ef->round(cur.data(), dst.data(), dev(r_prev), bind, size, round, check_claim ? degree : degree - 1, evals_host);
      factored_round = true;
      c.route.v[RouteStats::EF_ROUNDS]++;
      if (!ef->add_const.is_zero())
        for (int x = 0; x < degree - 1; x++) evals_host[x] = dev(hst(evals_host[x]) + ef->add_const);
      if (check_claim) {
        EqFactoring::One& e = ef->eqs[0];
        std::vector<HFr> shifted(degree);  // t -> q(t + 1), t = 0..D-1
        for (int x = 0; x < degree; x++) shifted[x] = hst(evals_host[x]);
        const HFr q0 = interpolate_evals(shifted, HFr::zero() - HFr::one());
        const HFr y0 = e.y[0];
        if ((HFr::one() - y0) * q0 + y0 * shifted[0] != ef->c) {
          // not the true sum: every round takes the standard path (eq tables built in full)
          for (EqFactoring::One& one : ef->eqs) {
            Fr* tab = c.arena.alloc_n<Fr>(len);
            k_eq_xy(c, (const Fr*)one.y, num_vars, tab);
            cur[one.table] = tab;
          }
          ef_on = factored_round = false;
          c.route.v[RouteStats::EF_ROUNDS]--, c.route.v[RouteStats::STD_ROUNDS]++;
          round_fn(cur.data(), dst.data(), dev(r_prev), bind, size, evals_host);
        } else {
          e.q.assign(degree, HFr::zero());
          e.q[0] = q0;
          for (int x = 1; x < degree; x++) e.q[x] = shifted[x - 1];
        }
      }
    } else {
      round_fn(cur.data(), dst.data(), dev(r_prev), bind, size, evals_host);
      c.route.v[RouteStats::STD_ROUNDS]++;
    }
    if (bind) {
      for (size_t i = 0; i < T; i++) cur[i] = dst[i];
      len >>= 1;
      flip ^= 1;
    }
    if (factored_round) {
      // rebuild the reference's round message p(1..D) from the factored sums (host.hpp EqFactoring)
      Fr std_sums[16];
      const HFr one = HFr::one();
      auto eq_at = [&](const HFr& yj, const HFr& x) { return (one - yj) * (one - x) + yj * x; };  // eq(y_j, x)
      if (!ef->per_term) {
        EqFactoring::One& e = ef->eqs[0];
        const HFr yj = e.y[round];
        if (round > 0 || ef->trusted_claim) {
          e.q.assign(degree, HFr::zero());  // q has degree D - 1: D values q(0..D-1)
          for (int x = 1; x < degree; x++) e.q[x] = hst(evals_host[x - 1]);
          e.q[0] = (ef->c - yj * e.q[1]) * ef->inv_1my[round];
        }
        for (int x = 1; x <= degree; x++) {
          const HFr fx = HFr::from_u64((uint64_t)x);
          const HFr qx = x < degree ? e.q[x] : interpolate_evals(e.q, fx);
          std_sums[x - 1] = dev(e.S * eq_at(yj, fx) * qx);
        }
        const HFr r = message(std_sums);
        ef->c = interpolate_evals(e.q, r);
        e.S_prev = e.S;
        e.S = e.S * eq_at(yj, r);
        r_prev = r;
      } else {
        HFr p1 = HFr::zero(), p2 = HFr::zero();
        const HFr two = HFr::from_u64(2);
        for (size_t m = 0; m < ef->eqs.size(); m++) {
          EqFactoring::One& e = ef->eqs[m];
          const HFr yj = e.y[round];
          e.q = {hst(evals_host[2 * m]), hst(evals_host[2 * m + 1])};
          p1 += e.S * yj * e.q[1];                                         // eq(y_j, 1) = y_j
          p2 += e.S * eq_at(yj, two) * (e.q[1].dbl() - e.q[0]);          // q(2) of a line
        }
        std_sums[0] = dev(p1), std_sums[1] = dev(p2);
        const HFr r = message(std_sums);
        for (EqFactoring::One& e : ef->eqs) {
          e.S_prev = e.S;
          e.S = e.S * eq_at(e.y[round], r);
        }
        r_prev = r;
      }
      continue;
    }

    r_prev = message(evals_host);
  }
  LH_REQUIRE(!sh && len == 2, LH_ERR_ARG, "sum-check: internal size mismatch");
  // into_evals: last bind (2 -> 1 entries) of every poly
  if (num_polys) {
    LH_REQUIRE(num_polys <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "sum-check: too many polys");
    Fr* out = (Fr*)c.pin((16 + SC_MAX_TABLES) * sizeof(Fr)) + 16;
    k_bind_first(c, cur.data(), num_polys, dev(r_prev), out);
    res.evals.resize(num_polys);
    memcpy(res.evals.data(), out, num_polys * sizeof(Fr));
  }
  return res;
}

// ------------------------------------------------------------------ ClassicSumCheck::prove
// reference piop/sum_check/classic.rs:208-240.  Round i: [fused bind with r_{i-1}] + evaluation on the
// GPU (k_sc_round), message to the transcript, squeeze r_i.  After the last squeeze one more bind gives
// table[0] of every poly (classic.rs:143-149).
//
// `sharded`: the tables are this rank's shards (dev.hpp Shard; this is ProverState::next_round, classic.rs:90-141, over a
// shard).  The first rounds run the very same kernels on the local shard - eq factoring included - and the partial sums
// of all ranks are added; once the residual tables are small (at the latest before round shard_bit, when the shard bits
// would become the pair bit) they are bound once more, exchanged, and the remaining rounds run replicated on every rank.
// The transcript sees exactly the single-GPU messages.
static SumCheckResult sum_check_prove_impl(Ctx& c, int prover_kind, size_t num_vars, const lh_sop& expr,
                                           const Fr* const* d_polys, size_t num_polys, const HFr* ys, size_t num_ys,
                                           const HFr& sum, Transcript& tr, bool sharded, bool sum_is_exact = false,
                                           const ScRwPairs* rw = nullptr) {
  LH_REQUIRE(num_vars > 0, LH_ERR_ARG, "sum-check needs num_vars > 0");  // classic.rs:42 assert
  const size_t T = num_polys + num_ys;
  LH_REQUIRE(T <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "sum-check: too many tables for one round kernel");
  LH_REQUIRE(expr.num_terms >= 1 && expr.num_terms <= LH_SC_MAX_TERMS, LH_ERR_ARG, "sum-check: bad term count");
  LH_REQUIRE(expr.global_eq < (int)num_ys, LH_ERR_ARG, "sum-check: global_eq out of range");
  const size_t rho = sharded ? log2_exact((size_t)c.comm.size) : 0, j = c.shard_bit;
  if (sharded) LH_REQUIRE(j >= 1 && j + rho <= num_vars, LH_ERR_ARG, "sharded sum-check: shard bits outside the table");

  // Expression::degree() (expression.rs:171-182): every table is degree 1, products add, sums max
  int degree = 0;
  ScRound rd;
  memset(&rd, 0, sizeof(rd));
  rd.num_tables = (uint32_t)T;
  rd.num_terms = expr.num_terms;
  rd.global_eq = expr.global_eq >= 0 ? (int)num_polys + expr.global_eq : -1;
  const HFr one = HFr::one();
  std::vector<char> used(T, 0);
  if (rd.global_eq >= 0) used[rd.global_eq] = 1;
  for (uint32_t m = 0; m < expr.num_terms; m++) {
    int nf = expr.num_factors[m];
    // (no factor at all: a constant times the global eq - the prover's own layer expressions use it)
    LH_REQUIRE((nf >= 1 || (nf == 0 && expr.global_eq >= 0)) && nf <= LH_SC_MAX_FACTORS, LH_ERR_ARG,
               "sum-check: bad factor count");
    degree = std::max(degree, nf + (expr.global_eq >= 0 ? 1 : 0));
    memcpy(&rd.coeff[m], &expr.coeff[m], 32);
    rd.coeff_is_one[m] = (memcmp(&expr.coeff[m], &one, 32) == 0);
    rd.nfac[m] = (uint8_t)nf;
    for (int k = 0; k < nf; k++) {
      LH_REQUIRE(expr.factor[m][k] < T, LH_ERR_ARG, "sum-check: factor id out of range");
      rd.fac[m][k] = expr.factor[m][k];
      used[expr.factor[m][k]] = 1;
    }
  }
  if (prover_kind == LH_SC_COEFFICIENTS)
    LH_REQUIRE(degree == 2, LH_ERR_ARG, "CoefficientsProver supports degree 2 only");  // coeff.rs:143 unimplemented!()
  else
    LH_REQUIRE(degree >= 2, LH_ERR_ARG, "EvaluationsProver needs degree >= 2");  // eval.rs:316 debug_assert

  ArenaScope scope(c.arena);
  const size_t len0 = (size_t)1 << (num_vars - rho);
  std::vector<const Fr*> cur(T);
  for (size_t i = 0; i < num_polys; i++) cur[i] = d_polys[i];
  auto round_fn = [&](const Fr* const* in, Fr* const* out, const Fr& r, bool bind, size_t size, Fr* evals_host) {
    for (size_t i = 0; i < T; i++) {
      rd.in[i] = in[i];
      rd.out[i] = out[i];
    }
    rd.r = r;
    k_sc_round(c, rd, degree, bind, size, evals_host);
  };

  // ---- eq factoring of the streaming rounds (host.hpp EqFactoring)
  const bool ef_enabled = c.opt.sc_eq_factoring != 0;  // 0: every round streams and binds its eq tables (A/B measurements)
  EqFactoring ef;
  bool use_ef = false;
  std::vector<size_t> term_poly;  // per-term shape: the poly of term m
  const size_t nvl = num_vars - rho;  // variables of the local tables
  // (sharded: the first two rounds must be local ones, and the claim is not checked against partial sums)
  if (ef_enabled && nvl >= 3 && (!sharded || j >= 2) && k_sc_round_streams(rd, degree, (size_t)1 << (nvl - 1)) &&
      k_sc_round_streams(rd, degree, (size_t)1 << (nvl - 2))) {
    if (rd.global_eq >= 0 && prover_kind == LH_SC_EVALUATIONS && degree >= 2 && (!sharded || sum_is_exact)) {
      // shape A: eq(ys[global_eq]) times a sum of products that does not use that eq table as a factor
      bool ok = true;
      for (uint32_t m = 0; m < rd.num_terms && ok; m++)
        for (int k = 0; k < rd.nfac[m]; k++) ok = ok && rd.fac[m][k] != rd.global_eq;
      const HFr* y = ys + (size_t)expr.global_eq * num_vars;
      std::vector<HFr> d(num_vars);
      for (size_t i = 0; i < num_vars && ok; i++) {
        d[i] = HFr::one() - y[i];
        ok = !d[i].is_zero();
      }
      if (ok) {
        // (1 - y_j)^-1 for all rounds with one inversion
        std::vector<HFr> pre(num_vars + 1);
        pre[0] = HFr::one();
        for (size_t i = 0; i < num_vars; i++) pre[i + 1] = pre[i] * d[i];
        HFr inv = pre[num_vars].inv();
        ef.inv_1my.resize(num_vars);
        for (size_t i = num_vars; i-- > 0;) {
          ef.inv_1my[i] = inv * pre[i];
          inv = inv * d[i];
        }
        ef.per_term = false;
        ef.trusted_claim = sum_is_exact;
        ef.c = sum;
        EqFactoring::One one;
        one.table = (size_t)rd.global_eq, one.y = y, one.S = one.S_prev = HFr::one();
        ef.eqs.push_back(one);
        use_ef = true;
      }
    } else if (rd.global_eq < 0 && prover_kind == LH_SC_COEFFICIENTS && rd.num_terms <= (uint32_t)SC_OPEN_MAX_TERMS &&
               T == 2 * (size_t)rd.num_terms) {
      // shape B (batch opening): sum_m eq_m * poly_m, coefficient one, every table in exactly one term
      bool ok = true;
      std::vector<char> seen(T, 0);
      for (uint32_t m = 0; m < rd.num_terms && ok; m++) {
        ok = rd.nfac[m] == 2 && rd.coeff_is_one[m];
        if (!ok) break;
        size_t a = rd.fac[m][0], b = rd.fac[m][1];
        if (a < num_polys) std::swap(a, b);  // a: the eq table, b: the poly
        ok = a >= num_polys && b < num_polys && !seen[a] && !seen[b];
        if (!ok) break;
        seen[a] = seen[b] = 1;
        EqFactoring::One one;
        one.table = a, one.y = ys + (a - num_polys) * num_vars, one.S = one.S_prev = HFr::one();
        ef.eqs.push_back(one);
        term_poly.push_back(b);
      }
      ef.per_term = true;
      use_ef = ok;
    }
  }
  // product-pair shape (dev.hpp ScRound::pp): every term c_m l_m r_m over 2 num_terms distinct tables, non-zero
  // coefficients.  Its factored rounds run sc_round_pp_kernel, and the first of them that binds folds the coefficients
  // into the left factors: from then on every kernel of this sum-check (streaming, LDS-staged, resident tail) sees
  // coefficients of one, and the final evaluations of the left factors are divided by c_m at the end.
  bool pp_shape = use_ef && !ef.per_term && !rw && rd.num_terms >= 2 && c.opt.sc_pp_fold != 0;
  {
    std::vector<char> seen(T, 0);
    for (uint32_t m = 0; m < rd.num_terms && pp_shape; m++) {
      HFr co;
      memcpy(&co, &rd.coeff[m], 32);
      pp_shape = rd.nfac[m] == 2 && !co.is_zero();
      for (int k = 0; k < 2 && pp_shape; k++) {
        pp_shape = rd.fac[m][k] < num_polys && !seen[rd.fac[m][k]];
        seen[rd.fac[m][k]] = 1;
      }
    }
  }
  std::vector<HFr> pp_folded;  // the coefficients that went into the left factors (empty: not folded)
  bool rw_folded = false;      // tree-pair rounds (ScRwPairs): the tables hold l' = cs (l + k), r' = r + k since the first bind
  if (use_ef) {
    const Shard shg(c);
    const size_t half = (size_t)1 << (nvl - 1);
    static const bool eq_levels_ahead = !(getenv("LH_SC_EQ_LEVELS_AHEAD") && atoi(getenv("LH_SC_EQ_LEVELS_AHEAD")) == 0);
    for (EqFactoring::One& one : ef.eqs) {
      // consecutive blocks of halving size in one buffer; E_0 (the eq table over variables 1..n-1; sharded: this rank's
      // shard of it) comes from the proof's shared tables when an evaluation at the same point built it already
      const Fr* shared = eq_half_lookup(c, one.y, num_vars, sharded);
      Fr* buf = c.arena.alloc_n<Fr>(shared ? half : 2 * half);
      one.level.resize(num_vars);
      size_t off = 0;
      for (size_t jl = 0; jl < nvl; jl++) {
        if (jl == 0 && shared) {
          one.level[0] = const_cast<Fr*>(shared);
          continue;
        }
        one.level[jl] = buf + off;
        off += half >> jl;
      }
      if (!shared) {
        if (sharded) eq_xy_shard(c, shg, one.y, num_vars, 1, buf);
        else k_eq_xy(c, (const Fr*)(one.y + 1), num_vars - 1, buf);
      }
      // E_{j+1} from E_j: the two entries that differ in variable j + 1 add up.  No level depends on a challenge: all of
      // them now, nine per launch (one launch per round in front of the round's kernel was 76 launches per 2^24 proof)
      std::vector<Fr*> lower;
      for (size_t jl = 1; jl < nvl; jl++) lower.push_back((Fr*)one.level[jl]);
      if (eq_levels_ahead) k_eq_levels(c, one.level[0], half, lower.data(), lower.size());
    }
    ef.streams = [&](bool, size_t size) { return k_sc_round_streams(rd, degree, size); };
    ef.round = [&](const Fr* const* in, Fr* const* out, const Fr& r, bool bind, size_t size, size_t round, int points,
                   Fr* out_host) {
      static const bool eq_levels_ahead = !(getenv("LH_SC_EQ_LEVELS_AHEAD") && atoi(getenv("LH_SC_EQ_LEVELS_AHEAD")) == 0);
      if (!eq_levels_ahead && round > 0)  // (A/B: one level per round, in front of the round's kernel, as before)
        for (EqFactoring::One& one : ef.eqs) {
          Fr* lvl = (Fr*)one.level[round];
          k_eq_levels(c, one.level[round - 1], 2 * size, &lvl, 1);
        }
      if (!ef.per_term && rw && rw_folded) {
        // the tables hold l' = cs (l + k), r' = r + k since the fold round: sum_p l'_p r'_p, the product-pair shape
        LH_REQUIRE(points == 2, LH_ERR_ARG, "sum-check: tree-pair rounds asked for an extra point after the fold");
        ScRound g;
        memset(&g, 0, sizeof(g));
        g.num_tables = 2 * rw->num_pairs, g.num_terms = rw->num_pairs;
        for (size_t i = 0; i < 2 * (size_t)rw->num_pairs; i++) g.in[i] = in[i], g.out[i] = out[i];
        for (uint32_t m = 0; m < rw->num_pairs; m++) {
          g.coeff[m] = dev(HFr::one()), g.coeff_is_one[m] = 1;
          g.nfac[m] = 2, g.fac[m][0] = (uint8_t)(2 * m), g.fac[m][1] = (uint8_t)(2 * m + 1);
        }
        g.r = r;
        g.global_eq = -1;
        g.eq_level = ef.eqs[0].level[round];
        g.pp = 1;
        k_sc_round(c, g, points, bind, size, out_host);
        ef.add_const = rw->const_total;
      } else if (!ef.per_term && rw && points == 2) {
        ScRwRound g;
        memset(&g, 0, sizeof(g));
        g.num_pairs = rw->num_pairs;
        for (uint32_t i = 0; i < rw->num_pairs; i++) {
          g.l[i] = in[2 * i], g.r[i] = in[2 * i + 1];
          g.lo[i] = out[2 * i], g.ro[i] = out[2 * i + 1];
          g.cs[i] = dev(rw->cs[i]), g.k[i] = dev(rw->k[i]);
        }
        g.eq_level = ef.eqs[0].level[round];
        g.rchal = r;
        const bool fold = bind && c.opt.sc_pp_fold != 0;  // the first binding round folds cs and k into the tables
        k_sc_round_rw(c, g, bind, size, out_host, fold);
        if (fold) {
          rw_folded = true;
          c.route.v[RouteStats::PP_FOLDS]++;
          // the expression over the tables as they are now, for whoever evaluates it in its general form from here on (the
          // launched small rounds, the generic resident tail): sum_p l'_p r'_p + const_total
          const uint32_t P = rw->num_pairs;
          rd.num_terms = P + 1;
          for (uint32_t m = 0; m < P; m++) {
            rd.coeff[m] = dev(HFr::one()), rd.coeff_is_one[m] = 1;
            rd.nfac[m] = 2, rd.fac[m][0] = (uint8_t)(2 * m), rd.fac[m][1] = (uint8_t)(2 * m + 1);
          }
          rd.coeff[P] = dev(rw->const_total), rd.coeff_is_one[P] = 0, rd.nfac[P] = 0;
        }
        c.route.v[RouteStats::RW_ROUNDS]++;
        ef.add_const = rw->const_total;  // added to q(1), q(2) by the round loop (the suffix eq sums to one - over all ranks)
      } else if (!ef.per_term) {
        ScRound g = rd;
        for (size_t i = 0; i < T; i++) g.in[i] = in[i], g.out[i] = out[i];
        g.r = r;
        g.global_eq = -1;
        g.eq_level = ef.eqs[0].level[round];
        g.pp = pp_shape && points == 2 ? 1 : 0;
        if (g.pp && bind && pp_folded.empty()) g.pp = 2;  // this round stores l'_m = c_m l_m
        k_sc_round(c, g, points, bind, size, out_host);
        if (g.pp == 2 && c.last_round_folded) {  // (the launch that was chosen for this size did fold)
          pp_folded.resize(rd.num_terms);
          for (uint32_t m = 0; m < rd.num_terms; m++) {
            memcpy(&pp_folded[m], &rd.coeff[m], 32);
            rd.coeff[m] = dev(HFr::one());
            rd.coeff_is_one[m] = 1;
          }
          c.route.v[RouteStats::PP_FOLDS]++;
        }
      } else {
        ScOpenRound g;
        g.num_terms = rd.num_terms;
        g.r = r;
        for (uint32_t m = 0; m < rd.num_terms; m++) {
          g.in[m] = in[term_poly[m]], g.out[m] = out[term_poly[m]];
          g.eq_level[m] = ef.eqs[m].level[round];
        }
        k_sc_round_open(c, g, bind, size, out_host);
      }
    };
  }
  if (use_ef && !ef.per_term && c.opt.gkr_resident && c.opt.sc_tail && (pp_shape || rw) && num_polys == 2 * (size_t)(rw ? rw->num_pairs : rd.num_terms) &&
      (rw ? rw->num_pairs : rd.num_terms) <= (uint32_t)GKR_MAX_TREES) {
    ef.resident_tail = [&](const std::vector<const Fr*>& cur_t, bool bind, const HFr& r_prev, size_t n0, size_t round,
                           const HFr& claim_now, Transcript& trr, SumCheckResult& res) {
      const size_t Bt = rw ? rw->num_pairs : rd.num_terms;
      GkrLayerDev L;
      memset(&L, 0, sizeof(L));
      L.B = (uint32_t)Bt;
      L.flags = GKR_F_SPLIT | GKR_F_NOMSG | GKR_F_EQ | (bind ? GKR_F_BIND : 0) | (rw && !rw_folded ? GKR_F_KOFF : 0);
      std::vector<size_t> li(Bt), ri(Bt);
      std::vector<HFr> co(Bt), ko(Bt, HFr::zero());
      for (size_t m = 0; m < Bt; m++) {
        li[m] = rw ? 2 * m : rd.fac[m][0], ri[m] = rw ? 2 * m + 1 : rd.fac[m][1];
        if (rw && rw_folded) co[m] = HFr::one();  // (l', r' as they are: the end of sum_check_prove_impl unfolds)
        else if (rw) co[m] = rw->cs[m], ko[m] = rw->k[m];
        else memcpy(&co[m], &rd.coeff[m], 32);
        if (co[m].is_zero()) return false;
        L.lv[m] = cur_t[li[m]], L.rv[m] = cur_t[ri[m]];
        L.coef[m] = dev(co[m]), L.koff[m] = dev(ko[m]);
      }
      EqFactoring::One& e = ef.eqs[0];
      L.eq_level = e.level[round];
      L.r_prev = dev(r_prev);
      std::vector<HFr> x, finals;
      if (!resident_tail_run(c, L, n0, e.y + round, ef.inv_1my.data() + round, e.S, ef.c, claim_now, rw ? rw->const_total : HFr::zero(),
                             trr, x, finals))
        return false;
      res.challenges.insert(res.challenges.end(), x.begin(), x.end());
      // unfold: l = l' / c - k, r = r' - k (one inversion for the coefficients)
      std::vector<HFr> pre(Bt + 1);
      pre[0] = HFr::one();
      for (size_t m = 0; m < Bt; m++) pre[m + 1] = pre[m] * co[m];
      HFr inv = pre[Bt].inv();
      res.evals.assign(num_polys, HFr::zero());
      for (size_t m = Bt; m-- > 0;) {
        res.evals[li[m]] = finals[2 * m] * (inv * pre[m]) - ko[m];
        res.evals[ri[m]] = finals[2 * m + 1] - ko[m];
        inv = inv * co[m];
      }
      return true;
    };
  }
  // ProverState::new: eq_xys (classic.rs:56-60); a factored eq table is not built - its slot is filled when the rounds
  // leave the streaming kernel
  for (size_t jy = 0; jy < num_ys; jy++) {
    bool factored = false;
    if (use_ef)
      for (const EqFactoring::One& one : ef.eqs) factored = factored || one.table == num_polys + jy;
    if (factored) continue;
    Fr* eq = c.arena.alloc_n<Fr>(len0);
    if (sharded) eq_xy_shard(c, Shard(c), ys + jy * num_vars, num_vars, 0, eq);
    else k_eq_xy(c, (const Fr*)(ys + jy * num_vars), num_vars, eq);
    cur[num_polys + jy] = eq;
  }
  SumCheckResult res = sum_check_loop(c, prover_kind, num_vars, degree, cur, used, num_polys, sum, tr, sharded, round_fn, &rd,
                                      use_ef ? &ef : nullptr);
  if (rw_folded) {
    // l' = cs (l + k), r' = r + k came out: l = l' / cs - k, r = r' - k (one inversion for the coefficients)
    const size_t K = rw->num_pairs;
    std::vector<HFr> pre(K + 1);
    pre[0] = HFr::one();
    for (size_t m = 0; m < K; m++) pre[m + 1] = pre[m] * rw->cs[m];
    HFr inv = pre[K].inv();
    for (size_t m = K; m-- > 0;) {
      res.evals[2 * m] = res.evals[2 * m] * (inv * pre[m]) - rw->k[m];
      res.evals[2 * m + 1] = res.evals[2 * m + 1] - rw->k[m];
      inv = inv * rw->cs[m];
    }
  }
  if (!pp_folded.empty()) {
    // the left factors came out times their coefficients: one inversion for all of them
    const size_t K = pp_folded.size();
    std::vector<HFr> pre(K + 1);
    pre[0] = HFr::one();
    for (size_t m = 0; m < K; m++) pre[m + 1] = pre[m] * pp_folded[m];
    HFr inv = pre[K].inv();
    for (size_t m = K; m-- > 0;) {
      res.evals[rd.fac[m][0]] = res.evals[rd.fac[m][0]] * (inv * pre[m]);
      inv = inv * pp_folded[m];
    }
  }
  return res;
}

SumCheckResult sum_check_prove(Ctx& c, int prover_kind, size_t num_vars, const lh_sop& expr,
                               const Fr* const* d_polys, size_t num_polys, const HFr* ys, size_t num_ys,
                               const HFr& sum, Transcript& tr, bool sum_is_exact, const ScRwPairs* rw, bool sharded) {
  if (rw)
    LH_REQUIRE(rw->num_pairs >= 1 && rw->num_pairs <= (uint32_t)SC_RW_MAX_PAIRS && num_polys == 2 * (size_t)rw->num_pairs &&
                   expr.global_eq >= 0 && sum_is_exact,
               LH_ERR_ARG, "sum-check: tree-pair rounds over the wrong shape");
  if (sharded) LH_REQUIRE(c.shard_active && c.has_comm, LH_ERR_ARG, "sharded sum-check outside a sharded proof");
  return sum_check_prove_impl(c, prover_kind, num_vars, expr, d_polys, num_polys, ys, num_ys, sum, tr, sharded, sum_is_exact,
                              rw);
}

// ------------------------------------------------------------------ prove_fractional_sum_check
// reference piop/gkr/fractional_sum_check.rs:89-190
static void download(Ctx& c, void* dst, const void* src, size_t bytes) { c.d2h(dst, src, bytes); }

FracSumCheckResult prove_fractional_sum_check(Ctx& c, size_t B, size_t num_vars, const HFr* const* claimed_p_0s,
                                              const HFr* const* claimed_q_0s, const Fr* const* d_ps,
                                              const Fr* const* d_qs, Transcript& tr) {
  LH_REQUIRE(B != 0, LH_ERR_ARG, "fractional sum-check: num_batching == 0");  // :103 assert
  LH_REQUIRE(num_vars >= 1, LH_ERR_ARG, "fractional sum-check: num_vars == 0");
  LH_REQUIRE(3 * B <= LH_SC_MAX_TERMS && 4 * B + 1 <= (size_t)SC_MAX_TABLES, LH_ERR_ARG,
             "fractional sum-check: too many fractions for one round kernel");
  ArenaScope scope(c.arena);
  // levels[h][b] = (p, q) arrays of 2^(num_vars-h) entries; Layer::bottom/up (:42-85) are views of them
  std::vector<std::vector<const Fr*>> lp(num_vars), lq(num_vars);
  for (size_t b = 0; b < B; b++) {
    lp[0].push_back(d_ps[b]);
    lq[0].push_back(d_qs[b]);
  }
  for (size_t h = 1; h < num_vars; h++) {
    size_t half = (size_t)1 << (num_vars - h);
    for (size_t b = 0; b < B; b++) {
      Fr* vp = c.arena.alloc_n<Fr>(half);
      Fr* vq = c.arena.alloc_n<Fr>(half);
      k_frac_up(c, lp[h - 1][b], lq[h - 1][b], half, vp, vq);
      lp[h].push_back(vp);
      lq[h].push_back(vq);
    }
  }
  // roots from the top (0-variable) layer (:116-125)
  std::vector<HFr> top(4 * B);
  {
    std::vector<const Fr*> heads;
    for (size_t b = 0; b < B; b++) {
      heads.push_back(lp[num_vars - 1][b]);
      heads.push_back(lq[num_vars - 1][b]);
    }
    Fr* out = (Fr*)c.pin(2 * SC_MAX_TABLES * sizeof(Fr));
    k_gather_heads(c, heads.data(), heads.size(), 2, out);
    memcpy(top.data(), out, 4 * B * sizeof(Fr));
  }
  std::vector<HFr> claimed_p(B), claimed_q(B);
  for (size_t b = 0; b < B; b++) {
    const HFr &p_l = top[4 * b], &p_r = top[4 * b + 1], &q_l = top[4 * b + 2], &q_r = top[4 * b + 3];
    claimed_p[b] = p_l * q_r + p_r * q_l;
    claimed_q[b] = q_l * q_r;
  }
  for (size_t b = 0; b < B; b++) {  // :127-142: Some -> common, None -> write
    if (claimed_p_0s && claimed_p_0s[b]) tr.common_field_element(claimed_p[b]);
    else tr.write_field_element(claimed_p[b]);
  }
  for (size_t b = 0; b < B; b++) {
    if (claimed_q_0s && claimed_q_0s[b]) tr.common_field_element(claimed_q[b]);
    else tr.write_field_element(claimed_q[b]);
  }

  std::vector<HFr> y;
  for (size_t h = num_vars; h-- > 0;) {  // layers.iter().rev()
    const size_t nv = num_vars - 1 - h;  // variables of this layer
    const size_t half = (size_t)1 << nv;
    std::vector<HFr> x, evals;
    if (nv == 0) {
      evals = top;  // (p_l, p_r, q_l, q_r) per fraction
    } else {
      HFr gamma = tr.squeeze_challenge();
      // sum_check_claim (:283-288) and sum_check_expression (:272-281)
      HFr claim = HFr::zero(), power = HFr::one();
      lh_sop expr;
      memset(&expr, 0, sizeof(expr));
      expr.global_eq = 0;
      std::vector<const Fr*> polys;
      for (size_t b = 0; b < B; b++) {
        claim += claimed_p[b] * power;
        HFr g_even = power;
        power *= gamma;
        claim += claimed_q[b] * power;
        HFr g_odd = power;
        power *= gamma;
        uint8_t p_l = 4 * b, p_r = 4 * b + 1, q_l = 4 * b + 2, q_r = 4 * b + 3;
        uint32_t m = expr.num_terms;
        memcpy(&expr.coeff[m], &g_even, 32);
        expr.num_factors[m] = 2, expr.factor[m][0] = p_l, expr.factor[m][1] = q_r;
        memcpy(&expr.coeff[m + 1], &g_even, 32);
        expr.num_factors[m + 1] = 2, expr.factor[m + 1][0] = p_r, expr.factor[m + 1][1] = q_l;
        memcpy(&expr.coeff[m + 2], &g_odd, 32);
        expr.num_factors[m + 2] = 2, expr.factor[m + 2][0] = q_l, expr.factor[m + 2][1] = q_r;
        expr.num_terms += 3;
        polys.push_back(lp[h][b]);
        polys.push_back(lp[h][b] + half);
        polys.push_back(lq[h][b]);
        polys.push_back(lq[h][b] + half);
      }
      SumCheckResult sc = sum_check_prove(c, LH_SC_EVALUATIONS, nv, expr, polys.data(), polys.size(), y.data(), 1,
                                          claim, tr, true);
      x = sc.challenges;
      evals = sc.evals;
    }
    tr.write_field_elements(evals);
    HFr mu = tr.squeeze_challenge();
    for (size_t b = 0; b < B; b++) {  // layer_down_claim (:290-296)
      const HFr &p_l = evals[4 * b], &p_r = evals[4 * b + 1], &q_l = evals[4 * b + 2], &q_r = evals[4 * b + 3];
      claimed_p[b] = p_l + mu * (p_r - p_l);
      claimed_q[b] = q_l + mu * (q_r - q_l);
    }
    x.push_back(mu);
    y = x;
  }
  return FracSumCheckResult{claimed_p, claimed_q, y};
}

// ------------------------------------------------------------------ grand product (Lasso memory check)
// Product-only layered circuit; schedule in oracle/pyref/gkr.py::prove_grand_product.
GrandProductResult prove_grand_product(Ctx& c, size_t B, const Fr* const* d_leaves, const size_t* num_vars,
                                       Transcript& tr, const Fr* const* d_level_up, const uint8_t* plus_one) {
  LH_REQUIRE(B != 0, LH_ERR_ARG, "grand product: no trees");
  size_t max_depth = 0;
  for (size_t b = 0; b < B; b++) {
    LH_REQUIRE(num_vars[b] >= 1 && num_vars[b] < 32, LH_ERR_ARG, "grand product: every tree needs >= 2 leaves");
    max_depth = std::max(max_depth, num_vars[b]);
  }
  LH_REQUIRE(2 * B + 1 <= (size_t)SC_MAX_TABLES && B <= LH_SC_MAX_TERMS, LH_ERR_ARG,
             "grand product: too many trees for one round kernel");
  ArenaScope scope(c.arena);
  // Inside a sharded proof (dev.hpp Shard) a level of 2^(h+1) nodes is held in shards while it has more than
  // shard_bit + rho variables: Layer::up (fractional_sum_check.rs:62-85; here v = l * r) pairs node i with node i + half,
  // the top index bit, which is local to a shard.  At the replication point the level is exchanged once (the shard bits
  // have become its top bits: a concatenation) and everything above is computed redundantly on every rank.
  const Shard sh(c);
  // level[b][h]: array with 2^(h+1) nodes, h = 0 (top, two nodes) .. depth-1 (the leaves)
  std::vector<std::vector<const Fr*>> level(B);
  {
    const size_t SMALL = 9;  // levels with <= 2^(SMALL+1) nodes are finished by one workgroup per tree
    std::vector<const Fr*> top_in(B);
    std::vector<Fr*> top_out(B);
    std::vector<int> top_H(B);
    std::vector<size_t> cur_h(B);  // lowest computed level of every tree
    size_t max_h = 0;
    for (size_t b = 0; b < B; b++) {
      level[b].resize(num_vars[b]);
      level[b][num_vars[b] - 1] = d_leaves[b];
      size_t h = num_vars[b] - 1;
      if (d_level_up && d_level_up[b] && h > SMALL && (!sh.on || sh.sharded(h))) {
        level[b][h - 1] = d_level_up[b];
        h--;
      }
      cur_h[b] = h;
      max_h = std::max(max_h, h);
    }
    // level by level, the trees of equal size in one launch
    for (size_t h = max_h; h >= 1 && (h > SMALL || sh.sharded(h + 1)); h--) {
      std::vector<const Fr*> ins;
      std::vector<Fr*> outs;
      std::vector<size_t> who;
      const bool in_sh = sh.sharded(h + 1), out_sh = sh.sharded(h);
      const size_t half = (size_t)1 << (in_sh ? h - sh.rho : h);
      for (size_t b = 0; b < B; b++) {
        if (cur_h[b] != h) continue;
        ins.push_back(level[b][h]);
        who.push_back(b);
        cur_h[b] = h - 1;
      }
      if (ins.empty()) continue;
      if (in_sh && !out_sh) {
        // replication point: the products of all trees into one block, one all-gather
        Fr* block = c.arena.alloc_n<Fr>(ins.size() * half);
        std::vector<Fr*> rep(ins.size());
        for (size_t k = 0; k < ins.size(); k++) {
          outs.push_back(block + k * half);
          rep[k] = c.arena.alloc_n<Fr>((size_t)1 << h);
        }
        k_tree_up_multi(c, ins.data(), outs.data(), ins.size(), half);
        comm_gather_tables(c, block, ins.size(), half, half, rep.data());
        c.route.v[RouteStats::SHARD_EXCHANGES]++;
        for (size_t k = 0; k < ins.size(); k++) level[who[k]][h - 1] = rep[k];
      } else {
        for (size_t k = 0; k < ins.size(); k++) {
          outs.push_back(c.arena.alloc_n<Fr>(half));
          level[who[k]][h - 1] = outs.back();
        }
        k_tree_up_multi(c, ins.data(), outs.data(), ins.size(), half);
      }
    }
    for (size_t b = 0; b < B; b++) {
      const size_t h = cur_h[b];
      // h <= SMALL: levels h-1 .. 0 in one go
      LH_REQUIRE(h <= SMALL && !sh.sharded(h + 1), LH_ERR_ARG, "grand product: internal level mismatch");
      Fr* tops = c.arena.alloc_n<Fr>(((size_t)2 << h));
      top_in[b] = level[b][h];
      top_out[b] = tops;
      top_H[b] = (int)h;
      for (size_t k = 0; k < h; k++) level[b][k] = tops + (((size_t)2 << k) - 2);
    }
    k_tree_tops(c, top_in.data(), top_out.data(), top_H.data(), B);
  }
  std::vector<HFr> top(2 * B);
  {
    std::vector<const Fr*> heads;
    for (size_t b = 0; b < B; b++) heads.push_back(level[b][0]);
    Fr* out = (Fr*)c.pin(2 * SC_MAX_TABLES * sizeof(Fr));
    k_gather_heads(c, heads.data(), heads.size(), 2, out);
    memcpy(top.data(), out, 2 * B * sizeof(Fr));
  }
  GrandProductResult res;
  res.roots.resize(B);
  res.claims.resize(B);
  res.points.resize(B);
  for (size_t b = 0; b < B; b++) res.roots[b] = top[2 * b] * top[2 * b + 1];
  tr.write_field_elements(res.roots);

  std::vector<HFr> claims = res.roots, y;
  HFr resident_lam;
  bool have_resident_lam = false;
  // the layers near the roots in ONE resident launch (kernels_gkr.hip; Options::gkr_resident): every layer from h = 1 up
  // whose tables fit, as long as it is an ordinary layer (all trees given, no (A, A + 1) leaf pairs, not sharded)
  GkrResident resident(c);
  std::vector<GkrLayerDev> resident_layers;
  if (c.opt.gkr_resident && c.opt.sc_tail && c.opt.sc_eq_factoring) {
    std::vector<GkrLayerDev> layers;
    for (size_t h = 1; h < max_depth; h++) {
      GkrLayerDev L;
      memset(&L, 0, sizeof(L));
      uint32_t nb = 0;
      bool ok = !sh.sharded(h + 1) && k_gkr_resident_geometry((uint32_t)h, &L.g, &L.s_log);
      bool any_leaf = false, all_leaf = true;
      for (size_t b = 0; b < B && ok; b++) {
        if (num_vars[b] <= h) continue;
        ok = nb < (uint32_t)GKR_MAX_TREES && level[b][h] != nullptr;
        if (!ok) break;
        any_leaf = any_leaf || num_vars[b] == h + 1;
        all_leaf = all_leaf && num_vars[b] == h + 1;
        L.lv[nb++] = level[b][h];
      }
      // (a layer at which trees end may be a paired leaf layer: those keep their own kernel)
      if (ok && plus_one && any_leaf && all_leaf) ok = false;
      if (!ok || nb == 0) break;
      L.h = (uint32_t)h, L.B = nb;
      layers.push_back(L);
    }
    resident_layers.swap(layers);
  }
  static const int hook_at = getenv("LH_GKR_HOOK_AT") ? atoi(getenv("LH_GKR_HOOK_AT")) : 0;  // development: layer at which the hook fires
  if (c.gkr_hook && hook_at <= 0) {  // (the trees are built: from here on the small layers leave most of the chip idle)
    // (before the resident launch: what the hook starts on another stream waits for an event recorded HERE on this
    // ctx's stream - behind the resident kernel it would wait for the whole resident phase)
    std::function<void()> hook;
    hook.swap(c.gkr_hook);
    hook();
  }
  if (!resident_layers.empty()) resident.launch(resident_layers);
  for (size_t h = 0; h < max_depth; h++) {
    std::vector<size_t> active;
    for (size_t b = 0; b < B; b++)
      if (num_vars[b] > h) active.push_back(b);
    if (c.gkr_hook && hook_at > 0 && (int)h >= hook_at && !resident.live) {
      std::function<void()> hook;
      hook.swap(c.gkr_hook);
      hook();
    }
    if (h >= 1 && resident.live && h <= resident.H) {
      // a resident layer: same transcript schedule, the sum-check's device half is already running
      HFr lam = tr.squeeze_challenge();
      HFr claim = HFr::zero(), power = HFr::one();
      std::vector<HFr> coeff;
      for (size_t b : active) {
        claim += claims[b] * power;
        coeff.push_back(power);
        power *= lam;
      }
      std::vector<HFr> x, evals;
      if (resident.layer(h, coeff, y, claim, tr, x, evals)) {
        c.route.v[RouteStats::RESIDENT_LAYERS]++;
        tr.write_field_elements(evals);
        HFr mu = tr.squeeze_challenge();
        x.push_back(mu);
        y = x;
        for (size_t k = 0; k < active.size(); k++) {
          const size_t b = active[k];
          const HFr &l = evals[2 * k], &r = evals[2 * k + 1];
          claims[b] = l + mu * (r - l);
          if (num_vars[b] == h + 1) {
            res.claims[b] = claims[b];
            res.points[b] = y;
          }
        }
        continue;
      }
      // (degenerate challenge: the kernel is gone; this layer and the following ones take the launched path - lambda is
      // squeezed already)
      resident_lam = lam, have_resident_lam = true;
    }
    const bool layer_sh = sh.sharded(h + 1);  // this layer's tables (h variables each) are shards
    const size_t half = (size_t)1 << (layer_sh ? h - sh.rho : h);
    std::vector<HFr> x, evals;
    if (h == 0) {
      for (size_t b : active) {
        evals.push_back(top[2 * b]);
        evals.push_back(top[2 * b + 1]);
      }
    } else {
      HFr lam = have_resident_lam ? resident_lam : tr.squeeze_challenge();
      have_resident_lam = false;
      HFr claim = HFr::zero(), power = HFr::one();
      lh_sop expr;
      memset(&expr, 0, sizeof(expr));
      expr.global_eq = 0;
      std::vector<const Fr*> polys;
      // leaf layer of (A, A + 1) tree pairs (`plus_one`): every active tree is at its leaf level and they pair up
      bool pairs = plus_one != nullptr && active.size() % 2 == 0 && active.size() / 2 <= (size_t)SC_RW_MAX_PAIRS;
      for (size_t k = 0; k < active.size() && pairs; k++) {
        const size_t b = active[k];
        pairs = num_vars[b] == h + 1 && (k % 2 == 0 ? !plus_one[b] : (plus_one[b] && active[k - 1] + 1 == b));
      }
      if (pairs) {
        // c_A l r + c_B (l + 1)(r + 1) = cs (l + k)(r + k) + c_B (1 - k),  cs = c_A + c_B, k = c_B / cs: only the A tables
        // are read and bound (dev.hpp ScRwRound); the same expression written out as products over the A tables, with its
        // constant term, serves the small rounds (LDS kernel, resident tail) and the degenerate cs = 0
        const size_t P = active.size() / 2;
        ScRwPairs rw;
        rw.num_pairs = (uint32_t)P;
        rw.const_total = HFr::zero();
        HFr cw_sum = HFr::zero();
        bool degenerate = false;
        uint32_t t = 0;
        for (size_t i = 0; i < P; i++) {
          const size_t a = active[2 * i], bb = active[2 * i + 1];
          const HFr c_a = power, c_b = power * lam;
          claim += claims[a] * c_a + claims[bb] * c_b;
          power = c_b * lam;
          const HFr cs = c_a + c_b;
          degenerate = degenerate || cs.is_zero();
          rw.cs[i] = cs;
          rw.k[i] = cs.is_zero() ? HFr::zero() : c_b * cs.inv();
          rw.const_total += c_b * (HFr::one() - rw.k[i]);
          cw_sum += c_b;
          const uint8_t li = (uint8_t)(2 * i), ri = (uint8_t)(2 * i + 1);
          memcpy(&expr.coeff[t], &cs, 32), expr.num_factors[t] = 2, expr.factor[t][0] = li, expr.factor[t][1] = ri, t++;
          memcpy(&expr.coeff[t], &c_b, 32), expr.num_factors[t] = 1, expr.factor[t][0] = li, t++;
          memcpy(&expr.coeff[t], &c_b, 32), expr.num_factors[t] = 1, expr.factor[t][0] = ri, t++;
          polys.push_back(level[a][h]);
          polys.push_back(level[a][h] + half);
        }
        memcpy(&expr.coeff[t], &cw_sum, 32), expr.num_factors[t] = 0, t++;
        expr.num_terms = t;
        SumCheckResult sc = sum_check_prove(c, LH_SC_EVALUATIONS, h, expr, polys.data(), polys.size(), y.data(), 1, claim, tr,
                                            true, degenerate ? nullptr : &rw, layer_sh);
        x = sc.challenges;
        for (size_t i = 0; i < P; i++) {  // evaluations of the B tables: those of the A tables + 1
          const HFr l = sc.evals[2 * i], r = sc.evals[2 * i + 1];
          evals.push_back(l), evals.push_back(r);
          evals.push_back(l + HFr::one()), evals.push_back(r + HFr::one());
        }
      } else {
      for (size_t k = 0; k < active.size(); k++) {
        size_t b = active[k];
        LH_REQUIRE(level[b][h] != nullptr, LH_ERR_ARG, "grand product: a tree given without leaves is not at a paired leaf layer");
        claim += claims[b] * power;
        memcpy(&expr.coeff[k], &power, 32);
        expr.num_factors[k] = 2;
        expr.factor[k][0] = (uint8_t)(2 * k);
        expr.factor[k][1] = (uint8_t)(2 * k + 1);
        power *= lam;
        polys.push_back(level[b][h]);
        polys.push_back(level[b][h] + half);
      }
      expr.num_terms = (uint32_t)active.size();
      SumCheckResult sc = sum_check_prove(c, LH_SC_EVALUATIONS, h, expr, polys.data(), polys.size(), y.data(), 1, claim, tr,
                                          true, nullptr, layer_sh);
      x = sc.challenges;
      evals = sc.evals;
      }
    }
    tr.write_field_elements(evals);
    HFr mu = tr.squeeze_challenge();
    x.push_back(mu);
    y = x;
    for (size_t k = 0; k < active.size(); k++) {
      size_t b = active[k];
      const HFr &l = evals[2 * k], &r = evals[2 * k + 1];
      claims[b] = l + mu * (r - l);
      if (num_vars[b] == h + 1) {
        res.claims[b] = claims[b];
        res.points[b] = y;
      }
    }
  }
  return res;
}

// ------------------------------------------------------------------ MultilinearKzg
// setup: reference pcs/multilinear/kzg.rs:166-228 with the trapdoor supplied by the caller.
Srs* mkzg_setup(Ctx& c, const HFr* ss, size_t num_vars) {
  LH_REQUIRE(num_vars < 31, LH_ERR_ARG, "setup: num_vars too large");
  Srs* srs = new Srs();
  srs->num_vars = num_vars;
  size_t total = ((size_t)2 << num_vars) - 1;
  LH_HIP(hipMalloc((void**)&srs->d_eqs, total * sizeof(G1Affine)));
  ArenaScope scope(c.arena);
  Fr* scal = c.arena.alloc_n<Fr>(total);
  // eqs[k] = eq table of (s_0..s_{k-1}) with s_{k-1} the top bit (kzg.rs:178-194) == eq_xy(s[..k])
  for (size_t k = 0; k <= num_vars; k++) k_eq_xy(c, (const Fr*)ss, k, scal + (((size_t)1 << k) - 1));
  k_fixed_base_mul_g(c, scal, total, srs->d_eqs);
  return srs;
}

static void check_commit_vars(const Srs& srs, size_t num_vars, const char* what) {
  if (num_vars > srs.num_vars)  // validate_input / err_too_many_variates (pcs/multilinear.rs:27-70)
    throw Error(LH_ERR_INVALID_PCS_PARAM,
                std::string("Too many variates of poly to ") + what + " (param supports variates up to " +
                    std::to_string(srs.num_vars) + " but got " + std::to_string(num_vars) + ")");
}

std::vector<HG1> mkzg_batch_commit(Ctx& c, const Srs& srs, const Fr* const* d_polys, size_t num_polys,
                                   size_t num_vars) {
  check_commit_vars(srs, num_vars, "batch commit");
  const Shard sh(c);
  if (sh.on && sh.sharded(num_vars)) {
    // inside a sharded proof the polys are this rank's shards: each is committed against the rank's share of the level's
    // bases - the chunk-split-then-sum of util/arithmetic/msm.rs:101-114 with the shards as chunks - and the partial
    // commitments are added over the ranks (one exchange per batch)
    const G1Affine* bases = srs_shard_level(c, srs, num_vars);
    std::vector<MsmJob> jobs(num_polys);
    for (size_t i = 0; i < num_polys; i++) jobs[i] = MsmJob{d_polys[i], false, bases, (size_t)1 << (num_vars - sh.rho)};
    std::vector<HG1> out(num_polys);
    msm_batch(c, jobs.data(), num_polys, (G1Affine*)out.data());
    if (num_polys) comm_sum_points(c, out.data(), num_polys);
    return out;
  }
  std::vector<MsmJob> jobs(num_polys);
  const Srs::WinTable* wt = srs_window_table(c, srs, num_vars);
  for (size_t i = 0; i < num_polys; i++) {
    jobs[i] = MsmJob{d_polys[i], false, srs.eq(num_vars), (size_t)1 << num_vars};
    if (wt) jobs[i].win_table = wt->d, jobs[i].win_table_c = wt->c, jobs[i].win_table_W = wt->W;
  }
  std::vector<HG1> out(num_polys);
  msm_batch(c, jobs.data(), num_polys, (G1Affine*)out.data());
  return out;
}

std::vector<HG1> mkzg_batch_commit_u32(Ctx& c, const Srs& srs, const uint32_t* const* d_polys, size_t num_polys,
                                       size_t num_vars) {
  check_commit_vars(srs, num_vars, "batch commit");
  std::vector<MsmJob> jobs(num_polys);
  for (size_t i = 0; i < num_polys; i++)
    jobs[i] = MsmJob{d_polys[i], true, srs.eq(num_vars), (size_t)1 << num_vars};
  std::vector<HG1> out(num_polys);
  msm_batch(c, jobs.data(), num_polys, (G1Affine*)out.data());
  return out;
}

// open: kzg.rs:276-302 + quotients pcs/multilinear.rs:72-107.  The n quotient polynomials are laid out
// back to back (q_i at offset 2^i - 1, exactly the flat SRS layout) and committed as ONE batched MSM.
//
// `small`: the opened poly is a scalar combination of small-valued columns, g' = sum_k coef_k col_k (a Lasso batch opening
// under a linear g).  The quotient operator is linear and the LARGEST quotient (half of all quotient entries) is a plain
// difference of halves, q_top = sum_k coef_k (hi_k - lo_k): its commitment is sum_k coef_k C(hi_k - lo_k) with 17-33-bit
// differences - one or two windows per column, pairs of narrow columns in one pass (MsmJob::pack_shift) - instead of 15
// windows over 2^(n-1) full-size scalars.  Differences are made non-negative by an offset 2^bits; the offsets cost one
// multiple of the level's base sum (Srs::level_sums, computed once).  The lower quotients come from the folded
// remainder as before.
static std::mutex level_sums_mu;

// bases of level `lvl` that belong to this rank (same index split as the tables); built on first use
const G1Affine* srs_shard_level(Ctx& c, const Srs& srs, size_t lvl) {
  const Shard g(c);
  LH_REQUIRE(g.on, LH_ERR_ARG, "srs shard: no sharded proof is running");
  std::lock_guard<std::mutex> lock(level_sums_mu);
  if (srs.shard_rank != (int)g.rank || srs.shard_R != g.R || srs.shard_j != g.j) {
    for (G1Affine* p : srs.shard_levels)
      if (p) {
        (void)hipFree(p);
      }
    srs.shard_levels.assign(srs.num_vars + 1, nullptr);
    srs.shard_level_sums.clear();
    srs.shard_rank = (int)g.rank, srs.shard_R = g.R, srs.shard_j = g.j;
  }
  LH_REQUIRE(lvl <= srs.num_vars && lvl >= g.j + g.rho, LH_ERR_ARG, "srs shard: level is not sharded");
  if (!srs.shard_levels[lvl]) {
    const size_t n_local = (size_t)1 << (lvl - g.rho);
    G1Affine* p = nullptr;
    LH_HIP(hipMalloc((void**)&p, n_local * sizeof(G1Affine)));
    k_shard_extract(c, srs.eq(lvl), n_local, g.j, g.rho, g.rank, sizeof(G1Affine), p);
    c.sync();
    srs.shard_levels[lvl] = p;
  }
  return srs.shard_levels[lvl];
}

// window table of a whole level (MsmJob::win_table), built on first use by the ctx that asks (the table belongs to the
// SRS: every ctx of the process sees it afterwards).  Levels above Options::msm_window_tables, levels too small to matter
// and levels whose table does not fit the device's free memory have none.
const Srs::WinTable* srs_window_table(Ctx& c, const Srs& srs, size_t lvl) {
  if (c.opt.msm_window_tables <= 0 || (int64_t)lvl > c.opt.msm_window_tables || lvl > srs.num_vars || lvl < 6) return nullptr;
  std::lock_guard<std::mutex> lock(level_sums_mu);
  auto it = srs.win_tables.find(lvl);
  if (it != srs.win_tables.end()) return it->second.d ? &it->second : nullptr;
  Srs::WinTable t;
  const size_t n = (size_t)1 << lvl;
  t.c = msm_window_bits(n);
  t.W = (255 + t.c - 1) / t.c;  // (254 bits + the head room of the signed digits)
  const size_t bytes = (size_t)t.W * n * sizeof(G1Affine);
  size_t free_b = 0, total_b = 0;
  if ((size_t)t.W * n >= ((size_t)1 << 31) || hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 4 ||
      hipMalloc((void**)&t.d, bytes) != hipSuccess) {
    (void)hipGetLastError();
    t.d = nullptr;
    srs.win_tables[lvl] = t;  // (remembered: not asked again)
    return nullptr;
  }
  k_msm_window_table(c, srs.eq(lvl), n, t.c, t.W, t.d);
  c.sync();
  srs.win_tables[lvl] = t;
  return &srs.win_tables[lvl];
}

// ------------------------------------------------------------------ the challenge-free half of the column route
// Which columns take part, how wide their differences are, how many levels go column by column and the MSM jobs that
// commit them depend on the witness columns alone; only the linear combination of the jobs' results uses the batch
// opening's coefficients and the fold weights.  mkzg_open builds this plan itself - or finds it already committed by
// open_precommit_start (a helper ctx on its own stream and host thread, beside the GKR phase).
struct ColTerm {  // result of job `job` (or its second output) times coef[k] * w(sidx) * factor goes into the commitment
  size_t job;
  bool second;
  size_t k, sidx;
  int factor;  // -1: only the low half is populated (hi - lo = -lo); 1; 65536: the high limb of a 33-bit difference
};
struct ColOffset {  // coef[k] * w(sidx) * off: times the level's base sum, subtracted
  size_t k, sidx;
  uint64_t off;
};
struct ColLevel {
  size_t level = 0;  // quotient level (number of variables)
  std::vector<ColTerm> terms;
  std::vector<ColOffset> offsets;
  bool need_sum = false;
  size_t sum_job = (size_t)-1;  // the all-ones MSM when the level's base sum is not cached yet
  HG1 base_sum;
};
struct ColumnPlan {
  std::vector<uint32_t> ors;  // per column: OR of its entries (a bound when the caller knows the width)
  size_t depth = 0;           // quotient levels that go column by column
  std::vector<MsmJob> jobs;
  std::vector<ColLevel> levels;
  std::vector<HG1> seconds;   // second outputs of packed jobs (MsmJob::out_second points in here: sized before the jobs)
};
static inline uint32_t bits_of_u32(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

// ---- which quotient levels go column by column: the largest always; the second largest too when few columns take
// part.  After d folds the remainder is sum_s w_s g'[s 2^(n-d) + .] over the 2^d settings s of the top d index bits
// (w_s = the product of x_j or 1 - x_j over those bits), so the quotient of level n-1-d is the same kind of sum of
// 2^d differences of sub-columns per column: 2^d times the passes of the top level, against ~15 windows.
static void column_shape(Ctx& c, const std::vector<SmallPoly>& cols, const std::vector<char>& zero, size_t n,
                         size_t num_vars, size_t cut, ColumnPlan& plan) {
  const size_t K = cols.size(), half_top = n >> 1;
  std::vector<uint32_t>& ors = plan.ors;
  ors.assign(K, 0);
  std::map<size_t, std::vector<size_t>> by_len;  // one OR pass per column length (normally none: the widths are known)
  for (size_t k = 0; k < K; k++) {
    // (every column that can reach past the low half of a sub-column at either depth needs its width)
    if (cols[k].len <= (half_top >> 1) || zero[k]) continue;
    if (cols[k].bits) ors[k] = cols[k].bits >= 32 ? 0xffffffffu : (1u << cols[k].bits) - 1u;  // known bound
    else by_len[cols[k].len].push_back(k);
  }
  for (const auto& grp : by_len) {
    std::vector<const uint32_t*> ptrs;
    for (size_t k : grp.second) ptrs.push_back(cols[k].ptr);
    std::vector<uint32_t> o(ptrs.size(), 0);
    k_or_u32(c, ptrs.data(), ptrs.size(), grp.first, o.data());
    for (size_t f = 0; f < ptrs.size(); f++) ors[grp.second[f]] = o[f];
  }
  size_t passes = 0, narrow_cols = 0;
  for (size_t k = 0; k < K; k++) {
    if (cols[k].len <= half_top || zero[k] || !ors[k]) continue;
    const uint32_t b1 = bits_of_u32(ors[k]) + 1;
    if (b1 > 32) passes += 2;
    else if (b1 <= MSM_PACK_MAX_BITS - 4) narrow_cols++;
    else passes += 1;
  }
  passes += (narrow_cols + 1) / 2;
  const int forced_depth = (int)c.opt.open_small_depth;  // 1 or 2 levels column by column, whatever the shape
  plan.depth = 1;
  if (num_vars >= cut + 3 && (forced_depth ? forced_depth >= 2 : 2 * passes <= 10)) plan.depth = 2;
}

// the MSM jobs of the column-wise levels (difference columns, limbs, packed pairs, the levels' base sums); temporaries
// from c's arena, kernels on c's stream
static void column_jobs(Ctx& c, const Srs& srs, const std::vector<SmallPoly>& cols, const std::vector<char>& zero,
                        size_t num_vars, size_t lsh, bool sharded, const std::function<const G1Affine*(size_t)>& level_bases,
                        ColumnPlan& plan) {
  const size_t depth = plan.depth;
  const std::vector<uint32_t>& ors = plan.ors;
  std::vector<MsmJob>& jobs = plan.jobs;
  plan.levels.assign(depth, ColLevel());
  size_t num_seconds = 0;
  for (size_t d = 0; d < depth; d++) num_seconds += cols.size() << d;
  plan.seconds.assign(num_seconds, HG1());
  size_t next_second = 0;
  for (size_t d = 0; d < depth; d++) {
    ColLevel& cl = plan.levels[d];
    cl.level = num_vars - 1 - d;
    const size_t half = (size_t)1 << (cl.level - lsh);
    const G1Affine* bases = level_bases(cl.level);
    struct Narrow {
      uint32_t bits;  // of the shifted difference
      uint32_t* col;
      size_t k, sidx;
    };
    std::vector<Narrow> narrow;
    for (size_t sidx = 0; sidx < ((size_t)1 << d); sidx++) {
      const size_t off_idx = sidx << (cl.level + 1 - lsh);
      for (size_t k = 0; k < cols.size(); k++) {
        const SmallPoly& sp = cols[k];
        if (zero[k] || sp.len <= off_idx) continue;
        const uint32_t* sub = sp.ptr + off_idx;
        const size_t sub_len = std::min(sp.len - off_idx, half << 1);
        if (sub_len <= half) {  // only the low half is populated: hi - lo = -lo, no offset
          jobs.push_back(MsmJob{sub, true, bases, sub_len});
          if (sp.bits) jobs.back().known_bits = sp.bits;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, -1});
          continue;
        }
        const uint32_t b = bits_of_u32(ors[k]);
        if (!b) continue;  // an all-zero column
        const uint64_t off = (uint64_t)1 << b;
        cl.offsets.push_back(ColOffset{k, sidx, off});
        cl.need_sum = true;
        if (b + 1 > 32) {  // 33-bit shifted differences: a 16-bit limb and a 17-bit limb
          uint32_t* lo = c.arena.alloc_n<uint32_t>(half);
          uint32_t* hi = c.arena.alloc_n<uint32_t>(half);
          k_delta_u32(c, sub, sub_len, half, off, lo, hi);
          jobs.push_back(MsmJob{lo, true, bases, half});
          jobs.back().known_bits = 16;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, 1});
          jobs.push_back(MsmJob{hi, true, bases, half});
          jobs.back().known_bits = b + 1 - 16;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, 65536});
          continue;
        }
        uint32_t* dcol = c.arena.alloc_n<uint32_t>(half);
        k_delta_u32(c, sub, sub_len, half, off, dcol, nullptr);
        if (b + 1 <= MSM_PACK_MAX_BITS - 4) {
          narrow.push_back(Narrow{b + 1, dcol, k, sidx});
        } else {
          jobs.push_back(MsmJob{dcol, true, bases, half});
          jobs.back().known_bits = b + 1;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, 1});
        }
      }
    }
    // narrow columns two by two (narrowest first) while the packed value stays within MSM_PACK_MAX_BITS
    std::stable_sort(narrow.begin(), narrow.end(), [](const Narrow& x, const Narrow& y) { return x.bits < y.bits; });
    for (size_t i = 0; i < narrow.size(); i++) {
      const Narrow& x = narrow[i];
      const uint32_t shift = std::max(x.bits, 4u);
      // (a packed pair saves one pass over `half` points and reduces a bucket set indexed by the packed value - ~5 curve
      // additions' worth per bucket: on a rank's shard of a level the second can outweigh the first)
      if (i + 1 < narrow.size() && shift + narrow[i + 1].bits <= MSM_PACK_MAX_BITS &&
          half >= ((size_t)MSM_PACK_MIN_POINTS_PER_BUCKET << (shift + narrow[i + 1].bits))) {
        const Narrow& y = narrow[i + 1];
        uint32_t* packed = c.arena.alloc_n<uint32_t>(half);
        k_pack_u32(c, x.col, y.col, shift, half, packed);
        MsmJob jb{packed, true, bases, half};
        jb.pack_shift = shift;
        jb.known_bits = shift + y.bits;
        jb.out_second = (G1Affine*)&plan.seconds[next_second++];
        jobs.push_back(jb);
        cl.terms.push_back(ColTerm{jobs.size() - 1, false, x.k, x.sidx, 1});
        cl.terms.push_back(ColTerm{jobs.size() - 1, true, y.k, y.sidx, 1});
        i++;
      } else {
        jobs.push_back(MsmJob{x.col, true, bases, half});
        jobs.back().known_bits = x.bits;
        cl.terms.push_back(ColTerm{jobs.size() - 1, false, x.k, x.sidx, 1});
      }
    }
    // the level's base sum (an MSM with all-one scalars, once per SRS and level)
    if (cl.need_sum) {
      std::lock_guard<std::mutex> lock(level_sums_mu);
      std::map<size_t, HG1>& sums = sharded ? srs.shard_level_sums : srs.level_sums;  // (sharded: of this rank's share)
      auto it = sums.find(cl.level);
      if (it != sums.end()) {
        cl.base_sum = it->second;
      } else {
        uint32_t* ones = c.arena.alloc_n<uint32_t>(half);
        k_fill_u32(c, ones, 1u, half);
        jobs.push_back(MsmJob{ones, true, bases, half});
        jobs.back().known_bits = 1;
        cl.sum_job = jobs.size() - 1;
      }
    }
  }
}
// after the jobs ran: the levels' base sums that were computed go into the SRS's cache
static void column_sums_store(const Srs& srs, bool sharded, ColumnPlan& plan, const HG1* out) {
  for (ColLevel& cl : plan.levels)
    if (cl.sum_job != (size_t)-1) {
      cl.base_sum = out[cl.sum_job];
      std::lock_guard<std::mutex> lock(level_sums_mu);
      (sharded ? srs.shard_level_sums : srs.level_sums)[cl.level] = cl.base_sum;
    }
}

// does an opening over these columns take the column route (mkzg_open's gating, on the local sizes)
static bool column_route_on(const Ctx& c, const std::vector<SmallPoly>& cols, const std::vector<char>& zero, size_t num_vars,
                            size_t lsh, size_t n, bool sharded, size_t cut) {
  if (num_vars - lsh < (size_t)c.opt.open_small_min_vars) {
    // below the general threshold the route still pays when only a few columns take part (the range check: two dim
    // and two read_ts columns - 2^20 lookups 11.7 -> 11.0 ms; the AND table's twelve columns lose there)
    size_t full = 0;
    for (size_t k = 0; k < cols.size(); k++) full += cols[k].len > (n >> 1) && !zero[k];
    if (!(num_vars - lsh >= 17 && full <= 4 && !c.opt.open_small_min_vars_forced)) return false;
  }
  if (num_vars < 2 || cols.empty()) return false;
  if (sharded && num_vars < cut + 2) return false;  // (the column-wise levels must be sharded ones)
  return true;
}

// ---- the plan committed ahead (Options::open_precommit): the helper ctx builds the same plan from the same columns and
// runs its jobs on its own stream, driven by its own host thread, while the ctx goes through the sum-checks.  mkzg_open
// takes the results only if columns, widths and depth are what it arrives at itself; otherwise it commits as before.
struct OpenPrecommit {
  HostWorker* worker = nullptr;  // the helper ctx's host thread, busy with this plan until wait() returns
  const Srs* srs = nullptr;
  size_t num_vars = 0;
  std::vector<SmallPoly> cols;
  std::vector<char> zero;
  ColumnPlan plan;
  std::vector<HG1> out;
  bool ok = false;
  std::string err;
  void join() {
    if (worker) worker->wait();
    worker = nullptr;
  }
  ~OpenPrecommit() { join(); }
};
void open_precommit_cancel(Ctx& c) {
  if (!c.precommit) return;
  delete (OpenPrecommit*)c.precommit;  // (joins)
  c.precommit = nullptr;
}
// the columns of a batch opening whose polys are all small-valued columns: `used` polys, a linear column's coefficient
// handed to the columns it combines, the same column under two names merged.  coef == nullptr: the structure alone
// (every column that survives gets coefficient one)
static bool small_open_columns(const SmallPoly* small, size_t num_polys, const lh_evaluation* evals, size_t num_evals,
                               size_t n, const HFr* coef_in, SmallOpen& so) {
  for (size_t i = 0; i < num_evals; i++)
    if (evals[i].poly >= num_polys || !small[evals[i].poly].ptr) return false;
  std::vector<HFr> coef(num_polys, HFr::zero());
  std::vector<char> used(num_polys, 0);
  for (size_t i = 0; i < num_evals; i++) {
    coef[evals[i].poly] = coef_in ? coef_in[evals[i].poly] : HFr::one();
    used[evals[i].poly] = 1;
  }
  // a column that is a linear combination of other opened columns hands its coefficient over to them
  for (size_t pi = 0; pi < num_polys; pi++) {
    const SmallLinear* lin = small[pi].linear;
    if (!used[pi] || !lin) continue;
    bool ok = !lin->poly.empty() && lin->poly.size() == lin->coeff.size();
    for (size_t k = 0; k < lin->poly.size() && ok; k++) ok = lin->poly[k] < num_polys && lin->poly[k] != pi && used[lin->poly[k]];
    if (!ok) continue;
    if (coef_in)
      for (size_t k = 0; k < lin->poly.size(); k++) coef[lin->poly[k]] += coef[pi] * lin->coeff[k];
    coef[pi] = HFr::zero();
    used[pi] = 0;
  }
  for (size_t pi = 0; pi < num_polys; pi++) {
    if (!used[pi]) continue;
    const SmallPoly sp{small[pi].ptr, std::min(small[pi].len, n), small[pi].bits};
    size_t k = 0;  // the same column under two names (Lasso's E = dim for an identity subtable): one coefficient
    while (k < so.cols.size() && !(so.cols[k].ptr == sp.ptr && so.cols[k].len == sp.len)) k++;
    if (k < so.cols.size()) {
      if (coef_in) so.coef[k] += coef[pi];
      so.cols[k].bits = so.cols[k].bits && sp.bits ? std::max(so.cols[k].bits, sp.bits) : 0;
    } else {
      so.cols.push_back(sp), so.coef.push_back(coef[pi]);
    }
  }
  return true;
}
void open_precommit_start(Ctx& c, const Srs& srs, size_t num_vars, const SmallPoly* small, size_t num_polys,
                          const lh_evaluation* evals, size_t num_evals) {
  open_precommit_cancel(c);
  // (the option is the smallest proof that does it; default 1: every proof that takes the column route - 2^17..2^19 range
  // lookups gain too: 7.0 -> 6.4-6.9, 8.1-9.1 -> 7.3-7.7, 9.7-9.9 -> 9.3-9.4 ms)
  if (c.opt.open_precommit <= 0 || (int64_t)num_vars < c.opt.open_precommit || !small || num_vars > srs.num_vars) return;
  // inside a sharded proof the columns are this rank's shards and the column-wise levels are committed against this rank's
  // share of the levels' bases (mkzg_open's geometry): nothing in this half of the route needs a peer
  const Shard sh(c);
  const bool sharded = sh.sharded(num_vars);
  const size_t lsh = sharded ? sh.rho : 0, cut = sharded ? sh.j + sh.rho : 0;
  const size_t n = (size_t)1 << (num_vars - lsh);
  SmallOpen so;
  if (!small_open_columns(small, num_polys, evals, num_evals, n, nullptr, so)) return;
  std::vector<char> zero(so.cols.size(), 0);
  if (!column_route_on(c, so.cols, zero, num_vars, lsh, n, sharded, cut)) return;
  // the bases of the (at most two) column-wise levels, resolved here: a rank's share of a level is made on first use by a
  // ctx that knows the shard geometry - this one
  std::vector<const G1Affine*> bases_of(num_vars + 1, nullptr);
  for (size_t d = 0; d < 2 && d + 1 <= num_vars; d++) {
    const size_t lvl = num_vars - 1 - d;
    if (sharded && lvl < cut) break;
    bases_of[lvl] = sharded ? srs_shard_level(c, srs, lvl) : srs.eq(lvl);
  }
  Ctx& h = ctx_helper(c);
  h.opt = c.opt;
  h.prof = c.prof;
  h.prof_recs.clear();
  OpenPrecommit* pc = new OpenPrecommit();
  pc->srs = &srs, pc->num_vars = num_vars, pc->cols = so.cols, pc->zero = zero;
  c.precommit = pc;
  // the witness columns are written by kernels queued on this ctx's stream and read by the helper's: an event between the
  // streams (not a host sync), and the helper's long-lived host thread (not a thread per proof)
  if (!c.handoff_ev) LH_HIP(hipEventCreateWithFlags(&c.handoff_ev, hipEventDisableTiming));
  LH_HIP(hipEventRecord(c.handoff_ev, c.stream));
  hipEvent_t handoff = c.handoff_ev;
  const int device = c.device;
  if (!h.worker) h.worker = new HostWorker();
  pc->worker = h.worker;
  h.worker->submit([pc, &h, &srs, num_vars, n, lsh, cut, sharded, bases_of, device, handoff] {
    try {
      LH_HIP(hipSetDevice(device));
      LH_HIP(hipStreamWaitEvent(h.stream, handoff, 0));
      ArenaScope scope(h.arena);
      column_shape(h, pc->cols, pc->zero, n, num_vars, cut, pc->plan);
      column_jobs(h, srs, pc->cols, pc->zero, num_vars, lsh, sharded,
                  [&bases_of](size_t lvl) {
                    LH_REQUIRE(lvl < bases_of.size() && bases_of[lvl], LH_ERR_ARG, "open precommit: level without bases");
                    return bases_of[lvl];
                  },
                  pc->plan);
      pc->out.resize(pc->plan.jobs.size());
      if (!pc->plan.jobs.empty()) msm_batch(h, pc->plan.jobs.data(), pc->plan.jobs.size(), (G1Affine*)pc->out.data());
      h.sync();
      column_sums_store(srs, sharded, pc->plan, pc->out.data());
      pc->ok = true;
    } catch (const std::exception& e) {
      pc->err = e.what();
    } catch (...) {
      pc->err = "unknown error";
    }
  });
  if (c.prof) pc->join();  // a profiled prove keeps its kernels one at a time (the records are merged when taken)
}
// the precommitted plan if it is the plan this opening would build (same SRS, columns, zero pattern, widths, depth)
static std::unique_ptr<OpenPrecommit> open_precommit_take(Ctx& c, const Srs& srs, size_t num_vars,
                                                          const std::vector<SmallPoly>& cols, const std::vector<char>& zero,
                                                          const ColumnPlan& own) {
  std::unique_ptr<OpenPrecommit> pc((OpenPrecommit*)c.precommit);
  c.precommit = nullptr;
  if (!pc) return nullptr;
  pc->join();
  if (c.prof && c.helper) {
    c.prof_recs.insert(c.prof_recs.end(), c.helper->prof_recs.begin(), c.helper->prof_recs.end());
    c.helper->prof_recs.clear();
  }
  bool same = pc->ok && pc->srs == &srs && pc->num_vars == num_vars && pc->cols.size() == cols.size() && pc->zero == zero &&
              pc->plan.depth == own.depth && pc->plan.ors == own.ors;
  for (size_t k = 0; same && k < cols.size(); k++)
    same = pc->cols[k].ptr == cols[k].ptr && pc->cols[k].len == cols[k].len && pc->cols[k].bits == cols[k].bits;
  if (!same) return nullptr;
  return pc;
}

// Inside a sharded proof (dev.hpp Shard) `d_poly` / the small columns are this rank's shards.  The quotient of level i
// is a difference of halves - the top index bit, local to a shard while i >= shard_bit + rho: those levels are computed
// and committed shard by shard against this rank's share of the level's bases (the chunk-split-then-sum of
// util/arithmetic/msm.rs:101-114, the chunks being the shards) and the partial commitments are added; the remainder at
// the replication point is exchanged once (2^(shard_bit + rho) entries) and the small levels run on every rank.
HFr mkzg_open(Ctx& c, const Srs& srs, const Fr* d_poly, size_t num_vars, const HFr* point, Transcript& tr,
              const SmallOpen* small) {
  check_commit_vars(srs, num_vars, "open");
  const Shard sh(c);
  const bool sharded = sh.sharded(num_vars);
  const size_t cut = sharded ? sh.j + sh.rho : 0;  // quotient levels >= cut are held in shards
  const SmallOpen* given = small;
  const size_t lsh = sharded ? sh.rho : 0;        // local length of level i: 2^(i - lsh), i >= cut
  const size_t n = (size_t)1 << (num_vars - lsh);  // entries of the (local) table
  // (which route an opening takes is decided per rank on its local sizes: every route yields the commitment of the rank's
  // shard of each quotient, so ranks may even differ)
  if (small) {
    std::vector<char> z(small->cols.size());
    for (size_t k = 0; k < z.size(); k++) z[k] = small->coef[k].is_zero();
    if (!column_route_on(c, small->cols, z, num_vars, lsh, n, sharded, cut)) small = nullptr;
  }
  if (!small) open_precommit_cancel(c);  // (whatever was committed ahead is not what this opening needs)
  ArenaScope scope(c.arena);
  if (!d_poly) {
    LH_REQUIRE(given && !given->merged.empty(), LH_ERR_ARG, "open: no polynomial");
    if (!small) {  // the plain route needs g' itself
      Fr* g = c.arena.alloc_n<Fr>(n);
      k_lincomb(c, given->merged.data(), given->merged_w.data(), given->merged.size(), n, g);
      d_poly = g;
    }
  }
  // ---- the challenge-free half of the column route (column_shape / column_jobs above)
  size_t depth = 0;
  ColumnPlan own;
  std::vector<char> zero;
  if (small) {
    zero.resize(small->cols.size());
    for (size_t k = 0; k < zero.size(); k++) zero[k] = small->coef[k].is_zero();
    column_shape(c, small->cols, zero, n, num_vars, cut, own);
    depth = own.depth;
  }
  // development (LH_OPEN_SMALL_CHECK): the column-wise levels whose quotient exists are committed the plain way too and
  // compared (stderr)
  static const bool self_check_env = getenv("LH_OPEN_SMALL_CHECK") != nullptr;
  const bool self_check = self_check_env && small;
  const size_t check_from = d_poly ? 0 : 1;  // (the first fold of a lazy g' leaves no quotient to compare with)
  // quotients back to back: the sharded levels (local halves) from the top down, then - replicated - the flat layout of
  // the small levels (q_i at offset 2^i - 1)
  Fr* q = c.arena.alloc_n<Fr>(n);  // n - 1 used
  Fr* remA = c.arena.alloc_n<Fr>(std::max<size_t>(n >> 1, 1));
  Fr* remB = c.arena.alloc_n<Fr>(std::max<size_t>(n >> 2, 1));
  std::vector<const Fr*> q_of(num_vars, nullptr);
  const Fr* rem = d_poly;
  size_t q_off = 0;
  for (size_t i = num_vars; i-- > cut;) {
    size_t half = (size_t)1 << (i - lsh);
    Fr* dst = ((num_vars - i) & 1) ? remA : remB;
    const bool keep_q = !(i + depth >= num_vars && !self_check);
    if (!rem)  // first step of the column route straight from the merged tables (g' is never formed)
      k_lincomb_fold(c, small->merged.data(), small->merged_w.data(), small->merged.size(), half, dev(point[i]), dst);
    else
      k_quotient_step(c, rem, half, dev(point[i]), keep_q ? q + q_off : nullptr, dst);
    if (keep_q && rem) q_of[i] = q + q_off;
    q_off += half;
    rem = dst;
  }
  if (sharded) {
    // remainder: 2^cut entries globally, the shard bits on top -> replicate and finish as on one GPU
    const size_t n_rep = (size_t)1 << cut;
    Fr* rep = c.arena.alloc_n<Fr>(n_rep);
    comm_gather_concat(c, rem, (size_t)1 << sh.j, rep);
    c.route.v[RouteStats::SHARD_EXCHANGES]++;
    Fr* q_rep = c.arena.alloc_n<Fr>(n_rep);
    Fr* repA = c.arena.alloc_n<Fr>(std::max<size_t>(n_rep >> 1, 1));
    Fr* repB = c.arena.alloc_n<Fr>(std::max<size_t>(n_rep >> 2, 1));
    rem = rep;
    for (size_t i = cut; i-- > 0;) {
      const size_t half = (size_t)1 << i;
      Fr* dst = ((cut - i) & 1) ? repA : repB;
      k_quotient_step(c, rem, half, dev(point[i]), q_rep + (half - 1), dst);
      q_of[i] = q_rep + (half - 1);
      rem = dst;
    }
  }
  HFr remainder;
  if (num_vars == 0) {
    download(c, &remainder, d_poly, sizeof(Fr));
    return remainder;
  }
  // bases of a level: this rank's share of a sharded level
  auto level_bases = [&](size_t lvl) { return lvl >= cut && sharded ? srs_shard_level(c, srs, lvl) : srs.eq(lvl); };
  const size_t plain = num_vars - depth;
  std::vector<MsmJob> jobs(plain);
  for (size_t i = 0; i < plain; i++) {
    size_t half = (size_t)1 << (i >= cut ? i - lsh : i);
    jobs[i] = MsmJob{q_of[i], false, level_bases(i), half};
    if (sharded && i < cut) {
      // a level below the replication point is the same on every rank: each commits ITS range of the quotient's entries
      // (dev.hpp ReplicatedRange: the chunk-then-sum of util/arithmetic/msm.rs:101-114) and the parts are added with the
      // sharded levels' partial commitments below - 2^cut points x ~20 windows that every rank used to repeat
      const ReplicatedRange rr(sh, half);
      jobs[i] = MsmJob{q_of[i] + rr.first, false, srs.eq(i) + rr.first, rr.count};
    }
    if (small) jobs[i].known_bits = 254;  // quotients of a random combination: full-size scalars, nothing to measure
    if (!sharded)                         // (a rank's share of a level has no window table)
      if (const Srs::WinTable* wt = srs_window_table(c, srs, i))
        jobs[i].win_table = wt->d, jobs[i].win_table_c = wt->c, jobs[i].win_table_W = wt->W;
  }
  // ---- the column-wise levels: committed ahead by open_precommit_start (same columns, widths and depth), or here
  const ColumnPlan* plan = nullptr;
  const std::vector<HG1>* pre_out = nullptr;
  std::unique_ptr<OpenPrecommit> pre;
  if (small) {
    pre = open_precommit_take(c, srs, num_vars, small->cols, zero, own);
    if (pre) {
      plan = &pre->plan, pre_out = &pre->out;
    } else {
      column_jobs(c, srs, small->cols, zero, num_vars, lsh, sharded, level_bases, own);
      plan = &own;
    }
  }
  const size_t col_base = jobs.size();  // (index of the plan's first job in this batch, when it runs here)
  if (plan && !pre_out) jobs.insert(jobs.end(), plan->jobs.begin(), plan->jobs.end());
  c.route.v[RouteStats::OPEN_DEPTH] = (uint32_t)depth;
  c.route.v[RouteStats::OPEN_PASSES] = (uint32_t)(plan ? plan->jobs.size() : 0);
  c.route.v[RouteStats::OPEN_PRECOMMIT] = pre_out ? 1u : 0u;
  const size_t check_base = jobs.size();
  if (self_check)
    for (size_t d = check_from; d < depth; d++) {
      const size_t lvl = num_vars - 1 - d, half = (size_t)1 << (lvl - lsh);
      jobs.push_back(MsmJob{q_of[lvl], false, level_bases(lvl), half});
    }
  std::vector<HG1> out(jobs.size());
  std::vector<HG1> col_comms(depth);  // commitments of the column-wise levels (index d: level num_vars - 1 - d)
  // the column-wise levels' commitments from the jobs' results (scalar multiplications on the host's threads, ~0.15 ms):
  // with the results committed ahead this runs WHILE the device works on the plain levels' MSM
  const std::function<void()> combine_columns = [&] {
  for (size_t d = depth; d-- > 0;) {
    const ColLevel& cl = plan->levels[d];
    // weights of the settings of the top d index bits (bit n-1-j of the index is bit d-1-j of sidx)
    std::vector<HFr> w_s((size_t)1 << d, HFr::one());
    for (size_t sidx = 0; sidx < w_s.size(); sidx++)
      for (size_t j = 0; j < d; j++) {
        const HFr& xj = point[num_vars - 1 - j];
        w_s[sidx] *= ((sidx >> (d - 1 - j)) & 1) ? xj : HFr::one() - xj;
      }
    // commitment = sum_t scale_t * result_t - offset_total * base sum  (scalar multiplications on the host's threads)
    std::vector<HG1> pts(cl.terms.size() + 1);
    std::vector<HFr> scal(cl.terms.size() + 1);
    for (size_t t = 0; t < cl.terms.size(); t++) {
      const ColTerm& tm = cl.terms[t];
      if (tm.second) memcpy(&pts[t], plan->jobs[tm.job].out_second, sizeof(HG1));
      else pts[t] = pre_out ? (*pre_out)[tm.job] : out[col_base + tm.job];
      const HFr co = small->coef[tm.k] * w_s[tm.sidx];
      scal[t] = tm.factor == -1 ? HFr::zero() - co : tm.factor == 1 ? co : co * HFr::from_u64((uint64_t)tm.factor);
    }
    HFr offset_total = HFr::zero();
    for (const ColOffset& o : cl.offsets) offset_total += small->coef[o.k] * w_s[o.sidx] * HFr::from_u64(o.off);
    pts[cl.terms.size()] = cl.need_sum ? cl.base_sum : HG1{host::Fq::zero(), host::Fq::zero()};
    scal[cl.terms.size()] = HFr::zero() - offset_total;
    std::vector<host::G1Xyzz> parts(pts.size(), host::G1Xyzz::identity());
    host_parallel_for(pts.size(), [&](size_t t) {
      if (!pts[t].is_identity() && !scal[t].is_zero()) parts[t] = host::g1_mul(host::g1_from_affine(pts[t]), scal[t]);
    });
    host::G1Xyzz acc = host::G1Xyzz::identity();
    for (const host::G1Xyzz& pt : parts) acc = host::g1_add(acc, pt);
    col_comms[d] = host::g1_to_affine(acc);
  }
  };
  const bool ahead = plan && pre_out && depth > 0;
  // the remainder (the opened value) is final before the MSM starts: its copy to the host is queued in front of the batch
  // and read behind it (fourth cache line of the ctx's pinned flag block) - not a synchronising download after it
  Fr* rem_host = (Fr*)((char*)c.flag + 192);
  LH_HIP(hipMemcpyAsync(rem_host, rem, sizeof(Fr), hipMemcpyDeviceToHost, c.stream));
  msm_batch(c, jobs.data(), jobs.size(), (G1Affine*)out.data(), ahead ? &combine_columns : nullptr);
  std::vector<HG1> comms(out.begin(), out.begin() + plain);
  if (plan && !pre_out) column_sums_store(srs, sharded, own, out.data() + col_base);
  if (!ahead) combine_columns();
  for (size_t d = depth; d-- > 0;) {  // levels in ascending order after the plain ones
    comms.push_back(col_comms[d]);
    if (self_check && d >= check_from && memcmp(&comms.back(), &out[check_base + d - check_from], sizeof(HG1)) != 0)
      fprintf(stderr, "[open] column-wise commitment of level %zu (depth %zu of %zu) differs from the plain one\n",
              plan->levels[d].level, d, depth);
  }
  c.host_stamp("open:columns");
  if (jobs.empty()) c.sync();  // (no batch ran: nothing waited for the stream yet)
  memcpy(&remainder, rem_host, sizeof(Fr));
  // what every rank holds is the commitment of its part of each quotient - its shard of a sharded level (column-wise
  // levels: of its share of the columns, offset term included - everything above is linear in the bases), its range of a
  // replicated one -> their sums, one exchange
  if (sharded) comm_sum_points(c, comms.data(), num_vars);
  c.host_stamp("open:summed");
  tr.write_commitments(comms);  // identity -> Error::Transcript (transcript.rs:172-179,216-219)
  c.host_stamp("open:written");
  return remainder;
}

// additive::batch_open (pcs/multilinear.rs:134-235), generic over the PCS: reduce to ONE opening of g' at the
// sum-check challenges and hand it to `open`
void additive_batch_open(Ctx& c, size_t num_vars, const Fr* const* d_polys, size_t num_polys, const HFr* points,
                         size_t num_points, const lh_evaluation* evals, size_t num_evals, Transcript& tr,
                         const std::function<void(const Fr* g_prime, const HFr* point)>& open, const SmallPoly* small,
                         const std::function<void(const Fr* g_prime, const HFr* point, const SmallOpen&)>& open_small) {
  LH_REQUIRE(num_vars >= 1, LH_ERR_ARG, "batch open: num_vars == 0");
  LH_REQUIRE(num_evals >= 2, LH_ERR_ARG,
             "batch open needs >= 2 evaluations (eq_xy of an empty point is the zero poly, multilinear.rs:92-94)");
  LH_REQUIRE(2 * num_points <= (size_t)SC_MAX_TABLES && num_points <= LH_SC_MAX_TERMS, LH_ERR_ARG,
             "batch open: too many points");
  for (size_t i = 0; i < num_evals; i++)
    LH_REQUIRE(evals[i].poly < num_polys && evals[i].point < num_points, LH_ERR_ARG, "batch open: bad evaluation");

  size_t ell = 0;
  while (((size_t)1 << ell) < num_evals) ell++;  // next_power_of_two().ilog2()
  std::vector<HFr> t = tr.squeeze_challenges(ell);
  std::vector<HFr> eq_xt = host_eq_xy(t);

  ArenaScope scope(c.arena);
  // inside a sharded proof (dev.hpp Shard) every poly is this rank's shard: the merges and g' are entry-wise, the
  // sum-check and the opening know about shards
  const Shard sh(c);
  const bool sharded = sh.on;
  if (sharded) LH_REQUIRE(sh.sharded(num_vars), LH_ERR_ARG, "batch open: too few variables for this shard geometry");
  const size_t n = (size_t)1 << (num_vars - (sharded ? sh.rho : 0));
  // merged_j = sum_{i : point(i) = j} eq_xt[i] * poly_i  (:155-170; the lazy first scalar there is a
  // representation detail, every field value below is the same)
  std::vector<const Fr*> merged(num_points);
  for (size_t j = 0; j < num_points; j++) {
    std::vector<const Fr*> src;
    std::vector<Fr> w, wsm;
    std::vector<const uint32_t*> sm;
    std::vector<size_t> sm_len;
    for (size_t i = 0; i < num_evals; i++)
      if (evals[i].point == j) {
        const size_t pi = evals[i].poly;
        if (small && small[pi].ptr) {  // a small-valued column: 8 multiply-adds per term, 4 bytes read instead of 32
          sm.push_back(small[pi].ptr);
          sm_len.push_back(std::min(small[pi].len, n));
          wsm.push_back(dev(eq_xt[i]));
        } else {
          src.push_back(d_polys[pi]);
          w.push_back(dev(eq_xt[i]));
        }
      }
    LH_REQUIRE(!src.empty() || !sm.empty(), LH_ERR_ARG, "batch open: a point without evaluations");
    Fr* m = c.arena.alloc_n<Fr>(n);
    if (sm.empty()) k_lincomb(c, src.data(), w.data(), src.size(), n, m);
    else k_lincomb_mixed(c, src.data(), w.data(), src.size(), sm.data(), sm_len.data(), wsm.data(), sm.size(), n, m);
    merged[j] = m;
  }
  lh_sop expr;
  memset(&expr, 0, sizeof(expr));
  expr.global_eq = -1;
  expr.num_terms = (uint32_t)num_points;
  const HFr one = HFr::one();
  for (size_t j = 0; j < num_points; j++) {
    memcpy(&expr.coeff[j], &one, 32);
    expr.num_factors[j] = 2;
    expr.factor[j][0] = (uint8_t)(num_points + j);  // eq_xy(j)
    expr.factor[j][1] = (uint8_t)j;                 // merged_j
  }
  HFr tilde_gs_sum = HFr::zero();
  for (size_t i = 0; i < num_evals; i++) {
    HFr v;
    memcpy(&v, &evals[i].value, 32);
    tilde_gs_sum += v * eq_xt[i];
  }
  SumCheckResult sc = sum_check_prove(c, LH_SC_COEFFICIENTS, num_vars, expr, merged.data(), num_points, points,
                                      num_points, tilde_gs_sum, tr, false, nullptr, sharded);
  // g' = sum_j eq_xy_eval(challenges, z_j) * merged_j  (:200-213)
  std::vector<Fr> w(num_points);
  for (size_t j = 0; j < num_points; j++)
    w[j] = dev(host_eq_xy_eval(sc.challenges.data(), points + j * num_vars, num_vars));
  // every opened poly a small-valued column: g' = sum_p coef_p col_p with coef_p = sum_{i: poly(i) = p} eq_xt[i] w[point(i)]
  if (open_small != nullptr && small != nullptr) {
    std::vector<HFr> coef(num_polys, HFr::zero());
    for (size_t i = 0; i < num_evals; i++) coef[evals[i].poly] += eq_xt[i] * hst(w[evals[i].point]);
    SmallOpen so;
    if (small_open_columns(small, num_polys, evals, num_evals, n, coef.data(), so)) {
      so.merged = merged;
      so.merged_w = w;
      open_small(nullptr, sc.challenges.data(), so);  // (g' is formed by the opening if it needs it)
      return;
    }
  }
  Fr* g_prime = c.arena.alloc_n<Fr>(n);
  k_lincomb(c, merged.data(), w.data(), num_points, n, g_prime);
  open(g_prime, sc.challenges.data());
}

void mkzg_batch_open(Ctx& c, const Srs& srs, size_t num_vars, const Fr* const* d_polys, size_t num_polys,
                     const HFr* points, size_t num_points, const lh_evaluation* evals, size_t num_evals,
                     Transcript& tr, const SmallPoly* small) {
  check_commit_vars(srs, num_vars, "batch open");
  additive_batch_open(
      c, num_vars, d_polys, num_polys, points, num_points, evals, num_evals, tr,
      [&](const Fr* g_prime, const HFr* point) { mkzg_open(c, srs, g_prime, num_vars, point, tr); }, small,
      [&](const Fr* g_prime, const HFr* point, const SmallOpen& so) { mkzg_open(c, srs, g_prime, num_vars, point, tr, &so); });
}

}  // namespace lh
