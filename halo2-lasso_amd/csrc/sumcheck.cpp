// piop::sum_check on the device: the round loop of ClassicSumCheck::prove (reference piop/sum_check/classic.rs:208-240), its
// sum-of-products front end with eq factoring, product-pair and tree-pair rounds, and the small host helpers every prover
// file shares (eq tables, interpolation, the communicator's sums).  Every function cites the reference routine whose
// transcript schedule it reproduces; all heavy loops run in HIP kernels (dev.hpp), the host keeps the Fiat-Shamir state.
#include <algorithm>
#include <functional>
#include <chrono>
#include <memory>
#include <thread>
#include "host.hpp"
#include "resident_host.hpp"

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
  bool resident_ok = true;  // cleared when the resident kernel's tail mode was tried and not taken
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
      if (ef_on && tail_ok && ef->resident_tail && !ef->per_term && !ef->untrusted && full >= 4 && full <= ((size_t)GKR_CAP * GKR_CAP)) {
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
    if (ef_on && !sh && tail_ok && resident_ok && ef->resident_tail) {
      // the factored rounds go on INSIDE the resident kernel once the tables fit it (kernels_gkr.hip tail mode): no eq
      // table is materialised, no round is launched any more.  (Not in round 0 of a claim that has yet to be checked.)
      const size_t n0 = bind ? len >> 1 : len;
      if (n0 >= 2 && n0 <= ((size_t)GKR_CAP * GKR_CAP) && (round > 0 || ef->trusted_claim) && !ef->untrusted) {
        resolve_claim();
        if (ef->resident_tail(cur, bind, r_prev, n0, round, claim, tr, res)) return res;
        resident_ok = false;  // (the kernel did not start, or a coefficient is zero: launched rounds and the generic tail)
      }
    }
    const bool tail_now = !sh && tail_ok && tail_cap && (bind ? len >> 1 : len) <= tail_cap;
    // A sum-check whose tail can run in the resident kernel stays factored until its tables fit it (sharded: until its
    // exchange): the one or two rounds before that run the factored kernels below their best size rather than lose the
    // factoring - and with it the resident tail - to materialised eq tables.  (Layers of four trees at 2^20 lookups leave the
    // streaming size at 2^15 entries per table, one round short of the resident kernel's 2^14: every one of them used to
    // fall back to eq tables, launched small rounds and the generic tail.)
    const bool keep_factored = tail_ok && resident_ok && ef_on && ef->resident_tail && !ef->per_term && !ef->untrusted &&
                               (sh || (bind ? len >> 1 : len) > ((size_t)GKR_CAP * GKR_CAP));
    if (ef_on && !keep_factored && (tail_now || !ef->streams(bind, bind ? len >> 2 : len >> 1))) {
      // the rounds leave the streaming kernel: materialise every factored eq table in the form the standard path
      // expects (the tables of the previous round, pending their bind with r_prev): S_{round-1} * E_{round-2}; before
      // round 2 that is the whole eq table of the point (S_0 = 1)
      for (EqFactoring::One& one : ef->eqs) {
        Fr* tab = c.arena.alloc_n<Fr>(len);
        if (round >= 2) {
          LH_REQUIRE(bind, LH_ERR_ARG, "sum-check: internal round mismatch");
          k_scale(c, one.level[round - 2], dev(one.S_prev), len, tab);
        } else if (sharded) {
          eq_xy_shard(c, Shard(c), one.y, num_vars, 0, tab);
        } else {
          k_eq_xy(c, (const Fr*)one.y, num_vars, tab);
        }
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
      // (a claim found untrue in round 0 - EqFactoring::untrusted - costs that point in every round: q(1..D) is all of q)
      const bool check_claim = !ef->per_term && round == 0 && !ef->trusted_claim;
      const bool all_points = check_claim || (!ef->per_term && ef->untrusted);
      const int npoints = all_points ? degree : degree - 1;
      ef->add_const = HFr::zero();
      ef->round(cur.data(), dst.data(), dev(r_prev), bind, size, round, npoints, evals_host);
      factored_round = true;
      c.route.v[RouteStats::EF_ROUNDS]++;
      if (!ef->add_const.is_zero())
        for (int x = 0; x < npoints; x++) evals_host[x] = dev(hst(evals_host[x]) + ef->add_const);
      if (all_points) {
        EqFactoring::One& e = ef->eqs[0];
        std::vector<HFr> shifted(degree);  // t -> q(t + 1), t = 0..D-1
        for (int x = 0; x < degree; x++) shifted[x] = hst(evals_host[x]);
        const HFr q0 = interpolate_evals(shifted, HFr::zero() - HFr::one());
        const HFr y0 = e.y[0];
        // not the true sum: the factored rounds go on, with nothing taken from the claim (until round 5 every round then
        // took the standard path over an eq table built in full - and round 0 ran twice)
        if (check_claim && (HFr::one() - y0) * q0 + y0 * shifted[0] != ef->c) ef->untrusted = true;
        e.q.assign(degree, HFr::zero());
        e.q[0] = q0;
        for (int x = 1; x < degree; x++) e.q[x] = shifted[x - 1];
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
        if ((round > 0 || ef->trusted_claim) && !ef->untrusted) {
          e.q.assign(degree, HFr::zero());  // q has degree D - 1: D values q(0..D-1)
          for (int x = 1; x < degree; x++) e.q[x] = hst(evals_host[x - 1]);
          e.q[0] = (ef->c - yj * e.q[1]) * ef->inv_1my[round];
        }
        for (int x = 1; x <= degree; x++) {
          const HFr fx = HFr::from_u64((uint64_t)x);
          const HFr qx = x < degree ? e.q[x] : interpolate_evals(e.q, fx);
          std_sums[x - 1] = dev(ef->kappa * e.S * eq_at(yj, fx) * qx + ef->lin0 + fx * (ef->lin1 - ef->lin0));
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
  const Ctx::ScU32 u32 = c.sc_u32;  // (dev.hpp: poly 0 still is a 32-bit column; decided below who turns it into field elements)
  c.sc_u32 = Ctx::ScU32();
  Ctx::ScU32Terms u32t;  // (dev.hpp: the polys still are combinations of 32-bit columns)
  std::swap(u32t, c.sc_u32_terms);
  bool u32t_rounds = false, u32t_bound = false;
  bool u32_rounds = false;
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
  // product-pair shape (dev.hpp ScRound::pp): every term c_m l_m r_m over 2 num_terms distinct tables, non-zero
  // coefficients - the generic layers of the grand products
  bool pp_terms = !rw && rd.num_terms >= 2 && c.opt.sc_pp_fold != 0;
  {
    std::vector<char> seen(T, 0);
    for (uint32_t m = 0; m < rd.num_terms && pp_terms; m++) {
      HFr co;
      memcpy(&co, &rd.coeff[m], 32);
      pp_terms = rd.nfac[m] == 2 && !co.is_zero();
      for (int k = 0; k < 2 && pp_terms; k++) {
        pp_terms = rd.fac[m][k] < num_polys && !seen[rd.fac[m][k]];
        seen[rd.fac[m][k]] = 1;
      }
    }
  }
  // ... whose tail the resident kernel takes (kernels_gkr.hip tail mode): then the factoring pays even when the first
  // rounds are below the streaming size - the tail runs factored inside the kernel, no eq table is ever built
  const uint32_t tail_B = rw ? rw->num_pairs : rd.num_terms;
  const bool tail_shape = c.opt.gkr_resident && c.opt.sc_tail && (pp_terms || rw) && num_polys == 2 * (size_t)tail_B &&
                          tail_B <= (uint32_t)GKR_MAX_TREES && rd.global_eq >= 0 && prover_kind == LH_SC_EVALUATIONS &&
                          degree >= 2 && sum_is_exact;
  const bool streams2 = nvl >= 3 && k_sc_round_streams(rd, degree, (size_t)1 << (nvl - 1)) &&
                        k_sc_round_streams(rd, degree, (size_t)1 << (nvl - 2));
  if (ef_enabled && nvl >= 3 && (!sharded || j >= 2) && (streams2 || tail_shape)) {
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
  // Its factored rounds run sc_round_pp_kernel, and the first of them that binds folds the coefficients into the left
  // factors: from then on every kernel of this sum-check (streaming, LDS-staged, resident tail) sees coefficients of one,
  // and the final evaluations of the left factors are divided by c_m at the end.
  // The first three rounds straight from a 32-bit column (Surge over the output column): one table, coefficient one, a
  // claim that is the true sum, one GPU, and a column long enough for round 2 to be a launched streaming round.  The sums of
  // rounds 0 and 1 need no challenge (k_inner_products_small_quads: the caller's, or made here), round 2 binds r0 and r1 from
  // the column (k_sc_round_u32_bind2); from round 3 on the table is field elements.  Round 1 binds nothing: the loop's
  // table pointer of that round names memory nobody reads.
  Ctx::ScU32 u32s = u32;
  Fr u32_r0;
  bool u32_bound = false;
  if (u32.col) {
    static const bool u32_off = getenv("LH_SC_U32") && atoi(getenv("LH_SC_U32")) == 0;  // (development A/B)
    u32_rounds = !u32_off && use_ef && !ef.per_term && !sharded && !rw && num_polys == 1 && rd.num_terms == 1 && rd.nfac[0] == 1 &&
                 rd.fac[0][0] == 0 && rd.coeff_is_one[0] && degree == 2 && sum_is_exact && streams2 && num_vars >= 6 &&
                 k_sc_round_streams(rd, degree, len0 >> 3) &&
                 (len0 >> 2) > std::max<size_t>(k_sc_tail_capacity(c, rd, degree), (size_t)GKR_CAP * GKR_CAP);
    if (!u32_rounds) k_fr_from_u32(c, u32.col, len0, const_cast<Fr*>(d_polys[0]));
  }
  // The batch opening's polys as combinations of 32-bit columns: rounds 0 and 1 from the columns' quad sums against E_1 of each
  // term (no challenge needed), round 2 binds r0 and r1 from the columns (k_lincomb_bind2).  Same conditions as above; when they
  // do not hold the tables are filled here the way the caller would have (k_lincomb_mixed).
  std::vector<Fr> u32t_sums;  // [term][4]: sum over the term's columns of w_k S_t(col_k)
  Fr u32t_r0;
  if (!u32t.polys.empty()) {
    static const bool u32t_off = getenv("LH_OPEN_U32_ROUNDS") && atoi(getenv("LH_OPEN_U32_ROUNDS")) == 0;  // (development A/B)
    LH_REQUIRE(u32t.polys.size() == num_polys, LH_ERR_ARG, "sum-check: the column hint does not match the polys");
    // (from 2^22 entries on: below that the ~11 quad-sum launches of a proof are latency - 2^20 AND 11.35 -> 12.1 ms, 2^20 range
    //  +0.15, 2^21 even, 2^22 range -0.25 ms; LH_OPEN_U32_MIN_VARS: development A/B)
    static const size_t u32t_min_vars = getenv("LH_OPEN_U32_MIN_VARS") ? (size_t)atoi(getenv("LH_OPEN_U32_MIN_VARS")) : 22;
    bool ok = !u32t_off && use_ef && ef.per_term && !sharded && degree == 2 && streams2 && num_vars >= u32t_min_vars &&
              k_sc_round_streams(rd, degree, len0 >> 3) &&
              (len0 >> 2) > std::max<size_t>(k_sc_tail_capacity(c, rd, degree), (size_t)GKR_CAP * GKR_CAP);
    for (const Ctx::ScU32Terms::Poly& pl : u32t.polys) {
      ok = ok && !pl.col.empty() && pl.col.size() <= 24;
      for (size_t ln : pl.len) ok = ok && ln % 4 == 0;
    }
    u32t_rounds = ok;
    if (!ok)
      for (size_t b = 0; b < num_polys; b++) {
        const Ctx::ScU32Terms::Poly& pl = u32t.polys[b];
        k_lincomb_mixed(c, nullptr, nullptr, 0, pl.col.data(), pl.len.data(), pl.w.data(), pl.col.size(), len0, const_cast<Fr*>(d_polys[b]));
      }
    c.sc_u32_terms.built = !ok;
  }
  const bool pp_shape = use_ef && !ef.per_term && pp_terms;
  std::vector<HFr> pp_folded;  // the coefficients that went into the left factors (empty: not folded)
  bool rw_folded = false;      // tree-pair rounds (ScRwPairs): the tables hold l' = cs (l + k), r' = r + k since the first bind
  if (use_ef) {
    const Shard shg(c);
    const size_t half = (size_t)1 << (nvl - 1);
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
      k_eq_levels(c, one.level[0], half, lower.data(), lower.size());
    }
    ef.streams = [&](bool, size_t size) { return k_sc_round_streams(rd, degree, size); };
    ef.round = [&](const Fr* const* in, Fr* const* out, const Fr& r, bool bind, size_t size, size_t round, int points,
                   Fr* out_host) {
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
        const bool fold = bind && c.opt.sc_pp_fold == 1;  // the first binding round folds cs and k into the tables
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
      } else if (!ef.per_term && u32_rounds && round <= 2) {
        LH_REQUIRE(points == 1 && bind == (round >= 1), LH_ERR_ARG, "sum-check: the 32-bit rounds met another shape");
        if (round == 0) {
          if (!u32s.have_sums) {
            Fr s4[4];
            k_inner_products_small_quads(c, u32.col, ef.eqs[0].level[0], size >> 1, s4);
            u32s.odd = s4[1], u32s.s2 = s4[2], u32s.s3 = s4[3], u32s.have_sums = true;
          }
          out_host[0] = u32s.odd;
        } else if (round == 1) {  // q(1) = (1 - r0) S2 + r0 S3; the bind waits for r1
          u32_r0 = r;
          const HFr r0 = hst(r);
          out_host[0] = dev((HFr::one() - r0) * hst(u32s.s2) + r0 * hst(u32s.s3));
        } else {
          k_sc_round_u32_bind2(c, u32.col, ef.eqs[0].level[2], u32_r0, r, size, out[0], out_host);
          u32_bound = true;
        }
      } else if (!ef.per_term) {
        ScRound g = rd;
        for (size_t i = 0; i < T; i++) g.in[i] = in[i], g.out[i] = out[i];
        g.r = r;
        g.global_eq = -1;
        g.eq_level = ef.eqs[0].level[round];
        g.pp = pp_shape && points == 2 ? 1 : 0;
        if (g.pp && bind && pp_folded.empty() && c.opt.sc_pp_fold == 1) g.pp = 2;  // this round stores l'_m = c_m l_m
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
      } else if (u32t_rounds && round <= 2) {
        LH_REQUIRE(bind == (round >= 1), LH_ERR_ARG, "sum-check: the column rounds met another shape");
        const uint32_t M = rd.num_terms;
        if (round == 0) {
          // the quad sums of every column against its term's E_1, one download; then per term the weighted sums
          ArenaScope scope(c.arena);
          size_t total = 0;
          for (uint32_t m = 0; m < M; m++) total += u32t.polys[term_poly[m]].col.size();
          Fr* d_s = c.arena.alloc_n<Fr>(4 * total);
          size_t off = 0;
          for (uint32_t m = 0; m < M; m++) {
            const Ctx::ScU32Terms::Poly& pl = u32t.polys[term_poly[m]];
            k_inner_products_quads(c, pl.col.data(), pl.len.data(), pl.col.size(), ef.eqs[m].level[1], size >> 1, d_s + 4 * off);
            off += pl.col.size();
          }
          std::vector<Fr> hs(4 * total);
          c.d2h(hs.data(), d_s, hs.size() * sizeof(Fr));
          u32t_sums.assign(4 * (size_t)M, dev(HFr::zero()));
          off = 0;
          for (uint32_t m = 0; m < M; m++) {
            const Ctx::ScU32Terms::Poly& pl = u32t.polys[term_poly[m]];
            HFr t4[4] = {HFr::zero(), HFr::zero(), HFr::zero(), HFr::zero()};
            for (size_t k = 0; k < pl.col.size(); k++)
              for (int t = 0; t < 4; t++) t4[t] += hst(pl.w[k]) * hst(hs[4 * (off + k) + t]);
            off += pl.col.size();
            for (int t = 0; t < 4; t++) u32t_sums[4 * m + t] = dev(t4[t]);
            // round 0 sums over E_0[2q + e] = eq(y_1, e) E_1[q]: even entries (1 - y1) S0 + y1 S2, odd ones (1 - y1) S1 + y1 S3
            const HFr y1 = ef.eqs[m].y[1], n1 = HFr::one() - y1;
            out_host[2 * m] = dev(n1 * t4[0] + y1 * t4[2]);
            out_host[2 * m + 1] = dev(n1 * t4[1] + y1 * t4[3]);
          }
        } else if (round == 1) {  // the table bound with r0, against E_1: entries 2q are (1 - r0) m[4q] + r0 m[4q+1]
          u32t_r0 = r;
          const HFr r0 = hst(r), n0 = HFr::one() - r0;
          for (uint32_t m = 0; m < M; m++) {
            const Fr* t4 = &u32t_sums[4 * m];
            out_host[2 * m] = dev(n0 * hst(t4[0]) + r0 * hst(t4[1]));
            out_host[2 * m + 1] = dev(n0 * hst(t4[2]) + r0 * hst(t4[3]));
          }
        } else {
          Fr* two = (Fr*)c.pin(65536) + 1024;  // (behind the argument block of k_lincomb_bind2; a launch per term)
          for (uint32_t m = 0; m < M; m++) {
            const Ctx::ScU32Terms::Poly& pl = u32t.polys[term_poly[m]];
            k_lincomb_bind2(c, pl.col.data(), pl.len.data(), pl.w.data(), pl.col.size(), u32t_r0, r, ef.eqs[m].level[2], size,
                            out[term_poly[m]], two);
            out_host[2 * m] = two[0], out_host[2 * m + 1] = two[1];
          }
          u32t_bound = true;
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
  LH_REQUIRE(!u32_rounds || u32_bound, LH_ERR_DEVICE, "sum-check: the 32-bit column was never bound");
  LH_REQUIRE(!u32t_rounds || u32t_bound, LH_ERR_DEVICE, "sum-check: the 32-bit columns were never bound");
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


}  // namespace lh
