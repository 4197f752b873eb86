// Host half of the prover: mirrors the reference's trait surface for the hot path
// (piop::sum_check, piop::gkr, pcs::multilinear::kzg) on top of the device kernels.
#pragma once
#include <string.h>
#include <functional>
#include <map>
#include <mutex>
#include <vector>
#include "dev.hpp"
#include "ff_host.hpp"

namespace lh {

const char* get_last_error();

typedef host::Fr HFr;
typedef host::G1Affine HG1;

static inline Fr dev(const HFr& f) {
  Fr r;
  memcpy(&r, &f, 32);
  return r;
}
static inline HFr hst(const Fr& f) {
  HFr r;
  memcpy(&r, &f, 32);
  return r;
}

// ------------------------------------------------------------------ hash + transcript
struct Keccak256 {
  static constexpr int RATE = 136;
  uint64_t state[25] = {0};
  uint8_t buf[RATE];
  size_t buf_len = 0;
  void absorb_block(const uint8_t* block);
  void update(const uint8_t* data, size_t len);
  void finalize_reset(uint8_t out[32]);
};

struct KeccakTranscript {
  lh_transcript vt;  // must stay the first member: lh_transcript* <-> KeccakTranscript*
  Keccak256 hash;
  std::vector<uint8_t> stream;
  size_t pos = 0;  // read cursor (from_proof transcripts)
  KeccakTranscript();
};

// typed view over the callback table (util/transcript.rs:15-97)
struct Transcript {
  lh_transcript* t;
  explicit Transcript(lh_transcript* t_) : t(t_) {
    LH_REQUIRE(t && t->write_field_element && t->common_field_element && t->squeeze_challenge &&
                   t->write_commitment && t->common_commitment,
               LH_ERR_ARG, "transcript callback table is incomplete");
  }
  void check(int rc) {
    if (rc != LH_OK) throw Error(rc, get_last_error()[0] ? get_last_error() : "transcript callback failed");
  }
  void write_field_element(const HFr& f) { check(t->write_field_element(t->user, (const lh_fr*)&f)); }
  void write_field_elements(const std::vector<HFr>& fs) {
    for (auto& f : fs) write_field_element(f);
  }
  void common_field_element(const HFr& f) { check(t->common_field_element(t->user, (const lh_fr*)&f)); }
  HFr squeeze_challenge() {
    HFr f;
    check(t->squeeze_challenge(t->user, (lh_fr*)&f));
    return f;
  }
  std::vector<HFr> squeeze_challenges(size_t n) {
    std::vector<HFr> v(n);
    for (auto& f : v) f = squeeze_challenge();
    return v;
  }
  void write_commitment(const HG1& p) { check(t->write_commitment(t->user, (const lh_g1*)&p)); }
  void write_commitments(const std::vector<HG1>& ps) {
    for (auto& p : ps) write_commitment(p);
  }
  void common_commitment(const HG1& p) { check(t->common_commitment(t->user, (const lh_g1*)&p)); }
  // TranscriptRead (util/transcript.rs:45-97)
  HFr read_field_element() {
    LH_REQUIRE(t->read_field_element, LH_ERR_ARG, "transcript cannot read (write-only callback table)");
    HFr f;
    check(t->read_field_element(t->user, (lh_fr*)&f));
    return f;
  }
  std::vector<HFr> read_field_elements(size_t n) {
    std::vector<HFr> v(n);
    for (auto& f : v) f = read_field_element();
    return v;
  }
  HG1 read_commitment() {
    LH_REQUIRE(t->read_commitment, LH_ERR_ARG, "transcript cannot read (write-only callback table)");
    HG1 p;
    check(t->read_commitment(t->user, (lh_g1*)&p));
    return p;
  }
  std::vector<HG1> read_commitments(size_t n) {
    std::vector<HG1> v(n);
    for (auto& p : v) p = read_commitment();
    return v;
  }
};

// PolynomialCommitmentScheme::batch_verify of the PCS a backend is generic over
typedef std::function<void(size_t num_vars, const HG1* comms, size_t num_comms, const HFr* points, size_t num_points,
                           const lh_evaluation* evals, size_t num_evals, Transcript& tr)>
    PcsBatchVerify;

// ------------------------------------------------------------------ poly helpers (host side, tiny inputs)
std::vector<HFr> host_eq_xy(const std::vector<HFr>& y);           // multilinear.rs:91-127
HFr host_eq_xy_eval(const HFr* x, const HFr* y, size_t n);         // sum_check.rs:112-121
// evaluations of `count` device tables at `point` (multilinear.rs:137-156)
std::vector<HFr> evaluate_polys(Ctx&, const Fr* const* d_polys, size_t count, size_t num_vars, const HFr* point);

// ------------------------------------------------------------------ piop::sum_check
struct SumCheckResult {
  std::vector<HFr> challenges;  // x
  std::vector<HFr> evals;       // every poly at x (classic.rs:143-149)
};
// `sum_is_exact`: the caller computed `sum` from these very tables (the provers' internal sum-checks: Surge, the GKR
// layers): eq factoring then trusts it from round 0 on; a claim from outside (the C-ABI) is checked first (EqFactoring)
// `rw` (optional): the expression is a grand-product layer over (A_i, A_i + 1) tree pairs written over the A tables
// alone (dev.hpp ScRwRound; polys = l_0, r_0, l_1, r_1, ...): its streaming rounds run the dedicated kernel
struct ScRwPairs {
  uint32_t num_pairs;
  HFr cs[SC_RW_MAX_PAIRS], k[SC_RW_MAX_PAIRS];
  HFr const_total;  // sum of the constants the factorisation leaves behind: added to q at every point
};
SumCheckResult sum_check_prove(Ctx&, int prover_kind, size_t num_vars, const lh_sop& expr, const Fr* const* d_polys,
                               size_t num_polys, const HFr* ys, size_t num_ys, const HFr& sum, Transcript& tr,
                               bool sum_is_exact = false, const ScRwPairs* rw = nullptr, bool sharded = false);
// `sharded` (inside a sharded proof, dev.hpp Shard): d_polys are this rank's shards of num_vars-variable tables; same
// messages, same result on every rank

// eq table of y[1..num_vars) (2^(num_vars-1) entries), shared through Ctx::eq_half_cache within one proof; `sharded`: this
// rank's shard of it with the rank's factor of the shard coordinates multiplied in
const Fr* eq_half_lookup(Ctx&, const HFr* y, size_t num_vars, bool sharded = false);
const Fr* eq_half_get(Ctx&, const HFr* y, size_t num_vars, bool sharded = false);  // built (in the arena, at the caller's depth) when absent
struct EqHalfScope {  // forgets, on exit, what was cached after its creation
  Ctx& c;
  size_t mark;
  explicit EqHalfScope(Ctx& c_) : c(c_), mark(c_.eq_half_cache.size()) {}
  ~EqHalfScope() { c.eq_half_cache.resize(mark); }
};

// the round loop shared by every sum-check front end (sumcheck.cpp)
typedef std::function<void(const Fr* const*, Fr* const*, const Fr&, bool, size_t, Fr*)> RoundFn;
// Eq factoring of the streaming rounds.  eq(y, x) = prod_i eq(y_i, x_i), so in round j (prefix bound to rho):
//   eq(y, (rho, X, b)) = S_j * eq(y_j, X) * E_j[b],   S_j = prod_{i<j} eq(y_i, rho_i),   E_j = eq table over variables > j.
// The round polynomial of  eq * g  is therefore  p(X) = S_j eq(y_j, X) q(X)  with  q(X) = sum_b E_j[b] g(rho, X, b)  of one
// degree less: the device evaluates q at one point fewer, reads ONE eq entry per pair instead of streaming and binding a
// whole eq table, and the host rebuilds exactly the reference's message p(0..D) (field arithmetic is exact: same bytes).
// q(0) follows from the claim:  claim / S_j = (1 - y_j) q(0) + y_j q(1).  When the rounds leave the streaming kernel the eq
// table is materialised once in its "previous round" form  S_{j-1} * E_{j-2}  and the standard path continues.
struct EqFactoring {
  struct One {
    size_t table;                    // index of the eq table in `cur` (filled at the switch)
    const HFr* y;                    // the eq point (num_vars coordinates)
    std::vector<const Fr*> level;    // level[j] = E_j, 2^(num_vars - 1 - j) entries on the device
    HFr S, S_prev;                   // S_j, S_{j-1}
    std::vector<HFr> q;              // q(0..) of the current round
  };
  std::vector<One> eqs;    // global-eq shape: one entry; per-term shape (sum_m eq_m * poly_m): one per term, in term order
  bool per_term = false;
  bool trusted_claim = false;  // the claim is known to be the true sum: no check (and no extra point) in round 0
  // set by the round loop when round 0 found the claim NOT to be the true sum (the reference still sends the true p(1..D),
  // eval.rs:129 - benches/zero_check.rs proves a false claim over random tables): every round then evaluates q at D points
  // and takes nothing from the claim; the eq table stays factored
  bool untrusted = false;
  std::vector<HFr> inv_1my;  // global-eq shape: (1 - y_j)^-1 for every round
  HFr c;                     // global-eq shape: claim / S_j
  HFr add_const;             // set by `round`: a constant the kernel left out of every q value (ScRwPairs::const_total)
  // global-eq shape beside a LINEAR part (the zero-check of HyperPlonk, expr.cpp): the round polynomial is
  //   p(X) = A(X) + kappa S_j eq(y_j, X) q(X),  A(X) = lin0 + X (lin1 - lin0)
  // with lin0 / lin1 = the linear part's sums over the even / odd entries of this round's tables, set by `round`;
  // `c` then is the eq part's claim alone (round 0: set by `round` too, (sum - lin0 - lin1) / kappa)
  HFr kappa = HFr::one(), lin0 = HFr::zero(), lin1 = HFr::zero();
  // would this round run the streaming kernel (else the eq tables are materialised and the standard path takes over)
  std::function<bool(bool bind, size_t size)> streams;
  // launches the factored round; device output: q(1..points) (global-eq shape; points = D - 1, or D in round 0 where the
  // claim is checked rather than trusted) or q_m(0), q_m(1) per term
  std::function<void(const Fr* const* in, Fr* const* out, const Fr& r, bool bind, size_t size, size_t round, int points,
                     Fr* out_host)>
      round;
  // the rest of the sum-check inside the resident kernel (kernels_gkr.hip, GKR_F_* tail mode) when the shape allows it:
  // `cur` the current tables (pending their bind with r_prev when `bind`), n0 entries each after that bind, `round` the
  // first resident round, `claim` the running claim.  Appends the challenges and fills the evaluations; false: not taken.
  std::function<bool(const std::vector<const Fr*>& cur, bool bind, const HFr& r_prev, size_t n0, size_t round, const HFr& claim,
                     Transcript& tr, SumCheckResult& res)>
      resident_tail;
};
SumCheckResult sum_check_loop(Ctx&, int prover_kind, size_t num_vars, int degree, std::vector<const Fr*> cur,
                              const std::vector<char>& used, size_t num_polys, const HFr& sum, Transcript& tr,
                              bool sharded, const RoundFn& round_fn, const ScRound* tail_rd = nullptr,
                              EqFactoring* ef = nullptr);
// general Expression (util/expression.rs) through EvaluationsProver; evals = every poly at x
SumCheckResult sum_check_prove_expr(Ctx&, size_t num_vars, const lh_expr& expr, const Fr* const* d_polys,
                                    size_t num_polys, const HFr* challenges, size_t num_challenges, const HFr* ys,
                                    size_t num_ys, const HFr& sum, Transcript& tr, bool sharded = false);

// one proof over several GPUs (dev.hpp Shard): tables are this rank's shards, results are global
std::vector<HFr> evaluate_polys(Ctx&, const Fr* const* d_polys, size_t count, size_t num_vars, const HFr* point, bool sharded);
void comm_sum_fr(Ctx&, HFr* v, size_t n);
void comm_sum_points(Ctx&, HG1* pts, size_t n);
void comm_gather_tables(Ctx&, const Fr* local_block, size_t count, size_t n_local, size_t block, Fr* const* out);
void comm_gather_concat(Ctx&, const Fr* local, size_t n_local, Fr* out);
// this rank's shard of eq_xy(y[first..num_vars)) (the shard coordinates dropped, the rank's factor multiplied in)
void eq_xy_shard(Ctx&, const Shard&, const HFr* y, size_t num_vars, size_t first, Fr* out_local);

// ------------------------------------------------------------------ piop::gkr
struct FracSumCheckResult {
  std::vector<HFr> p_xs, q_xs, x;
};
FracSumCheckResult prove_fractional_sum_check(Ctx&, size_t num_batching, size_t num_vars,
                                              const HFr* const* claimed_p_0s, const HFr* const* claimed_q_0s,
                                              const Fr* const* d_ps, const Fr* const* d_qs, Transcript& tr);
struct GrandProductResult {
  std::vector<HFr> roots, claims;
  std::vector<std::vector<HFr>> points;
};
// d_level_up (optional): per tree the level above the leaves (2^(num_vars-1) nodes, node i = leaf[i] * leaf[i + half]) when
// the caller made it together with the leaves, else null
// plus_one (optional): plus_one[b] != 0 says that tree b's leaves are tree (b - 1)'s leaves + 1, entry by entry (same
// depth; Lasso's write set over its read set).  The leaf layer of such a pair then runs over tree (b - 1)'s tables
// alone (dev.hpp ScRwRound) and d_leaves[b] is never read (it may be null; d_level_up[b] must be given).
GrandProductResult prove_grand_product(Ctx&, size_t num_trees, const Fr* const* d_leaves, const size_t* num_vars,
                                       Transcript& tr, const Fr* const* d_level_up = nullptr,
                                       const uint8_t* plus_one = nullptr);

// ------------------------------------------------------------------ pcs::multilinear::kzg
struct Srs {
  G1Affine* d_eqs = nullptr;  // flat: level k at offset 2^k - 1
  size_t num_vars = 0;
  const G1Affine* eq(size_t k) const { return d_eqs + (((size_t)1 << k) - 1); }
  // this rank's share of the bases of every sharded level (sharded proving; built on first use)
  mutable std::vector<G1Affine*> shard_levels;
  mutable int shard_rank = -1;
  mutable size_t shard_R = 0, shard_j = 0;
  // sum of all bases of a level (mkzg_open over small-valued columns), computed on first use
  mutable std::map<size_t, HG1> level_sums;  // (guarded by one process-wide mutex, open_columns.hpp srs_cache_mu)
  mutable std::map<size_t, HG1> shard_level_sums;  // the same over this rank's share of a sharded level
  // window tables of whole levels (dev.hpp MsmJob::win_table; Options::msm_window_tables), built on first use
  struct WinTable {
    G1Affine* d = nullptr;
    uint32_t c = 0, W = 0;
  };
  mutable std::map<size_t, WinTable> win_tables;
};
// the window table of a level (nullptr: tables are off for this level - Options::msm_window_tables - or do not fit)
const Srs::WinTable* srs_window_table(Ctx&, const Srs&, size_t level);
const G1Affine* srs_shard_level(Ctx&, const Srs&, size_t level);  // this rank's share of a level's bases (sharded proofs)
Srs* mkzg_setup(Ctx&, const HFr* ss, size_t num_vars);
std::vector<HG1> mkzg_batch_commit(Ctx&, const Srs&, const Fr* const* d_polys, size_t num_polys, size_t num_vars);
std::vector<HG1> mkzg_batch_commit_u32(Ctx&, const Srs&, const uint32_t* const* d_polys, size_t num_polys,
                                       size_t num_vars);
struct SmallOpen;
// small (optional): d_poly = sum_k coef[k] * cols[k] with small-valued columns: the largest quotient is then committed
// column by column from 32-bit differences (mkzg.cpp, open_columns.cpp)
HFr mkzg_open(Ctx&, const Srs&, const Fr* d_poly, size_t num_vars, const HFr* point, Transcript& tr,
              const SmallOpen* small = nullptr);
// a poly handed over as its small-valued u32 column (`len` entries, zero beyond): Lasso's dim / read_ts / E / final_cts
// reach the batch opening without a field-element view (d_polys[i] may then be null)
struct SmallLinear {  // column = sum_k coeff[k] * (the column of poly[k]) entry by entry (poly: indices of the same opening)
  std::vector<size_t> poly;
  std::vector<HFr> coeff;
};
struct SmallPoly {
  const uint32_t* ptr = nullptr;
  size_t len = 0;
  uint32_t bits = 0;  // every entry < 2^bits when the caller knows (0: unknown)
  // optional: this column is an exact linear combination of other columns of the opening (Lasso's output a = g(E) under a
  // linear g): the small-column opening then folds its coefficient into theirs and never touches it
  const SmallLinear* linear = nullptr;
};
void mkzg_batch_open(Ctx&, const Srs&, size_t num_vars, const Fr* const* d_polys, size_t num_polys,
                     const HFr* points, size_t num_points, const lh_evaluation* evals, size_t num_evals,
                     Transcript& tr, const SmallPoly* small = nullptr);

// g' of a batch opening whose polys are ALL small-valued columns: g' = sum_k coef[k] * cols[k] (scalar coefficients)
struct SmallOpen {
  std::vector<SmallPoly> cols;
  std::vector<HFr> coef;
  // the same g' as a combination of field-element tables (the merged polys of the batch opening): when the opening is
  // handed a null g', it forms what it needs from these (the first fold directly, g' itself only on the plain route)
  std::vector<const Fr*> merged;
  std::vector<Fr> merged_w;
  // the merged tables may not have been written yet (the batch opening's sum-check ran from the columns: sumcheck.cpp,
  // Ctx::sc_u32_terms): whoever needs them calls this first (null: they are there)
  std::function<void()> ensure_merged;
};
// Options::open_precommit: commit the challenge-free half of the coming batch opening's column route on the ctx's helper
// ctx, starting now (the opening's polys, their small columns and the (poly, point) pairs as mkzg_batch_open will get them;
// the evaluations' values are not used).  mkzg_open picks the results up when they fit, and commits as before when not.
void open_precommit_start(Ctx&, const Srs&, size_t num_vars, const SmallPoly* small, size_t num_polys,
                          const lh_evaluation* evals, size_t num_evals);
// `open_small` (optional) is called instead of `open` when every opened poly came as a small-valued column
void additive_batch_open(Ctx&, size_t num_vars, const Fr* const* d_polys, size_t num_polys, const HFr* points,
                         size_t num_points, const lh_evaluation* evals, size_t num_evals, Transcript& tr,
                         const std::function<void(const Fr* g_prime, const HFr* point)>& open,
                         const SmallPoly* small = nullptr,
                         const std::function<void(const Fr* g_prime, const HFr* point, const SmallOpen&)>& open_small = nullptr);

// ------------------------------------------------------------------ pcs::multilinear::zeromorph over pcs::univariate::kzg
struct USrs {  // UnivariateKzgParam (univariate/kzg.rs:38-66): powers_of_s_g1 on the device
  G1Affine* d_powers = nullptr;
  size_t size = 0;
};
USrs* ukzg_setup(Ctx&, const HFr& s, size_t poly_size);
// poly_size: the trim size (zeromorph.rs:90-108): commits use powers[..poly_size], the final quotient powers[size - poly_size..]
std::vector<HG1> zeromorph_batch_commit(Ctx&, const USrs&, size_t poly_size, const Fr* const* d_polys, size_t num_polys,
                                        size_t num_vars);
void zeromorph_open(Ctx&, const USrs&, size_t poly_size, const Fr* d_poly, size_t num_vars, const HFr* point,
                    Transcript& tr);
void zeromorph_batch_open(Ctx&, const USrs&, size_t poly_size, size_t num_vars, const Fr* const* d_polys,
                          size_t num_polys, const HFr* points, size_t num_points, const lh_evaluation* evals,
                          size_t num_evals, Transcript& tr, const SmallPoly* small = nullptr);
// zeromorph.rs:258-296 -> (eval_scalar, q_scalars)
std::pair<HFr, std::vector<HFr>> zeromorph_scalars(const HFr& y, const HFr& x, const HFr& z, const HFr* u, size_t n);
struct ZmVerifierParams;  // ZeromorphKzgVerifierParam (zeromorph.rs:42-65)
ZmVerifierParams* zeromorph_vp_setup(const HFr& s, size_t param_size, size_t poly_size);
ZmVerifierParams* zeromorph_vp_new(const lh_g1& g1, const lh_g2& g2, const lh_g2& s_g2, const lh_g2& s_offset_g2);
void zeromorph_vp_export(const ZmVerifierParams&, lh_g1* g1, lh_g2* g2, lh_g2* s_g2, lh_g2* s_offset_g2);
void zeromorph_vp_free(ZmVerifierParams*);
void zeromorph_verify(const ZmVerifierParams&, const HG1& comm, const HFr* point, size_t num_vars, const HFr& eval,
                      Transcript& tr);
void zeromorph_batch_verify(const ZmVerifierParams&, size_t num_vars, const HG1* comms, size_t num_comms,
                            const HFr* points, size_t num_points, const lh_evaluation* evals, size_t num_evals,
                            Transcript& tr);

// ------------------------------------------------------------------ Lasso
// what the Lasso prover needs from its PCS: bases whose first 2^nv points commit a zero-padded table of 2^nv entries
// (the eq basis of level nv, or the powers of s), the largest nv they cover, and batch_open
struct LassoPcs {
  std::function<const G1Affine*(size_t nv)> commit_bases;
  std::function<const G1Affine*(size_t nv)> shard_bases;  // this rank's share of them (sharded proofs; null: unsupported)
  size_t max_vars;
  std::function<void(size_t num_vars, const Fr* const* d_polys, size_t num_polys, const HFr* points, size_t num_points,
                     const lh_evaluation* evals, size_t num_evals, Transcript& tr, const SmallPoly* small)>
      batch_open;
  // optional (multilinear KZG): start committing the challenge-free half of that batch opening now (open_precommit_start)
  std::function<void(size_t num_vars, const SmallPoly* small, size_t num_polys, const lh_evaluation* evals, size_t num_evals)>
      precommit;
};
LassoPcs lasso_mkzg_pcs(Ctx&, const Srs&);
LassoPcs lasso_zeromorph_pcs(Ctx&, const USrs&, size_t poly_size);
// pieces of the argument shared by the standalone prover (lasso.cpp) and HyperPlonk's Lasso lookups (hyperplonk.cpp)
struct LassoColumns {  // small-valued witness columns as u32 (arena memory of the caller's scope)
  std::vector<uint32_t*> rts, fcs, E;
  std::vector<uint32_t*> dim_sorted, dim_index;  // (keep_sorted) every dim column sorted by value + source positions
};
struct LassoClaims {  // points and claimed evaluations that remain to be opened
  std::vector<HFr> r, r_z, r_N, r_M;
  HFr v;                  // a(r)
  std::vector<HFr> e_rz;  // E_i(r_z)
  std::vector<HFr> ev_n;  // dim_j | read_ts_j | E_i at r_N
  std::vector<HFr> ev_l;  // final_cts_j at r_M
};
// T[d] and the d's sorted by T[d] for the bitwise subtables, on the device (arena memory of the caller's scope; the host
// staging lives as long as this object, i.e. past the msm_batch that consumes them)
struct SubtableOrders {
  Ctx& c;
  size_t l;
  std::vector<uint32_t> h_tab[3], h_ord[3];
  uint32_t *d_tab[3] = {nullptr, nullptr, nullptr}, *d_ord[3] = {nullptr, nullptr, nullptr};
  SubtableOrders(Ctx& c_, size_t l_) : c(c_), l(l_) {}
  void get(int kind, const uint32_t** table, const uint32_t** order) {
    if (!d_tab[kind]) {
      const size_t M = (size_t)1 << l, h = l / 2, V = (size_t)1 << h;
      std::vector<uint32_t>&tab = h_tab[kind], &ord = h_ord[kind];
      tab.resize(M), ord.resize(M);
      std::vector<uint32_t> start(V + 1, 0);
      for (size_t d = 0; d < M; d++) {
        const uint32_t x = (uint32_t)(d >> h), y = (uint32_t)(d & (V - 1));
        tab[d] = kind == LH_SUBTABLE_AND ? (x & y) : (x ^ y);
        start[tab[d] + 1]++;
      }
      for (size_t v = 0; v < V; v++) start[v + 1] += start[v];
      for (size_t d = 0; d < M; d++) ord[start[tab[d]]++] = (uint32_t)d;  // counting sort by T
      d_tab[kind] = c.arena.alloc_n<uint32_t>(M);
      d_ord[kind] = c.arena.alloc_n<uint32_t>(M);
      LH_HIP(hipMemcpyAsync(d_tab[kind], tab.data(), M * 4, hipMemcpyHostToDevice, c.stream));
      LH_HIP(hipMemcpyAsync(d_ord[kind], ord.data(), M * 4, hipMemcpyHostToDevice, c.stream));
    }
    *table = d_tab[kind], *order = d_ord[kind];
  }
};

void lasso_check_table(const lh_lasso_table& tb);
// a_out (optional): the output column a = g(E) as field elements; a_small_out (optional): the same as a 32-bit column
// when g is linear with small coefficients and the value fits (then *a_out stays null)
LassoColumns lasso_witness_columns(Ctx&, const lh_lasso_table&, size_t n, const uint32_t* const* d_dims, Fr** a_out,
                                   uint32_t** a_small_out = nullptr, bool keep_sorted = false);
// `a`: the output column as field elements, or null with `a_small` given
LassoClaims lasso_argue(Ctx&, const lh_lasso_table&, size_t n, const LassoColumns& w, const uint32_t* const* d_dims,
                        const Fr* a, const Fr* const* E_fr, Transcript& tr,
                        const std::function<void(int)>& lap = nullptr, const uint32_t* a_small = nullptr);
// commitment framing of the Lasso argument: identity mask as one field element, then the non-identity commitments
void lasso_write_commitments(Transcript& tr, const std::vector<HG1>& comms);
std::vector<HG1> lasso_read_commitments(Transcript& tr, size_t count);
void lasso_prove(Ctx&, const LassoPcs&, const lh_lasso_table& table, size_t num_vars, const uint32_t* const* d_dims,
                 Transcript& tr);
void lasso_prove_sharded(Ctx&, const Srs&, const lh_lasso_table& table, size_t num_vars,
                         const uint32_t* const* d_dims, Transcript& tr);

// ------------------------------------------------------------------ verifiers (verifier.cpp; host only)
HFr interpolate_evals(const std::vector<HFr>& evals, const HFr& x);  // barycentric over 0..d (arithmetic.rs:108-136)
HFr horner(const std::vector<HFr>& coeffs, const HFr& x);
struct VerifierParams;  // MultilinearKzgVerifierParams (kzg.rs:79-101)
VerifierParams* mkzg_vp_setup(const HFr* ss, size_t num_vars);
VerifierParams* mkzg_vp_new(const lh_g1& g1, const lh_g2& g2, const lh_g2* ss, size_t num_vars);
void mkzg_vp_export(const VerifierParams&, lh_g1* g1, lh_g2* g2, lh_g2* ss);
size_t mkzg_vp_num_vars(const VerifierParams&);
void mkzg_vp_free(VerifierParams*);
bool pairing_check(const lh_g1* ps, const lh_g2* qs, size_t n);
void mkzg_verify(const VerifierParams&, const HG1& comm, const HFr* point, size_t num_vars, const HFr& eval,
                 Transcript& tr);
void mkzg_batch_verify(const VerifierParams&, size_t num_vars, const HG1* comms, size_t num_comms, const HFr* points,
                       size_t num_points, const lh_evaluation* evals, size_t num_evals, Transcript& tr);
// -> (final claim, challenges)
std::pair<HFr, std::vector<HFr>> sum_check_verify(int prover_kind, size_t num_vars, size_t degree, const HFr& sum,
                                                  Transcript& tr);
void lasso_verify(const PcsBatchVerify& batch_verify, const lh_lasso_table& table, size_t num_vars, Transcript& tr);
// the verifier's side of lasso_argue: Surge and memory-checking identities; claims left to check against commitments
LassoClaims lasso_check(const lh_lasso_table& table, size_t num_vars, Transcript& tr);
void hyperplonk_verify(const PcsBatchVerify& batch_verify, const lh_hp_vparam& vp, const HFr* const* instances,
                       Transcript& tr);
void hyperplonk_verify_phases(const PcsBatchVerify& batch_verify, const lh_hp_vparam& vp,
                              const std::vector<size_t>& num_witness_polys, const std::vector<size_t>& num_challenges,
                              const HFr* const* instances, Transcript& tr);

// ------------------------------------------------------------------ HyperPlonk (hyperplonk.cpp)
// the PolynomialCommitmentScheme the backend is generic over (backend/hyperplonk.rs:76-95)
struct PcsProver {
  std::function<std::vector<HG1>(const Fr* const* d_polys, size_t num_polys, size_t num_vars)> batch_commit;
  // bases whose first 2^nv points commit a (zero-padded) table of 2^nv entries: the small-valued Lasso columns are
  // committed as u32 MSMs against them
  std::function<const G1Affine*(size_t nv)> commit_bases;
  std::function<void(size_t num_vars, const Fr* const* d_polys, size_t num_polys, const HFr* points, size_t num_points,
                     const lh_evaluation* evals, size_t num_evals, Transcript& tr)>
      batch_open;
  // one proof over several GPUs (dev.hpp Shard): batch_commit / batch_open take this rank's shards and add the ranks'
  // partial commitments; shard_bases: this rank's share of commit_bases
  bool sharded_ok = false;
  std::function<const G1Affine*(size_t nv)> shard_bases;
};
PcsProver mkzg_pcs(Ctx&, const Srs&);
PcsProver zeromorph_pcs(Ctx&, const USrs&, size_t poly_size);
void hyperplonk_prove(Ctx&, const PcsProver&, const lh_hp_param& pp, const HFr* const* instances,
                      const Fr* const* d_witness, Transcript& tr);
// multi-phase circuits (hyperplonk.rs:185-205): phase r synthesizes num_witness_polys[r] device tables from the
// challenges of the earlier phases (PlonkishCircuit::synthesize, backend.rs:139), then num_challenges[r] are squeezed
struct HpPhases {
  std::vector<size_t> num_witness_polys, num_challenges;
  std::function<std::vector<const Fr*>(size_t round, const std::vector<HFr>& challenges)> synthesize;
};
void hyperplonk_prove_phases(Ctx&, const PcsProver&, const lh_hp_param& pp, const HpPhases& phases,
                             const HFr* const* instances, Transcript& tr);

}  // namespace lh
