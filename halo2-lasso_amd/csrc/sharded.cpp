// One Lasso proof over 2^rho GPUs.  The reference is single-process: there is nothing to
// cite here; the transcript and proof bytes are exactly those of the single-GPU prover (lasso.cpp).
//
// Layout: a table of 2^m entries is split on the index bits [j, j+rho) (j = ctx.shard_bit): rank s holds
// the 2^(m-rho) entries (hi || lo).  Every kernel is index-agnostic over such a local table because
//   * sum-check pairs are (2b, 2b+1)      -> bit 0, local while it is not a shard bit (rounds 0..j-1),
//   * product-tree / quotient halves      -> top bit, local while the table has more than j+rho variables.
// What crosses ranks (comm.cpp: RCCL all-gathers on the ctx's stream on a multi-GPU node): the u32 lookup columns
// once (the access counters need the global lookup order), D partial sums per sharded round, one partial point
// per MSM job, the residual tables of a sum-check when the shard bits reach bit 0 (2^(m-j) entries per table),
// the tree level / quotient remainder at 2^(j+rho) entries.  Tables of <= j+rho variables (the subtable
// side of the memory check, the top of every tree) are replicated and computed redundantly.
#include <algorithm>
#include <chrono>
#include "host.hpp"

namespace lh {

static size_t lg(size_t v) {
  size_t l = 0;
  while (((size_t)1 << l) < v) l++;
  return l;
}

struct ShardGeom {
  size_t rho, j, rank, R;
  explicit ShardGeom(const Ctx& c) : rho(lg((size_t)c.comm.size)), j(c.shard_bit), rank((size_t)c.comm.rank), R((size_t)c.comm.size) {}
  // a table of num_vars variables is held in shards (a single rank, rho = 0, goes through the same code paths: its
  // "exchange" round still needs one round before it)
  bool sharded(size_t num_vars) const { return num_vars >= j + (rho ? rho : 1) + 1; }
};

// ------------------------------------------------------------------ grand product
struct ShardedLeaves {
  const Fr* ptr;    // local shard (2^(nv-rho)) when sharded, the full table otherwise
  size_t num_vars;  // global
  bool plus_one = false;  // its leaves are the previous tree's leaves + 1 (prover.cpp prove_grand_product: plus_one)
};

static GrandProductResult prove_grand_product_sharded(Ctx& c, const std::vector<ShardedLeaves>& trees, Transcript& tr) {
  const ShardGeom g(c);
  const size_t B = trees.size();
  LH_REQUIRE(B != 0 && 2 * B + 1 <= (size_t)SC_MAX_TABLES && B <= LH_SC_MAX_TERMS, LH_ERR_ARG,
             "grand product: bad number of trees");
  ArenaScope scope(c.arena);
  size_t max_depth = 0;
  // level[b][h]: 2^(h+1) nodes; sharded (local 2^(h+1-rho)) iff h+1 >= j+rho+1
  std::vector<std::vector<const Fr*>> level(B);
  for (size_t b = 0; b < B; b++) {
    const size_t nv = trees[b].num_vars;
    LH_REQUIRE(nv >= 1 && nv < 32, LH_ERR_ARG, "grand product: every tree needs >= 2 leaves");
    max_depth = std::max(max_depth, nv);
    level[b].resize(nv);
    level[b][nv - 1] = trees[b].ptr;
    for (size_t h = nv - 1; h >= 1; h--) {  // level h-1 (2^h nodes) from level h (2^(h+1) nodes)
      if (g.sharded(h + 1)) {
        const size_t half_local = (size_t)1 << (h - g.rho);
        Fr* up = c.arena.alloc_n<Fr>(half_local);
        k_tree_up(c, level[b][h], half_local, up);
        if (g.sharded(h)) {
          level[b][h - 1] = up;
        } else {  // 2^h = 2^(j+rho) nodes, shard bits on top: exchange once, replicated from here up
          Fr* rep = c.arena.alloc_n<Fr>((size_t)1 << h);
          comm_gather_concat(c, up, half_local, rep);
          level[b][h - 1] = rep;
        }
      } else {
        const size_t half = (size_t)1 << h;
        Fr* up = c.arena.alloc_n<Fr>(half);
        k_tree_up(c, level[b][h], half, up);
        level[b][h - 1] = up;
      }
    }
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
  for (size_t h = 0; h < max_depth; h++) {
    std::vector<size_t> active;
    for (size_t b = 0; b < B; b++)
      if (trees[b].num_vars > h) active.push_back(b);
    const bool sh = g.sharded(h + 1);
    const size_t half = sh ? (size_t)1 << (h - g.rho) : (size_t)1 << h;
    std::vector<HFr> x, evals;
    if (h == 0) {
      for (size_t b : active) {
        evals.push_back(top[2 * b]);
        evals.push_back(top[2 * b + 1]);
      }
    } else {
      HFr lam = tr.squeeze_challenge();
      HFr claim = HFr::zero(), power = HFr::one();
      lh_sop expr;
      memset(&expr, 0, sizeof(expr));
      expr.global_eq = 0;
      std::vector<const Fr*> polys;
      // leaf layer of (A, A + 1) tree pairs: c_A l r + c_B (l + 1)(r + 1) written out over the A tables alone, constant
      // term included (prover.cpp prove_grand_product): half the tables in every round and in the residual exchange
      bool pairs = active.size() % 2 == 0;
      for (size_t k = 0; k < active.size() && pairs; k++) {
        const size_t b = active[k];
        pairs = trees[b].num_vars == h + 1 && (k % 2 == 0 ? !trees[b].plus_one : (trees[b].plus_one && active[k - 1] + 1 == b));
      }
      SumCheckResult sc;
      if (pairs && 3 * (active.size() / 2) + 1 <= (size_t)LH_SC_MAX_TERMS) {
        const size_t P = active.size() / 2;
        HFr cw_sum = HFr::zero();
        uint32_t t = 0;
        for (size_t i = 0; i < P; i++) {
          const size_t a = active[2 * i], bb = active[2 * i + 1];
          const HFr c_a = power, c_b = power * lam, cs = c_a + c_b;
          claim += claims[a] * c_a + claims[bb] * c_b;
          power = c_b * lam;
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
        sc = sh ? sum_check_prove_sharded(c, LH_SC_EVALUATIONS, h, expr, polys.data(), polys.size(), y.data(), 1, claim, tr)
                : sum_check_prove(c, LH_SC_EVALUATIONS, h, expr, polys.data(), polys.size(), y.data(), 1, claim, tr);
        x = sc.challenges;
        for (size_t i = 0; i < P; i++) {
          const HFr l = sc.evals[2 * i], r = sc.evals[2 * i + 1];
          evals.push_back(l), evals.push_back(r);
          evals.push_back(l + HFr::one()), evals.push_back(r + HFr::one());
        }
      } else {
      for (size_t k = 0; k < active.size(); k++) {
        size_t b = active[k];
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
      sc = sh ? sum_check_prove_sharded(c, LH_SC_EVALUATIONS, h, expr, polys.data(), polys.size(),
                                                       y.data(), 1, claim, tr)
                             : sum_check_prove(c, LH_SC_EVALUATIONS, h, expr, polys.data(), polys.size(), y.data(), 1,
                                               claim, tr);
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
      if (trees[b].num_vars == h + 1) {
        res.claims[b] = claims[b];
        res.points[b] = y;
      }
    }
  }
  return res;
}

// ------------------------------------------------------------------ SRS shards
// bases of level `lvl` that belong to this rank (same index split as the tables); built on first use
static const G1Affine* srs_shard_level(Ctx& c, const Srs& srs, size_t lvl) {
  const ShardGeom g(c);
  if (srs.shard_rank != (int)g.rank || srs.shard_R != g.R || srs.shard_j != g.j) {
    for (G1Affine* p : srs.shard_levels)
      if (p) (void)hipFree(p);
    srs.shard_levels.assign(srs.num_vars + 1, nullptr);
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

// ------------------------------------------------------------------ open / batch_open on shards
static void mkzg_open_sharded(Ctx& c, const Srs& srs, const Fr* d_poly_local, size_t num_vars, const HFr* point,
                              Transcript& tr) {
  const ShardGeom g(c);
  LH_REQUIRE(num_vars <= srs.num_vars, LH_ERR_INVALID_PCS_PARAM, "Too many variates of poly to open");
  LH_REQUIRE(num_vars >= g.j + g.rho, LH_ERR_ARG, "sharded open: poly too small");
  ArenaScope scope(c.arena);
  const size_t n_local = (size_t)1 << (num_vars - g.rho);
  const size_t cut = g.j + g.rho;  // levels >= cut are sharded, below replicated
  Fr* q_local = c.arena.alloc_n<Fr>(std::max<size_t>(n_local, 1));       // q_i local at offset 2^(i-rho) - 2^j... packed below
  Fr* remA = c.arena.alloc_n<Fr>(std::max<size_t>(n_local >> 1, 1));
  Fr* remB = c.arena.alloc_n<Fr>(std::max<size_t>(n_local >> 2, 1));
  std::vector<MsmJob> jobs(num_vars);
  const Fr* rem = d_poly_local;
  size_t q_off = 0;
  int flip = 0;
  for (size_t i = num_vars; i-- > cut;) {
    const size_t half_local = (size_t)1 << (i - g.rho);
    Fr* dst = flip ? remB : remA;
    flip ^= 1;
    k_quotient_step(c, rem, half_local, dev(point[i]), q_local + q_off, dst);
    jobs[i] = MsmJob{q_local + q_off, false, srs_shard_level(c, srs, i), half_local};
    q_off += half_local;
    rem = dst;
  }
  // remainder: 2^cut entries globally, shard bits on top -> replicate and finish as on one GPU
  const size_t n_rep = (size_t)1 << cut;
  Fr* rep = c.arena.alloc_n<Fr>(n_rep);
  comm_gather_concat(c, rem, (size_t)1 << g.j, rep);
  Fr* q_rep = c.arena.alloc_n<Fr>(n_rep);
  Fr* repA = c.arena.alloc_n<Fr>(std::max<size_t>(n_rep >> 1, 1));
  Fr* repB = c.arena.alloc_n<Fr>(std::max<size_t>(n_rep >> 2, 1));
  rem = rep;
  for (size_t i = cut; i-- > 0;) {
    const size_t half = (size_t)1 << i;
    Fr* dst = ((cut - i) & 1) ? repA : repB;
    k_quotient_step(c, rem, half, dev(point[i]), q_rep + (half - 1), dst);
    jobs[i] = MsmJob{q_rep + (half - 1), false, srs.eq(i), half};
    rem = dst;
  }
  std::vector<HG1> comms(num_vars);
  msm_batch(c, jobs.data(), num_vars, (G1Affine*)comms.data());
  if (num_vars > cut) comm_sum_points(c, comms.data() + cut, num_vars - cut);  // partial sums of the sharded levels
  tr.write_commitments(comms);
}

static void mkzg_batch_open_sharded(Ctx& c, const Srs& srs, size_t num_vars, const Fr* const* d_polys_local,
                                    size_t num_polys, const HFr* points, size_t num_points,
                                    const lh_evaluation* evals, size_t num_evals, Transcript& tr) {
  const ShardGeom g(c);
  LH_REQUIRE(num_evals >= 2 && 2 * num_points <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "batch open: bad shape");
  size_t ell = 0;
  while (((size_t)1 << ell) < num_evals) ell++;
  std::vector<HFr> t = tr.squeeze_challenges(ell);
  std::vector<HFr> eq_xt = host_eq_xy(t);
  ArenaScope scope(c.arena);
  const size_t n_local = (size_t)1 << (num_vars - g.rho);
  std::vector<const Fr*> merged(num_points);
  for (size_t jp = 0; jp < num_points; jp++) {
    std::vector<const Fr*> src;
    std::vector<Fr> w;
    for (size_t i = 0; i < num_evals; i++)
      if (evals[i].point == jp) {
        LH_REQUIRE(evals[i].poly < num_polys, LH_ERR_ARG, "batch open: bad evaluation");
        src.push_back(d_polys_local[evals[i].poly]);
        w.push_back(dev(eq_xt[i]));
      }
    LH_REQUIRE(!src.empty(), LH_ERR_ARG, "batch open: a point without evaluations");
    Fr* m = c.arena.alloc_n<Fr>(n_local);
    k_lincomb(c, src.data(), w.data(), src.size(), n_local, m);
    merged[jp] = m;
  }
  lh_sop expr;
  memset(&expr, 0, sizeof(expr));
  expr.global_eq = -1;
  expr.num_terms = (uint32_t)num_points;
  const HFr one = HFr::one();
  for (size_t jp = 0; jp < num_points; jp++) {
    memcpy(&expr.coeff[jp], &one, 32);
    expr.num_factors[jp] = 2;
    expr.factor[jp][0] = (uint8_t)(num_points + jp);
    expr.factor[jp][1] = (uint8_t)jp;
  }
  HFr tilde = HFr::zero();
  for (size_t i = 0; i < num_evals; i++) {
    HFr v;
    memcpy(&v, &evals[i].value, 32);
    tilde += v * eq_xt[i];
  }
  SumCheckResult sc = sum_check_prove_sharded(c, LH_SC_COEFFICIENTS, num_vars, expr, merged.data(), num_points, points,
                                              num_points, tilde, tr);
  std::vector<Fr> w(num_points);
  for (size_t jp = 0; jp < num_points; jp++)
    w[jp] = dev(host_eq_xy_eval(sc.challenges.data(), points + jp * num_vars, num_vars));
  Fr* g_prime = c.arena.alloc_n<Fr>(n_local);
  k_lincomb(c, merged.data(), w.data(), num_points, n_local, g_prime);
  mkzg_open_sharded(c, srs, g_prime, num_vars, sc.challenges.data(), tr);
}

// ------------------------------------------------------------------ Lasso
static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

void lasso_prove_sharded(Ctx& c, const Srs& srs, const lh_lasso_table& tb, size_t n,
                         const uint32_t* const* d_dims_local, Transcript& tr) {
  LH_REQUIRE(c.has_comm, LH_ERR_ARG, "lasso_prove_sharded: no communicator attached");
  const ShardGeom g(c);
  LH_REQUIRE(g.R >= 1 && ((size_t)1 << g.rho) == g.R, LH_ERR_ARG, "sharded prove: the number of ranks must be a power of two");
  const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  LH_REQUIRE(cc >= 1 && cc <= LH_LASSO_MAX_CHUNKS && alpha >= 1 && alpha <= LH_LASSO_MAX_MEMORIES, LH_ERR_ARG,
             "lasso: bad table shape");
  LH_REQUIRE(8 * alpha + 1 <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "lasso: too many memories for one GKR batch");
  LH_REQUIRE(g.j >= 1 && l <= g.j + g.rho, LH_ERR_ARG, "sharded prove: need shard_bit + rho >= chunk_bits (subtables replicated)");
  LH_REQUIRE(n >= g.j + g.rho + 1 && n >= l, LH_ERR_ARG, "sharded prove: 2^num_vars lookups are too few to shard");
  if (n > srs.num_vars) throw Error(LH_ERR_INVALID_PCS_PARAM, "Too many variates of poly to commit");
  for (size_t i = 0; i < alpha; i++) {
    LH_REQUIRE(tb.memory_chunk[i] < cc && tb.memory_subtable[i] <= LH_SUBTABLE_XOR, LH_ERR_ARG, "lasso: bad memory");
    if (tb.memory_subtable[i] != LH_SUBTABLE_IDENTITY) LH_REQUIRE(l % 2 == 0, LH_ERR_ARG, "lasso: odd chunk_bits");
  }
  const size_t N = (size_t)1 << n, M = (size_t)1 << l, NL = N >> g.rho;
  double t0 = now_ms(), t_prev = t0;
  double* ph = c.lasso_ms;
  auto lap = [&](int idx) {
    c.sync();
    double t = now_ms();
    ph[idx] = t - t_prev;
    t_prev = t;
  };
  ArenaScope scope(c.arena);
  std::vector<uint32_t*> rts_pre(cc), fcs_pre(cc);
  for (size_t j = 0; j < cc; j++) {
    rts_pre[j] = c.arena.alloc_n<uint32_t>(NL);
    fcs_pre[j] = c.arena.alloc_n<uint32_t>(M);
  }

  // ---- witness: read_ts[k] = number of earlier lookups (global order) of the same address: a stable sort of the
  // whole column.  The 4-byte columns are all-gathered once (1/8 of a field-sized table) and every rank runs the
  // counters on the full column (3 % of a 2^24 proof), keeping its own share of read_ts; final_cts is replicated.
  std::vector<uint32_t*> rts_l(cc), fcs(cc), E_l(alpha);
  std::vector<const uint32_t*> dim_l(cc);
  {
    ArenaScope wscope(c.arena);  // the full columns are only needed here
    uint32_t* gathered = c.arena.alloc_n<uint32_t>(N);
    uint32_t* full = c.arena.alloc_n<uint32_t>(N);
    uint32_t* rts_full = c.arena.alloc_n<uint32_t>(N);
    // (the outputs rts_pre / fcs_pre were allocated in the outer scope above)
    for (size_t j = 0; j < cc; j++) {
      dim_l[j] = d_dims_local[j];
      comm_all_gather_dev(c, d_dims_local[j], gathered, NL * sizeof(uint32_t));
      k_shard_merge(c, gathered, NL, g.j, g.rho, 4, full);
      k_lasso_counters(c, full, N, M, rts_full, fcs_pre[j]);
      k_shard_extract(c, rts_full, NL, g.j, g.rho, g.rank, 4, rts_pre[j]);
    }
  }
  for (size_t j = 0; j < cc; j++) rts_l[j] = rts_pre[j], fcs[j] = fcs_pre[j];
  LassoG gg;
  memset(&gg, 0, sizeof(gg));
  for (size_t i = 0; i < alpha; i++) {
    E_l[i] = c.arena.alloc_n<uint32_t>(NL);
    k_lasso_subtable_read(c, (int)tb.memory_subtable[i], (uint32_t)l, dim_l[tb.memory_chunk[i]], NL, E_l[i]);
    gg.e[i] = E_l[i];
  }
  gg.num_terms = tb.num_terms;
  for (uint32_t m = 0; m < tb.num_terms; m++) {
    memcpy(&gg.coeff[m], &tb.g_coeff[m], 32);
    gg.nfac[m] = tb.g_num_factors[m];
    for (int k = 0; k < LH_SC_MAX_FACTORS; k++) gg.fac[m][k] = tb.g_factor[m][k];
  }
  Fr* a = c.arena.alloc_n<Fr>(NL);
  k_lasso_output(c, gg, NL, a);
  lap(0);

  // ---- commitments: partial MSMs over the local shards, final_cts (replicated) in full
  tr.common_field_element(HFr::from_u64(n));
  tr.common_field_element(HFr::from_u64(l));
  tr.common_field_element(HFr::from_u64(cc));
  tr.common_field_element(HFr::from_u64(alpha));
  {
    // as in lasso.cpp: columns that are linear in others are committed by linearity (after the partial commitments
    // of the shards have been summed), not by an MSM of their own
    const G1Affine* bases_l = srs_shard_level(c, srs, n);
    bool linear_g = true;
    for (uint32_t m = 0; m < tb.num_terms; m++) linear_g = linear_g && tb.g_num_factors[m] == 1;
    const size_t total = 1 + 3 * cc + alpha;
    std::vector<MsmJob> jobs;
    std::vector<size_t> slot;
    auto add_job = [&](size_t pos, const void* col, bool u32, const G1Affine* bases, size_t len) {
      jobs.push_back(MsmJob{col, u32, bases, len});
      slot.push_back(pos);
    };
    if (!linear_g) add_job(0, a, false, bases_l, NL);
    std::vector<size_t> dim_job(cc);
    for (size_t j = 0; j < cc; j++) dim_job[j] = jobs.size(), add_job(1 + j, dim_l[j], true, bases_l, NL);
    // read_ts columns in packed pairs, E = T[dim] from dim's bucket sums: as in lasso.cpp, on this rank's shards (the
    // replicated final_cts bound every shard's read_ts)
    std::vector<HG1> second(cc);
    std::vector<size_t> second_of;
    {
      std::vector<uint32_t> ors(cc, 0);
      std::vector<const uint32_t*> cols(fcs.begin(), fcs.end());
      if (cc >= 2 && NL >= ((size_t)1 << 12)) k_or_u32(c, cols.data(), cc, M, ors.data());
      auto bits_of = [](uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; };
      for (size_t j = 0; j < cc; j++) {
        const uint32_t b0 = std::max(bits_of(ors[j]), 4u), b1 = j + 1 < cc ? bits_of(ors[j + 1]) : 0;
        if (ors[j] && j + 1 < cc && ors[j + 1] && b0 + b1 <= MSM_PACK_MAX_BITS) {
          uint32_t* packed = c.arena.alloc_n<uint32_t>(NL);
          k_pack_u32(c, rts_l[j], rts_l[j + 1], b0, NL, packed);
          add_job(1 + cc + j, packed, true, bases_l, NL);
          jobs.back().pack_shift = b0;
          jobs.back().out_second = (G1Affine*)&second[j + 1];
          second_of.push_back(j + 1);
          j++;
        } else {
          add_job(1 + cc + j, rts_l[j], true, bases_l, NL);
        }
      }
    }
    SubtableOrders orders(c, l);
    for (size_t i = 0; i < alpha; i++)
      if (tb.memory_subtable[i] != LH_SUBTABLE_IDENTITY) {
        add_job(1 + 2 * cc + i, E_l[i], true, bases_l, NL);
        if (l <= 20) {
          MsmJob& jb = jobs.back();
          jb.derived_parent = (int)dim_job[tb.memory_chunk[i]];
          orders.get((int)tb.memory_subtable[i], &jb.d_table, &jb.d_order);
          jb.table_in_bits = (uint32_t)l, jb.table_out_bits = (uint32_t)(l / 2);
        }
      }
    const size_t num_sharded = jobs.size();
    for (size_t j = 0; j < cc; j++) add_job(1 + 2 * cc + alpha + j, fcs[j], true, srs.eq(n), M);
    std::vector<HG1> part(jobs.size()), comms(total);
    msm_batch(c, jobs.data(), jobs.size(), (G1Affine*)part.data());
    {
      // partial commitments of the shards (the second outputs of packed jobs included) -> their sums
      std::vector<HG1> sums(part.begin(), part.begin() + num_sharded);
      for (size_t j : second_of) sums.push_back(second[j]);
      comm_sum_points(c, sums.data(), sums.size());
      for (size_t k = 0; k < num_sharded; k++) part[k] = sums[k];
      for (size_t q = 0; q < second_of.size(); q++) second[second_of[q]] = sums[num_sharded + q];
    }
    for (size_t j : second_of) comms[1 + cc + j] = second[j];
    for (size_t k = 0; k < jobs.size(); k++) comms[slot[k]] = part[k];
    for (size_t i = 0; i < alpha; i++)
      if (tb.memory_subtable[i] == LH_SUBTABLE_IDENTITY) comms[1 + 2 * cc + i] = comms[1 + tb.memory_chunk[i]];
    if (linear_g) {
      host::G1Xyzz acc = host::G1Xyzz::identity();
      for (uint32_t m = 0; m < tb.num_terms; m++) {
        HFr co;
        memcpy(&co, &tb.g_coeff[m], 32);
        acc = host::g1_add(acc, host::g1_mul(host::g1_from_affine(comms[1 + 2 * cc + tb.g_factor[m][0]]), co));
      }
      comms[0] = host::g1_to_affine(acc);
    }
    lasso_write_commitments(tr, comms);
  }
  lap(1);

  // ---- field views (local shards; final_cts both as the l-variable table and as the padded n-variable shard)
  const size_t num_n = 1 + 2 * cc + alpha;
  std::vector<const Fr*> polys_n(num_n), fcs_full(cc), fcs_pad_l(cc);
  polys_n[0] = a;
  auto fr_view = [&](const uint32_t* src, size_t len) {
    Fr* d = c.arena.alloc_n<Fr>(len);
    k_fr_from_u32(c, src, len, d);
    return d;
  };
  for (size_t j = 0; j < cc; j++) {
    polys_n[1 + j] = fr_view(dim_l[j], NL);
    polys_n[1 + cc + j] = fr_view(rts_l[j], NL);
    fcs_full[j] = fr_view(fcs[j], M);
    // padded to 2^(j+rho) entries, then this rank's 2^j of them, then zeros up to the local length
    const size_t cutn = (size_t)1 << (g.j + g.rho);
    Fr* padded = c.arena.alloc_n<Fr>(cutn);
    LH_HIP(hipMemcpyAsync(padded, fcs_full[j], M * sizeof(Fr), hipMemcpyDeviceToDevice, c.stream));
    if (M < cutn) LH_HIP(hipMemsetAsync(padded + M, 0, (cutn - M) * sizeof(Fr), c.stream));
    Fr* loc = c.arena.alloc_n<Fr>(NL);
    LH_HIP(hipMemsetAsync(loc, 0, NL * sizeof(Fr), c.stream));
    k_shard_extract(c, padded, (size_t)1 << g.j, g.j, g.rho, g.rank, sizeof(Fr), loc);
    fcs_pad_l[j] = loc;
  }
  for (size_t i = 0; i < alpha; i++) polys_n[1 + 2 * cc + i] = fr_view(E_l[i], NL);
  const Fr* const* E_fr = polys_n.data() + 1 + 2 * cc;

  // ---- Surge
  std::vector<HFr> r = tr.squeeze_challenges(n);
  HFr v = evaluate_polys_sharded(c, &polys_n[0], 1, n, r.data())[0];
  tr.write_field_element(v);
  lh_sop surge;
  memset(&surge, 0, sizeof(surge));
  surge.global_eq = 0;
  surge.num_terms = tb.num_terms;
  for (uint32_t m = 0; m < tb.num_terms; m++) {
    surge.coeff[m] = tb.g_coeff[m];
    surge.num_factors[m] = tb.g_num_factors[m];
    for (int k = 0; k < LH_SC_MAX_FACTORS; k++) surge.factor[m][k] = tb.g_factor[m][k];
  }
  bool linear_g = true;
  for (uint32_t m = 0; m < tb.num_terms; m++) linear_g = linear_g && tb.g_num_factors[m] == 1;
  SumCheckResult sc;
  if (linear_g) {
    // g linear: eq * g(E) is eq * a entry by entry and binding is linear (lasso.cpp lasso_argue): the rounds run over the
    // output column's shard alone; E_i(r_z) by sharded inner products
    lh_sop one_term;
    memset(&one_term, 0, sizeof(one_term));
    one_term.global_eq = 0;
    one_term.num_terms = 1;
    const HFr one = HFr::one();
    memcpy(&one_term.coeff[0], &one, 32);
    one_term.num_factors[0] = 1;
    one_term.factor[0][0] = 0;
    sc = sum_check_prove_sharded(c, LH_SC_EVALUATIONS, n, one_term, &polys_n[0], 1, r.data(), 1, v, tr);
    sc.evals = evaluate_polys_sharded(c, E_fr, alpha, n, sc.challenges.data());
  } else {
    sc = sum_check_prove_sharded(c, LH_SC_EVALUATIONS, n, surge, E_fr, alpha, r.data(), 1, v, tr);
  }
  const std::vector<HFr>& r_z = sc.challenges;
  tr.write_field_elements(sc.evals);
  lap(2);

  // ---- memory checking
  HFr gamma = tr.squeeze_challenge();
  HFr tau = tr.squeeze_challenge();
  HFr gamma2 = gamma * gamma;
  std::vector<ShardedLeaves> trees(4 * alpha);
  for (size_t i = 0; i < alpha; i++) {
    size_t j = tb.memory_chunk[i];
    Fr* rs = c.arena.alloc_n<Fr>(NL);
    Fr* ws = c.arena.alloc_n<Fr>(NL);
    Fr* in = c.arena.alloc_n<Fr>(M);
    Fr* fi = c.arena.alloc_n<Fr>(M);
    k_lasso_rw_leaves(c, dim_l[j], E_l[i], rts_l[j], NL, dev(gamma), dev(gamma2), dev(tau), rs, ws);
    k_lasso_if_leaves(c, (int)tb.memory_subtable[i], (uint32_t)l, fcs[j], M, dev(gamma), dev(gamma2), dev(tau), in, fi);
    trees[2 * i] = ShardedLeaves{rs, n};
    trees[2 * i + 1] = ShardedLeaves{ws, n, true};
    trees[2 * alpha + 2 * i] = ShardedLeaves{in, l};
    trees[2 * alpha + 2 * i + 1] = ShardedLeaves{fi, l};
  }
  lap(3);
  GrandProductResult gp = prove_grand_product_sharded(c, trees, tr);
  const std::vector<HFr>& r_N = gp.points[0];
  const std::vector<HFr>& r_M = gp.points[2 * alpha];
  lap(4);

  std::vector<HFr> ev_n = evaluate_polys_sharded(c, polys_n.data() + 1, 2 * cc + alpha, n, r_N.data());
  std::vector<HFr> ev_l = evaluate_polys(c, fcs_full.data(), cc, l, r_M.data());
  tr.write_field_elements(ev_n);
  tr.write_field_elements(ev_l);
  lap(5);

  {
    std::vector<HFr> points(4 * n, HFr::zero());
    std::copy(r.begin(), r.end(), points.begin());
    std::copy(r_z.begin(), r_z.end(), points.begin() + n);
    std::copy(r_N.begin(), r_N.end(), points.begin() + 2 * n);
    std::copy(r_M.begin(), r_M.end(), points.begin() + 3 * n);
    std::vector<lh_evaluation> evs;
    auto push = [&](size_t poly, size_t point, const HFr& val) {
      lh_evaluation e;
      e.poly = (uint32_t)poly;
      e.point = (uint32_t)point;
      memcpy(&e.value, &val, 32);
      evs.push_back(e);
    };
    push(0, 0, v);
    for (size_t i = 0; i < alpha; i++) push(1 + 2 * cc + i, 1, sc.evals[i]);
    for (size_t j = 0; j < cc; j++) push(1 + j, 2, ev_n[j]);
    for (size_t j = 0; j < cc; j++) push(1 + cc + j, 2, ev_n[cc + j]);
    for (size_t i = 0; i < alpha; i++) push(1 + 2 * cc + i, 2, ev_n[2 * cc + i]);
    for (size_t j = 0; j < cc; j++) push(num_n + j, 3, ev_l[j]);
    std::vector<const Fr*> all(polys_n);
    all.insert(all.end(), fcs_pad_l.begin(), fcs_pad_l.end());
    mkzg_batch_open_sharded(c, srs, n, all.data(), all.size(), points.data(), 4, evs.data(), evs.size(), tr);
  }
  lap(6);
  ph[7] = 0;
  ph[8] = now_ms() - t0;
}

}  // namespace lh
