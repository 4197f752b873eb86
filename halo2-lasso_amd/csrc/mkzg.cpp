// pcs::multilinear::kzg on the device: setup / trim / commit / open (reference pcs/multilinear/kzg.rs:166-302, quotients
// pcs/multilinear.rs:72-107) and additive::batch_open (pcs/multilinear.rs:134-235), on one GPU or on a rank's shards.
#include <algorithm>
#include <functional>
#include <chrono>
#include <memory>
#include <thread>
#include "open_columns.hpp"

namespace lh {

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

std::mutex srs_cache_mu;

// bases of level `lvl` that belong to this rank (same index split as the tables); built on first use
const G1Affine* srs_shard_level(Ctx& c, const Srs& srs, size_t lvl) {
  const Shard g(c);
  LH_REQUIRE(g.on, LH_ERR_ARG, "srs shard: no sharded proof is running");
  std::lock_guard<std::mutex> lock(srs_cache_mu);
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
  std::lock_guard<std::mutex> lock(srs_cache_mu);
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
      if (given->ensure_merged) given->ensure_merged();
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
    if (!rem) {  // first step of the column route straight from the 32-bit columns (g' is never formed), else the merged tables
      std::vector<const uint32_t*> cp;
      std::vector<size_t> cl;
      std::vector<Fr> cf;
      for (size_t k = 0; k < small->cols.size(); k++)
        if (!small->coef[k].is_zero()) cp.push_back(small->cols[k].ptr), cl.push_back(small->cols[k].len), cf.push_back(dev(small->coef[k]));
      static const bool from_cols = !(getenv("LH_OPEN_FOLD_COLS") && atoi(getenv("LH_OPEN_FOLD_COLS")) == 0);  // (development A/B)
      if (!(from_cols && !cp.empty() && k_lincomb_fold_small(c, cp.data(), cl.data(), cf.data(), cp.size(), half, dev(point[i]), dst))) {
        if (small->ensure_merged) small->ensure_merged();
        k_lincomb_fold(c, small->merged.data(), small->merged_w.data(), small->merged.size(), half, dev(point[i]), dst);
      }
    }
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
    c.d2h(&remainder, d_poly, sizeof(Fr));
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
  const bool batch_waited = msm_batch(c, jobs.data(), jobs.size(), (G1Affine*)out.data(), ahead ? &combine_columns : nullptr);
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
  if (!batch_waited) c.sync();  // (no entry anywhere in the batch - e.g. a rank whose ranges of the replicated levels are all
                                // empty -: nothing waited for the stream yet, the copy above may not have landed)
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
  // Every opened poly a 32-bit column and an opening that takes them as columns (one GPU): the merged tables are not even
  // written - the sum-check gets the columns and their weights (Ctx::sc_u32_terms), runs its first three rounds from them
  // and binds the merged polys twice in one pass; whoever still needs a table in full (the plain opening route) makes it then.
  bool all_small = small != nullptr && open_small != nullptr && !sharded;
  for (size_t i = 0; i < num_evals && all_small; i++) all_small = small[evals[i].poly].ptr != nullptr;
  Ctx::ScU32Terms hint;
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
    if (all_small) hint.polys.push_back(Ctx::ScU32Terms::Poly{sm, sm_len, wsm});
    else if (sm.empty()) k_lincomb(c, src.data(), w.data(), src.size(), n, m);
    else k_lincomb_mixed(c, src.data(), w.data(), src.size(), sm.data(), sm_len.data(), wsm.data(), sm.size(), n, m);
    merged[j] = m;
  }
  bool merged_there = !all_small;
  auto ensure_merged = [&] {
    if (merged_there) return;
    for (size_t j = 0; j < num_points; j++) {
      const Ctx::ScU32Terms::Poly& pl = hint.polys[j];
      k_lincomb_mixed(c, nullptr, nullptr, 0, pl.col.data(), pl.len.data(), pl.w.data(), pl.col.size(), n, const_cast<Fr*>(merged[j]));
    }
    merged_there = true;
  };
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
  struct HintGuard {  // the hint is this sum-check's alone, whatever way it ends
    Ctx& c;
    ~HintGuard() { c.sc_u32_terms = Ctx::ScU32Terms(); }
  } hint_guard{c};
  if (all_small) c.sc_u32_terms = hint;
  SumCheckResult sc = sum_check_prove(c, LH_SC_COEFFICIENTS, num_vars, expr, merged.data(), num_points, points,
                                      num_points, tilde_gs_sum, tr, false, nullptr, sharded);
  if (all_small) merged_there = c.sc_u32_terms.built;
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
      so.ensure_merged = ensure_merged;
      open_small(nullptr, sc.challenges.data(), so);  // (g' is formed by the opening if it needs it)
      return;
    }
  }
  ensure_merged();
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
