// HyperPlonk::prove (reference backend/hyperplonk.rs:164-291) with LogUp lookups and the permutation
// argument, single- and multi-phase circuits (the phase loop of hyperplonk.rs:185-205).  Transcript schedule: SURVEY.md §3.1; restated in
// oracle/pyref/hyperplonk.py, which the tests compare against byte for byte.
#include <algorithm>
#include <chrono>
#include <map>
#include <set>
#include "host.hpp"
#include "expr.hpp"

namespace lh {

// ------------------------------------------------------------------ rotation points (poly/multilinear.rs:477-549)
std::vector<size_t> rotation_point_pattern(bool next, size_t num_vars, size_t distance) {
  const size_t rem = next ? bh_primitive(num_vars) : bh_x_inv(num_vars);
  std::vector<size_t> pat((size_t)1 << distance, 0);
  for (size_t depth = 0; depth < distance; depth++) {
    size_t step = (size_t)1 << (distance - depth);
    for (size_t e = 0; e < pat.size(); e += step) {
      size_t o = e + step / 2;
      size_t rot = next ? pat[e] << 1 : pat[e] >> 1;
      pat[o] = rot ^ rem;
      pat[e] = rot;
    }
  }
  return pat;
}

std::vector<std::vector<HFr>> rotation_eval_points(const std::vector<HFr>& x, int rotation) {
  if (rotation == 0) return {x};
  const size_t n = x.size(), distance = (size_t)std::abs(rotation), num_x = n - distance;
  std::vector<std::vector<HFr>> out;
  const HFr one = HFr::one(), zero = HFr::zero();
  auto bit = [](size_t p, size_t i) { return (p >> i) & 1; };
  if (rotation < 0) {
    for (size_t p : rotation_point_pattern(false, n, distance)) {
      std::vector<HFr> pt;
      for (size_t i = 0; i < num_x; i++) pt.push_back(bit(p, i) ? one - x[distance + i] : x[distance + i]);
      for (size_t i = 0; i < distance; i++) pt.push_back(bit(p, i + num_x) ? one : zero);
      out.push_back(pt);
    }
  } else {
    for (size_t p : rotation_point_pattern(true, n, distance)) {
      std::vector<HFr> pt;
      for (size_t i = 0; i < distance; i++) pt.push_back(bit(p, i) ? one : zero);
      for (size_t i = 0; i < num_x; i++) pt.push_back(bit(p, i + distance) ? one - x[i] : x[i]);
      out.push_back(pt);
    }
  }
  return out;
}

// ------------------------------------------------------------------ lookup_compressed_polys (prover.rs:50-137)
// sum_i beta^i * expr_i as ONE monomial list over row atoms, evaluated row-wise by expr_rows_kernel
static void compressed_poly(Ctx& c, const lh_expr* exprs, size_t width, const std::vector<HFr>& betas,
                            const std::vector<const Fr*>& polys, const HFr* challenges, size_t num_challenges,
                            size_t num_vars, Fr* out) {
  std::vector<RowsAtom> atoms;
  std::vector<Fr> coeff;
  std::vector<uint32_t> off{0};
  std::vector<uint8_t> fac;
  const size_t n = (size_t)1 << num_vars;
  for (size_t w = 0; w < width; w++) {
    ExpandedExpr ex = expand_expr(exprs[w], challenges, num_challenges);
    std::vector<int> id(ex.atoms.size());
    for (size_t a = 0; a < ex.atoms.size(); a++) {
      const ExprAtom& at = ex.atoms[a];
      RowsAtom ra;
      memset(&ra, 0, sizeof(ra));
      if (at.kind == LH_EX_POLYNOMIAL) {
        LH_REQUIRE((size_t)at.a < polys.size(), LH_ERR_ARG, "lookup expression: poly index out of range");
        ra.kind = ROWS_ATOM_POLY, ra.table = polys[at.a], ra.rot = at.b;
      } else if (at.kind == LH_EX_IDENTITY) {
        ra.kind = ROWS_ATOM_IDENTITY;
      } else if (at.kind == LH_EX_LAGRANGE) {
        long long m = (long long)at.a % (long long)n;
        if (m < 0) m += (long long)n;
        ra.kind = ROWS_ATOM_LAGRANGE, ra.hot = bh_nth(num_vars, (size_t)m);
      } else {
        throw Error(LH_ERR_ARG, "lookup expression: eq_xy is not allowed here");  // prover.rs:108 unreachable!()
      }
      LH_REQUIRE(atoms.size() < 250, LH_ERR_ARG, "lookup expression: too many atoms");
      id[a] = (int)atoms.size();
      atoms.push_back(ra);
    }
    for (const ExprMono& m : ex.monos) {
      coeff.push_back(dev(m.coeff * betas[w]));
      for (uint16_t a : m.atoms) fac.push_back((uint8_t)id[a]);
      off.push_back((uint32_t)fac.size());
    }
  }
  if (coeff.empty()) {
    LH_HIP(hipMemsetAsync(out, 0, n * sizeof(Fr), c.stream));
    return;
  }
  if (fac.empty()) fac.push_back(0);
  if (atoms.empty()) atoms.push_back(RowsAtom{nullptr, 0, ROWS_ATOM_IDENTITY, 0});
  ArenaScope scope(c.arena);
  Fr* d_coeff = c.arena.alloc_n<Fr>(coeff.size());
  uint32_t* d_off = c.arena.alloc_n<uint32_t>(off.size());
  uint8_t* d_fac = c.arena.alloc_n<uint8_t>(fac.size());
  RowsAtom* d_atoms = c.arena.alloc_n<RowsAtom>(atoms.size());
  LH_HIP(hipMemcpyAsync(d_coeff, coeff.data(), coeff.size() * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemcpyAsync(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemcpyAsync(d_fac, fac.data(), fac.size(), hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemcpyAsync(d_atoms, atoms.data(), atoms.size() * sizeof(RowsAtom), hipMemcpyHostToDevice, c.stream));
  RowsExpr re;
  re.num_terms = (uint32_t)coeff.size(), re.num_vars = (uint32_t)num_vars;
  re.primitive = bh_primitive(num_vars), re.x_inv = bh_x_inv(num_vars);
  re.coeff = d_coeff, re.off = d_off, re.fac = d_fac, re.atoms = d_atoms;
  k_expr_rows(c, re, n, out);
  c.sync();  // host staging vectors die with this frame
}

// ------------------------------------------------------------------ HyperPlonk::prove
PcsProver mkzg_pcs(Ctx& c, const Srs& srs) {
  PcsProver p;
  p.batch_commit = [&c, &srs](const Fr* const* polys, size_t n, size_t nv) { return mkzg_batch_commit(c, srs, polys, n, nv); };
  p.sharded_ok = true;  // (mkzg_batch_commit / mkzg_batch_open read the ctx's Shard geometry)
  p.shard_bases = [&c, &srs](size_t nv) { return srs_shard_level(c, srs, nv); };
  p.commit_bases = [&srs](size_t nv) {
    LH_REQUIRE(nv <= srs.num_vars, LH_ERR_INVALID_PCS_PARAM, "Too many variates of poly to commit");
    return srs.eq(nv);
  };
  p.batch_open = [&c, &srs](size_t nv, const Fr* const* polys, size_t n, const HFr* points, size_t np,
                            const lh_evaluation* evals, size_t ne, Transcript& tr) {
    mkzg_batch_open(c, srs, nv, polys, n, points, np, evals, ne, tr);
  };
  return p;
}
PcsProver zeromorph_pcs(Ctx& c, const USrs& srs, size_t poly_size) {
  PcsProver p;
  p.batch_commit = [&c, &srs, poly_size](const Fr* const* polys, size_t n, size_t nv) {
    return zeromorph_batch_commit(c, srs, poly_size, polys, n, nv);
  };
  p.commit_bases = [&srs, poly_size](size_t nv) {
    LH_REQUIRE(((size_t)1 << nv) <= poly_size, LH_ERR_INVALID_PCS_PARAM, "Too many variates of poly to commit");
    return (const G1Affine*)srs.d_powers;
  };
  p.batch_open = [&c, &srs, poly_size](size_t nv, const Fr* const* polys, size_t n, const HFr* points, size_t np,
                                       const lh_evaluation* evals, size_t ne, Transcript& tr) {
    zeromorph_batch_open(c, srs, poly_size, nv, polys, n, points, np, evals, ne, tr);
  };
  return p;
}

namespace {
struct PhaseTimer {  // LH_HP_DEBUG=1: wall-clock per phase on stderr (development aid)
  Ctx& c;
  bool on;
  double t;
  static double now() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  explicit PhaseTimer(Ctx& c_) : c(c_), on(getenv("LH_HP_DEBUG") != nullptr), t(now()) {}
  void lap(const char* what) {
    if (!on) return;
    c.sync();
    double n = now();
    fprintf(stderr, "[hyperplonk] %-28s %8.3f ms\n", what, n - t);
    t = n;
  }
};
}  // namespace

void hyperplonk_prove(Ctx& c, const PcsProver& pcs, const lh_hp_param& pp, const HFr* const* instances,
                      const Fr* const* d_witness, Transcript& tr) {
  // single phase: synthesize(0, []) = d_witness
  HpPhases ph;
  ph.num_witness_polys = {pp.num_witness_polys};
  ph.num_challenges = {pp.num_challenges};
  ph.synthesize = [&](size_t, const std::vector<HFr>&) {
    return std::vector<const Fr*>(d_witness, d_witness + pp.num_witness_polys);
  };
  hyperplonk_prove_phases(c, pcs, pp, ph, instances, tr);
}

void hyperplonk_prove_phases(Ctx& c, const PcsProver& pcs, const lh_hp_param& pp, const HpPhases& ph,
                             const HFr* const* instances, Transcript& tr) {
  PhaseTimer pt(c);
  LH_REQUIRE(ph.num_witness_polys.size() == ph.num_challenges.size() && ph.synthesize, LH_ERR_ARG,
             "hyperplonk: phases are malformed");  // zip_eq, hyperplonk.rs:186-190
  const size_t nv = pp.num_vars, n = (size_t)1 << nv;
  LH_REQUIRE(nv >= 1 && nv < 32, LH_ERR_ARG, "hyperplonk: bad num_vars");
  // One proof over the 2^rho ranks of the ctx's communicator (dev.hpp Shard; lh_hyperplonk_prove_sharded): every poly the
  // caller hands over - preprocess, permutation, witness - is THIS RANK'S shard (n_loc rows) and so is every poly made
  // here.  What crosses ranks: partial commitments (one exchange per commit round), the zero-check's partial sums and its
  // residual tables (sum_check_loop), the rows of polys queried at a rotation (gathered once, expr.cpp), the per-row
  // products of the permutation argument (gathered once: the prefix product in hypercube order runs on every rank,
  // prover.rs:308-323), the Lasso lookups' exchanges (lasso.cpp) and the shared batch opening's (mkzg.cpp).  LogUp
  // lookups (the m poly is a global sort-merge join) are not sharded: such circuits run as replicas.
  const Shard sh(c);
  const bool shn = sh.on;
  const size_t n_loc = shn ? n >> sh.rho : n;
  if (shn) {
    LH_REQUIRE(sh.j >= 1 && sh.sharded(nv), LH_ERR_ARG, "sharded hyperplonk: the circuit is too small for this shard geometry");
    LH_REQUIRE(pp.num_lookups == 0, LH_ERR_ARG,
               "sharded hyperplonk: LogUp lookups do not shard (global sort-merge join); use Lasso lookups or replicas");
    LH_REQUIRE(pcs.sharded_ok, LH_ERR_ARG, "sharded hyperplonk: implemented for multilinear KZG");
  }
  auto local_row = [&](size_t g, size_t* loc) {  // global row -> this rank's local index (false: another rank's row)
    if (!shn) {
      *loc = g;
      return true;
    }
    if (((g >> sh.j) & (sh.R - 1)) != sh.rank) return false;
    *loc = ((g >> (sh.j + sh.rho)) << sh.j) | (g & (((size_t)1 << sh.j) - 1));
    return true;
  };
  ArenaScope scope(c.arena);
  EqHalfScope eq_scope(c);  // (lasso_argue shares eq tables of its points: arena memory of this scope)

  // BooleanHypercube order / nth_map (bh.rs:127-141), generated on the device: order[k] = x^(k-1) in GF(2^nv)
  uint32_t* d_order = c.arena.alloc_n<uint32_t>(n);
  uint32_t* d_nth = c.arena.alloc_n<uint32_t>(n);
  k_bh_order(c, nv, bh_primitive(nv), d_order, d_nth);

  pt.lap("hypercube order tables");
  // instances: hashed, then placed on rows bh[1], bh[2], .. (hyperplonk.rs:170-177,365-369; prover.rs:32-48)
  std::vector<const Fr*> polys;
  for (size_t i = 0; i < pp.num_instance_polys; i++) {
    const size_t cnt = pp.num_instances[i];
    LH_REQUIRE(cnt <= n, LH_ERR_ARG, "hyperplonk: too many instances");
    std::vector<uint32_t> rows;
    std::vector<HFr> vals;
    size_t b = 1;  // bh.iter(): 0, 1, x, x^2, ...
    for (size_t k = 0; k < cnt; k++) {
      tr.common_field_element(instances[i][k]);
      size_t loc;
      if (local_row(k + 1 < n ? b : 0, &loc)) {  // row_mapping = bh.iter().skip(1).chain([0]); sharded: this rank's rows
        rows.push_back((uint32_t)loc);
        vals.push_back(instances[i][k]);
      }
      b = bh_next(b, nv);
    }
    const size_t mine = rows.size();
    Fr* tab = c.arena.alloc_n<Fr>(n_loc);
    uint32_t* d_rows = c.arena.alloc_n<uint32_t>(std::max<size_t>(mine, 1));
    Fr* d_vals = c.arena.alloc_n<Fr>(std::max<size_t>(mine, 1));
    if (mine) {
      LH_HIP(hipMemcpyAsync(d_rows, rows.data(), mine * 4, hipMemcpyHostToDevice, c.stream));
      LH_HIP(hipMemcpyAsync(d_vals, vals.data(), mine * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
    }
    k_scatter_rows(c, d_rows, d_vals, mine, n_loc, tab);
    c.sync();
    polys.push_back(tab);
  }
  for (size_t i = 0; i < pp.num_preprocess_polys; i++) polys.push_back((const Fr*)pp.d_preprocess_polys[i]);

  pt.lap("instance polys");
  // Lasso lookups (oracle/pyref/hyperplonk.py LassoLookup): witness columns from the circuit's chunk polys, the
  // small-valued columns committed as u32 MSMs, framed with the identity mask
  struct LassoState {
    const lh_hp_lasso_lookup* lk;
    std::vector<uint32_t*> dims;
    LassoColumns cols;
    std::vector<const Fr*> dim_fr, rts_fr, E_fr, fcs_fr;  // fcs_fr: 2^nv entries (zero padded), read as l-variable tables too
  };
  std::vector<LassoState> lasso(pp.num_lasso_lookups);
  std::vector<MsmJob> lasso_jobs;  // the Lasso columns' commitments (u32 MSMs), filled by lasso_prepare
  auto lasso_prepare = [&] {
    if (!pp.num_lasso_lookups) return;
    LH_REQUIRE(pp.lasso_lookups != nullptr, LH_ERR_ARG, "hyperplonk: lasso_lookups is null");
    {
      // all Lasso commitments of a proof share ONE identity mask (a field element read as 63 bits, lasso.cpp)
      size_t total = 0;
      for (size_t k = 0; k < pp.num_lasso_lookups; k++)
        total += 2 * (size_t)pp.lasso_lookups[k].table.num_chunks + pp.lasso_lookups[k].table.num_memories;
      LH_REQUIRE(total <= LH_HP_LASSO_MAX_COMMITMENTS, LH_ERR_ARG,
                 "hyperplonk: the Lasso lookups of one circuit commit to more than 63 polys (sum of 2 * chunks + memories)");
    }
    const G1Affine* bases = shn ? pcs.shard_bases(nv) : pcs.commit_bases(nv);
    const G1Affine* bases_full = pcs.commit_bases(nv);
    std::vector<MsmJob>& jobs = lasso_jobs;
    uint32_t bad_input = 0;        // sharded: a rank that finds an invalid lookup must not leave its peers in a collective
    for (size_t k = 0; k < pp.num_lasso_lookups; k++) {
      LassoState& st = lasso[k];
      st.lk = &pp.lasso_lookups[k];
      const lh_lasso_table& tb = st.lk->table;
      lasso_check_table(tb);
      const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories, M = (size_t)1 << l;
      if (l > nv) throw Error(LH_ERR_INVALID_SNARK, "Lasso subtable larger than the circuit");
      if (shn) LH_REQUIRE(l <= sh.j + sh.rho, LH_ERR_ARG, "sharded hyperplonk: need shard_bit + rho >= chunk_bits (subtables replicated)");
      // (the same range on the verifier's side, verifier.cpp: preprocess and witness polys - committed, and known before
      // the lookup argument starts)
      LH_REQUIRE(st.lk->output_poly >= pp.num_instance_polys && st.lk->output_poly < polys.size(), LH_ERR_ARG,
                 "hyperplonk: lasso output poly out of range");
      st.dims.resize(cc);
      for (size_t j = 0; j < cc; j++) {
        LH_REQUIRE(st.lk->chunk_polys[j] >= pp.num_instance_polys && st.lk->chunk_polys[j] < polys.size(), LH_ERR_ARG,
                   "hyperplonk: lasso chunk poly out of range");
        st.dim_fr.push_back(polys[st.lk->chunk_polys[j]]);
        st.dims[j] = c.arena.alloc_n<uint32_t>(n_loc);
        if (!k_fr_to_index(c, st.dim_fr[j], n_loc, (uint32_t)l, st.dims[j])) {
          if (!shn) throw Error(LH_ERR_INVALID_SNARK, "Invalid lookup input");
          bad_input = 1;
        }
      }
      if (shn) {  // every rank learns whether any rank saw an index out of range BEFORE the counters' exchange starts
        std::vector<uint32_t> all(sh.R);
        comm_all_gather_host(c, &bad_input, all.data(), sizeof(uint32_t));
        for (uint32_t v : all)
          if (v) throw Error(LH_ERR_INVALID_SNARK, "Invalid lookup input");
      }
      Fr* a = nullptr;
      st.cols = lasso_witness_columns(c, tb, nv, st.dims.data(), &a);
      if (!k_fr_tables_equal(c, a, polys[st.lk->output_poly], n_loc)) bad_input = 1;
      if (shn) {
        std::vector<uint32_t> all(sh.R);
        comm_all_gather_host(c, &bad_input, all.data(), sizeof(uint32_t));
        for (uint32_t v : all) bad_input |= v;
      }
      if (bad_input) throw Error(LH_ERR_INVALID_SNARK, "Invalid lookup input");
      auto fr_view = [&](const uint32_t* src, size_t len) {
        Fr* d = c.arena.alloc_n<Fr>(n_loc);
        k_fr_from_u32(c, src, len, d);
        if (len < n_loc) LH_HIP(hipMemsetAsync(d + len, 0, (n_loc - len) * sizeof(Fr), c.stream));
        return (const Fr*)d;
      };
      for (size_t j = 0; j < cc; j++) st.rts_fr.push_back(fr_view(st.cols.rts[j], n_loc));
      for (size_t i = 0; i < alpha; i++) st.E_fr.push_back(fr_view(st.cols.E[i], n_loc));
      for (size_t j = 0; j < cc; j++) {
        // final_cts as a poly of nv variables is the 2^l counts followed by zeros; sharded: this rank's rows of THAT are
        // the slice [rank 2^shard_bit, (rank + 1) 2^shard_bit) of the (replicated) counts at its local rows [0, 2^shard_bit)
        if (!shn) {
          st.fcs_fr.push_back(fr_view(st.cols.fcs[j], M));
        } else {
          const size_t first = sh.rank << sh.j;
          st.fcs_fr.push_back(first < M ? fr_view(st.cols.fcs[j] + first, std::min(M - first, (size_t)1 << sh.j)) : fr_view(st.cols.fcs[j], 0));
        }
      }
      for (size_t j = 0; j < cc; j++) jobs.push_back(MsmJob{st.cols.rts[j], true, bases, n_loc});
      for (size_t i = 0; i < alpha; i++) jobs.push_back(MsmJob{st.cols.E[i], true, bases, n_loc});
      // (final_cts is replicated: every rank commits ITS range of the counts - lasso.cpp - and the parts are summed)
      const ReplicatedRange fc_range(sh, M);
      for (size_t j = 0; j < cc; j++) jobs.push_back(MsmJob{st.cols.fcs[j] + fc_range.first, true, bases_full + fc_range.first, fc_range.count});
    }
  };
  std::vector<HG1> lasso_comms;
  bool lasso_committed = false;
  // rounds 0..n (hyperplonk.rs:185-205): per phase synthesize from the challenges so far, commit, squeeze
  std::vector<HFr> challenges;
  for (size_t round = 0; round < ph.num_witness_polys.size(); round++) {
    std::vector<const Fr*> w = ph.synthesize(round, challenges);
    LH_REQUIRE(w.size() == ph.num_witness_polys[round], LH_ERR_ARG,
               "hyperplonk: synthesize returned the wrong number of witness polys");  // assert_eq hyperplonk.rs:198
    polys.insert(polys.end(), w.begin(), w.end());
    std::vector<HG1> comms;
    if (round + 1 == ph.num_witness_polys.size() && pp.num_lasso_lookups && pcs.commit_bases && (!shn || pcs.shard_bases)) {
      // The Lasso lookups' witness columns (access counters, subtable reads) are functions of the circuit's polys alone -
      // no challenge enters them - and every poly exists once the last phase is synthesized: their commitments join THIS
      // phase's witness commitments in one MSM batch (a batch's latency-bound tail - continuation levels, bucket
      // reduction, window sums, the host's combine: ~0.6 ms - is paid once instead of twice).  What the transcript sees
      // and when is unchanged: the Lasso commitments are written where oracle/pyref/hyperplonk.py writes them.
      lasso_prepare();
      std::vector<MsmJob> jobs;
      const G1Affine* wb = shn ? pcs.shard_bases(nv) : pcs.commit_bases(nv);
      for (const Fr* poly : w) jobs.push_back(MsmJob{poly, false, wb, n_loc});
      jobs.insert(jobs.end(), lasso_jobs.begin(), lasso_jobs.end());
      std::vector<HG1> out(jobs.size());
      msm_batch(c, jobs.data(), jobs.size(), (G1Affine*)out.data());
      if (shn) comm_sum_points(c, out.data(), out.size());  // the ranks' partial commitments -> their sums, one exchange
      comms.assign(out.begin(), out.begin() + w.size());
      lasso_comms.assign(out.begin() + w.size(), out.end());
      lasso_committed = true;
    } else {
      comms = pcs.batch_commit(w.data(), w.size(), nv);
    }
    tr.write_commitments(comms);
    std::vector<HFr> ch = tr.squeeze_challenges(ph.num_challenges[round]);
    challenges.insert(challenges.end(), ch.begin(), ch.end());
  }

  pt.lap("witness commitments");
  // round n: beta, lookup m polys
  HFr beta = tr.squeeze_challenge();
  size_t width = 0;
  for (size_t k = 0; k < pp.num_lookups; k++) width = std::max(width, pp.lookups[k].width);
  std::vector<HFr> betas(width);
  for (size_t i = 0; i < width; i++) betas[i] = i ? betas[i - 1] * beta : HFr::one();
  std::vector<Fr*> comp_in(pp.num_lookups), comp_tab(pp.num_lookups), m_polys(pp.num_lookups), h_polys(pp.num_lookups);
  for (size_t k = 0; k < pp.num_lookups; k++) {
    comp_in[k] = c.arena.alloc_n<Fr>(n);
    comp_tab[k] = c.arena.alloc_n<Fr>(n);
    m_polys[k] = c.arena.alloc_n<Fr>(n);
    h_polys[k] = c.arena.alloc_n<Fr>(n);
    compressed_poly(c, pp.lookups[k].inputs, pp.lookups[k].width, betas, polys, challenges.data(), challenges.size(), nv,
                    comp_in[k]);
    compressed_poly(c, pp.lookups[k].tables, pp.lookups[k].width, betas, polys, challenges.data(), challenges.size(), nv,
                    comp_tab[k]);
    if (!k_lookup_m(c, comp_in[k], comp_tab[k], n, m_polys[k]))
      throw Error(LH_ERR_INVALID_SNARK, "Invalid lookup input");  // prover.rs:176-178
  }
  {
    std::vector<const Fr*> mp(m_polys.begin(), m_polys.end());
    std::vector<HG1> comms = pcs.batch_commit(mp.data(), mp.size(), nv);
    tr.write_commitments(comms);
  }
  // Lasso lookups: their commitments - computed together with the last phase's witness commitments when the PCS allows
  // it (below) - enter the transcript here, framed with the identity mask (lasso.cpp)
  if (pp.num_lasso_lookups) {
    if (!lasso_committed) {
      lasso_prepare();
      lasso_comms.resize(lasso_jobs.size());
      msm_batch(c, lasso_jobs.data(), lasso_jobs.size(), (G1Affine*)lasso_comms.data());
      if (shn) comm_sum_points(c, lasso_comms.data(), lasso_comms.size());  // the ranks' partial commitments -> their sums
    }
    lasso_write_commitments(tr, lasso_comms);
  }

  pt.lap("lookup compressed + m + commit");
  // round n+1: gamma, lookup h polys and permutation z polys
  HFr gamma = tr.squeeze_challenge();
  for (size_t k = 0; k < pp.num_lookups; k++) k_lookup_h(c, comp_in[k], comp_tab[k], m_polys[k], dev(gamma), n, h_polys[k]);
  std::vector<Fr*> z_polys(pp.num_permutation_z_polys);
  for (auto& z : z_polys) z = c.arena.alloc_n<Fr>(n_loc);
  {
    std::vector<const Fr*> values(pp.num_permutation_polys), perms(pp.num_permutation_polys);
    for (size_t k = 0; k < pp.num_permutation_polys; k++) {
      LH_REQUIRE(pp.permutation_poly_index[k] < polys.size(), LH_ERR_ARG, "hyperplonk: permutation poly out of range");
      values[k] = polys[pp.permutation_poly_index[k]];
      perms[k] = (const Fr*)pp.d_permutation_polys[k];
    }
    if (!shn) {
      k_permutation_z(c, values.data(), perms.data(), pp.num_permutation_polys, pp.num_permutation_z_polys, nv, dev(beta),
                      dev(gamma), d_order, d_nth, z_polys.data());
    } else if (pp.num_permutation_polys) {
      // the per-row products on this rank's rows, ONE gather of them (num_z tables), then the hypercube-order prefix
      // product on every rank (prover.rs:308-323 is serial in the rows' order, which no index-bit split respects),
      // of which this rank keeps its rows
      ArenaScope tmp(c.arena);
      const size_t nz = pp.num_permutation_z_polys;
      Fr* block = c.arena.alloc_n<Fr>(nz * n_loc);
      std::vector<Fr*> prod_loc(nz), prod_full(nz);
      for (size_t k = 0; k < nz; k++) prod_loc[k] = block + k * n_loc, prod_full[k] = c.arena.alloc_n<Fr>(n);
      k_permutation_products(c, values.data(), perms.data(), pp.num_permutation_polys, nz, nv, dev(beta), dev(gamma), n_loc,
                             sh.j, sh.rho, sh.rank, prod_loc.data());
      comm_gather_tables(c, block, nz, n_loc, (size_t)1 << sh.j, prod_full.data());
      c.route.v[RouteStats::SHARD_EXCHANGES]++;
      std::vector<const Fr*> pf(prod_full.begin(), prod_full.end());
      k_permutation_z_from_products(c, pf.data(), nz, nv, d_order, d_nth, z_polys.data(), sh.j, sh.rho, sh.rank);
      c.sync();  // (the temporaries are released with this scope)
    }
  }
  {
    std::vector<const Fr*> hz(h_polys.begin(), h_polys.end());
    hz.insert(hz.end(), z_polys.begin(), z_polys.end());
    std::vector<HG1> comms = pcs.batch_commit(hz.data(), hz.size(), nv);
    tr.write_commitments(comms);
  }

  pt.lap("h, z polys + commit");
  // round n+2: alpha, y, zero-check
  HFr alpha = tr.squeeze_challenge();
  std::vector<HFr> y = tr.squeeze_challenges(nv);
  for (size_t k = 0; k < pp.num_permutation_polys; k++) polys.push_back((const Fr*)pp.d_permutation_polys[k]);
  for (auto p : m_polys) polys.push_back(p);
  for (auto p : h_polys) polys.push_back(p);
  for (auto p : z_polys) polys.push_back(p);
  challenges.push_back(beta);
  challenges.push_back(gamma);
  challenges.push_back(alpha);
  SumCheckResult sc = sum_check_prove_expr(c, nv, pp.expression, polys.data(), polys.size(), challenges.data(),
                                           challenges.size(), y.data(), 1, HFr::zero(), tr, shn);
  const std::vector<HFr>& x = sc.challenges;

  pt.lap("zero-check sum-check");
  // evaluations in pcs_query order (verifier.rs:147-180, prover.rs:388-406)
  std::set<std::pair<size_t, int>> query;
  for (size_t i = 0; i < pp.expression.num_nodes; i++) {
    const lh_expr_node& nd = pp.expression.nodes[i];
    if (nd.op == LH_EX_POLYNOMIAL && (size_t)nd.a >= pp.num_instance_polys) query.insert({(size_t)nd.a, nd.b});
  }
  std::set<int> rots;
  for (auto& q : query) rots.insert(q.second);
  std::map<int, size_t> point_off;
  std::vector<HFr> points;  // flattened, nv each
  size_t num_points = 0;
  std::map<int, std::vector<std::vector<HFr>>> rot_points;
  for (int r : rots) {
    point_off[r] = num_points;
    rot_points[r] = rotation_eval_points(x, r);
    for (auto& pt : rot_points[r]) points.insert(points.end(), pt.begin(), pt.end());
    num_points += rot_points[r].size();
  }
  std::vector<lh_evaluation> evals;
  std::vector<HFr> eval_values;
  for (auto& q : query) {
    std::vector<HFr> vals;
    if (q.second == 0) {
      vals.push_back(sc.evals[q.first]);
    } else {  // evaluate_for_rotation (multilinear.rs:191-264): the poly at the rotation's points
      for (auto& pt : rot_points[q.second]) vals.push_back(evaluate_polys(c, &polys[q.first], 1, nv, pt.data(), shn)[0]);
    }
    for (size_t k = 0; k < vals.size(); k++) {
      lh_evaluation e;
      e.poly = (uint32_t)q.first;
      e.point = (uint32_t)(point_off[q.second] + k);
      memcpy(&e.value, &vals[k], 32);
      evals.push_back(e);
      eval_values.push_back(vals[k]);
    }
  }
  tr.write_field_elements(eval_values);
  pt.lap("rotation evaluations");
  // Lasso lookups: the argument itself, then its claims join the one batch opening
  for (LassoState& st : lasso) {
    const lh_lasso_table& tb = st.lk->table;
    const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
    tr.common_field_element(HFr::from_u64(nv));
    tr.common_field_element(HFr::from_u64(l));
    tr.common_field_element(HFr::from_u64(cc));
    tr.common_field_element(HFr::from_u64(alpha));
    LassoClaims cl = lasso_argue(c, tb, nv, st.cols, st.dims.data(), polys[st.lk->output_poly], st.E_fr.data(), tr);
    const size_t base = polys.size(), p0 = num_points;
    for (const Fr* p : st.rts_fr) polys.push_back(p);
    for (const Fr* p : st.E_fr) polys.push_back(p);
    for (const Fr* p : st.fcs_fr) polys.push_back(p);
    for (const std::vector<HFr>* ptv : {&cl.r, &cl.r_z, &cl.r_N, &cl.r_M}) {
      points.insert(points.end(), ptv->begin(), ptv->end());
      points.insert(points.end(), nv - ptv->size(), HFr::zero());
    }
    num_points += 4;
    auto push = [&](size_t poly, size_t point, const HFr& val) {
      lh_evaluation e;
      e.poly = (uint32_t)poly, e.point = (uint32_t)point;
      memcpy(&e.value, &val, 32);
      evals.push_back(e);
    };
    push(st.lk->output_poly, p0, cl.v);
    for (size_t i = 0; i < alpha; i++) push(base + cc + i, p0 + 1, cl.e_rz[i]);
    for (size_t j = 0; j < cc; j++) push(st.lk->chunk_polys[j], p0 + 2, cl.ev_n[j]);
    for (size_t j = 0; j < cc; j++) push(base + j, p0 + 2, cl.ev_n[cc + j]);
    for (size_t i = 0; i < alpha; i++) push(base + cc + i, p0 + 2, cl.ev_n[2 * cc + i]);
    for (size_t j = 0; j < cc; j++) push(base + cc + alpha + j, p0 + 3, cl.ev_l[j]);
  }
  if (!lasso.empty()) pt.lap("lasso lookups");
  pcs.batch_open(nv, polys.data(), polys.size(), points.data(), num_points, evals.data(), evals.size(), tr);
  pt.lap("batch open");
  c.host_stamps_print();  // (LH_HOST_TRACE: the stamps of this prove's MSM batches and openings)
}

}  // namespace lh
