// Lasso lookup argument prover (Surge sum-check + offline memory checking by grand products).
// The reference snapshot holds no Lasso code (README.md:1-9 only); the protocol and
// transcript schedule are specified in oracle/pyref/lasso.py and reproduced here byte for byte.
#include <algorithm>
#include <chrono>
#include "host.hpp"

namespace lh {

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

LassoPcs lasso_mkzg_pcs(Ctx& c, const Srs& srs) {
  LassoPcs p;
  p.commit_bases = [&srs](size_t nv) { return srs.eq(nv); };
  p.shard_bases = [&c, &srs](size_t nv) { return srs_shard_level(c, srs, nv); };
  p.max_vars = srs.num_vars;
  p.batch_open = [&c, &srs](size_t nv, const Fr* const* polys, size_t np, const HFr* points, size_t npts,
                            const lh_evaluation* evals, size_t ne, Transcript& tr, const SmallPoly* small) {
    mkzg_batch_open(c, srs, nv, polys, np, points, npts, evals, ne, tr, small);
  };
  p.precommit = [&c, &srs](size_t nv, const SmallPoly* small, size_t np, const lh_evaluation* evals, size_t ne) {
    open_precommit_start(c, srs, nv, small, np, evals, ne);
  };
  return p;
}
LassoPcs lasso_zeromorph_pcs(Ctx& c, const USrs& srs, size_t poly_size) {
  LH_REQUIRE(poly_size >= 1 && poly_size <= srs.size, LH_ERR_INVALID_PCS_PARAM, "Too large poly_size to trim to");
  LassoPcs p;
  p.commit_bases = [&srs](size_t) { return (const G1Affine*)srs.d_powers; };
  p.max_vars = 0;
  while (((size_t)2 << p.max_vars) <= poly_size) p.max_vars++;
  p.batch_open = [&c, &srs, poly_size](size_t nv, const Fr* const* polys, size_t np, const HFr* points, size_t npts,
                                       const lh_evaluation* evals, size_t ne, Transcript& tr, const SmallPoly* small) {
    zeromorph_batch_open(c, srs, poly_size, nv, polys, np, points, npts, evals, ne, tr, small);
  };
  return p;
}

// Commitment framing of the Lasso argument (oracle/pyref/lasso.py write_commitments / read_commitments).  A committed
// column that is identically zero commits to the identity - the high-limb dim of a 32-bit range check whose values are
// all below 2^16, read_ts when the indices are pairwise distinct - and the reference's transcript cannot encode the
// identity (util/transcript.rs:172-179,216-219).  So: ONE field element whose bit i says that commitment i is the
// identity, then the other commitments in order; only calls every TranscriptWrite / TranscriptRead offers.
void lasso_write_commitments(Transcript& tr, const std::vector<HG1>& comms) {
  LH_REQUIRE(comms.size() < 64, LH_ERR_ARG, "lasso: too many commitments for the identity mask");
  uint64_t mask = 0;
  for (size_t i = 0; i < comms.size(); i++)
    if (comms[i].is_identity()) mask |= (uint64_t)1 << i;
  tr.write_field_element(HFr::from_u64(mask));
  for (const HG1& cm : comms)
    if (!cm.is_identity()) tr.write_commitment(cm);
}

std::vector<HG1> lasso_read_commitments(Transcript& tr, size_t count) {
  LH_REQUIRE(count < 64, LH_ERR_ARG, "lasso: too many commitments for the identity mask");
  const HFr m = tr.read_field_element();
  uint64_t canon[4];
  m.to_canonical(canon);
  if (canon[1] || canon[2] || canon[3] || (canon[0] >> count))
    throw Error(LH_ERR_INVALID_SNARK, "lasso: commitment mask out of range");
  std::vector<HG1> comms(count);
  for (size_t i = 0; i < count; i++)
    comms[i] = (canon[0] >> i) & 1 ? HG1{host::Fq::zero(), host::Fq::zero()} : tr.read_commitment();
  return comms;
}

void lasso_check_table(const lh_lasso_table& tb) {
  const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  LH_REQUIRE(cc >= 1 && cc <= LH_LASSO_MAX_CHUNKS, LH_ERR_ARG, "lasso: bad num_chunks");
  LH_REQUIRE(alpha >= 1 && alpha <= LH_LASSO_MAX_MEMORIES, LH_ERR_ARG, "lasso: bad num_memories");
  LH_REQUIRE(8 * alpha + 1 <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "lasso: too many memories for one GKR batch");
  LH_REQUIRE(l >= 1 && l < 31, LH_ERR_ARG, "lasso: need at least one variable");
  LH_REQUIRE(tb.num_terms >= 1 && tb.num_terms <= LH_LASSO_MAX_TERMS, LH_ERR_ARG, "lasso: bad g term count");
  for (size_t i = 0; i < alpha; i++) {
    LH_REQUIRE(tb.memory_chunk[i] < cc, LH_ERR_ARG, "lasso: memory chunk out of range");
    LH_REQUIRE(tb.memory_subtable[i] <= LH_SUBTABLE_XOR, LH_ERR_ARG, "lasso: unknown subtable");
    if (tb.memory_subtable[i] != LH_SUBTABLE_IDENTITY)
      LH_REQUIRE(l % 2 == 0, LH_ERR_ARG, "lasso: bitwise subtables need an even chunk_bits");
  }
  for (uint32_t m = 0; m < tb.num_terms; m++) {
    LH_REQUIRE(tb.g_num_factors[m] >= 1 && tb.g_num_factors[m] <= LH_SC_MAX_FACTORS, LH_ERR_ARG, "lasso: bad g term");
    for (int k = 0; k < tb.g_num_factors[m]; k++)
      LH_REQUIRE(tb.g_factor[m][k] < alpha, LH_ERR_ARG, "lasso: g factor out of range");
  }
}

// Access counters of a sharded proof: read_ts[k] = number of earlier lookups - in the GLOBAL lookup order - of the same
// address, final_cts[a] = number of lookups of a.  rts[j]: this rank's shard (2^(n - rho)), fcs[j]: replicated (2^l).
// No rank ever holds a whole column: the lookups are repartitioned by address (owner = address mod R), the owner sorts
// the keys it receives and ranks every lookup inside its address run, the ranks travel back the same way.  All chunk
// columns go together: per PROOF two personalised exchanges of ~cc 2^(n - rho) entries per rank (4 B out - kernels_poly.hip:
// the key needs neither the sender nor the low index bits -, 4 B back), one host all-gather of the segment boundaries and
// one all-gather of the cc 2^l / R counts of every owner.  Work and traffic per rank shrink with R (a column that hits one
// address only degenerates to the single-GPU sort on that address's owner).
static void lasso_counters_sharded(Ctx& c, const Shard& sh, const uint32_t* const* d_dims_local, size_t cc, size_t n, size_t l,
                                   uint32_t* const* rts, uint32_t* const* fcs) {
  const size_t R = sh.R, me = sh.rank, M = (size_t)1 << l, NL = (size_t)1 << (n - sh.rho);
  const size_t m_loc = std::max<size_t>(M >> sh.rho, 1), R1 = R + 1;
  unsigned a_bits = 0;
  while (((size_t)1 << a_bits) < m_loc) a_bits++;
  const unsigned hi_bits = (unsigned)(n - sh.rho - sh.j);  // the local index bits above the shard bits
  ArenaScope scope(c.arena);
  uint32_t* sidx_all = c.arena.alloc_n<uint32_t>(cc * NL);
  uint32_t* send_all = c.arena.alloc_n<uint32_t>(cc * NL);
  std::vector<uint32_t*> sidx(cc), send(cc);
  for (size_t q = 0; q < cc; q++) sidx[q] = sidx_all + q * NL, send[q] = send_all + q * NL;
  // everybody's segment boundaries: starts[s][q][o] = where, in rank s's send buffer of column q, the lookups for owner o
  // begin - and everybody's verdict on its own shard: an index out of range on ANY rank fails the prove on EVERY rank,
  // after the exchange (a rank that threw on its own would leave its peers waiting in this collective)
  const size_t W = cc * R1 + 1;
  std::vector<uint32_t> start(W), starts(W * R);
  bool bad = false;
  k_cs_partition(c, d_dims_local, cc, NL, M, (unsigned)sh.rho, (unsigned)sh.j, hi_bits, sidx.data(), send.data(), start.data(), &bad);
  start[cc * R1] = bad ? 1u : 0u;
  comm_all_gather_host(c, start.data(), starts.data(), W * sizeof(uint32_t));
  for (size_t s = 0; s < R; s++)
    LH_REQUIRE(!starts[s * W + cc * R1], LH_ERR_ARG, "lasso: chunk index out of range (>= 2^chunk_bits)");
  auto seg = [&](size_t s, size_t q, size_t o) { return (size_t)(starts[s * W + q * R1 + o + 1] - starts[s * W + q * R1 + o]); };
  std::vector<size_t> s_off(cc * R), s_cnt(cc * R), r_off(cc * R), r_cnt(cc * R), peer_off(cc * R), back_peer_off(cc * R, 0);
  std::vector<size_t> recv_total(cc, 0);
  size_t recv_max = 0, recv_sum = 0;
  for (size_t q = 0; q < cc; q++) {
    for (size_t p = 0; p < R; p++) {
      s_off[q * R + p] = start[q * R1 + p], s_cnt[q * R + p] = seg(me, q, p);
      r_off[q * R + p] = recv_total[q], r_cnt[q * R + p] = seg(p, q, me), peer_off[q * R + p] = starts[p * W + q * R1 + me];
      recv_total[q] += r_cnt[q * R + p];
      // (where, in owner p's return buffer, the segment for this rank begins: the lookups of ranks 0..me-1 come first)
      for (size_t s = 0; s < me; s++) back_peer_off[q * R + p] += seg(s, q, p);
    }
    recv_sum += recv_total[q];
    for (size_t o = 0; o < R; o++) {  // (the largest receive buffer of any owner and column: the stride of the way back)
      size_t t = 0;
      for (size_t s = 0; s < R; s++) t += seg(s, q, o);
      recv_max = std::max(recv_max, t);
    }
  }
  recv_max = std::max<size_t>(recv_max, 1);
  uint32_t* recv_all = c.arena.alloc_n<uint32_t>(std::max<size_t>(recv_sum, 1));
  uint32_t* ret_all = c.arena.alloc_n<uint32_t>(cc * recv_max);
  uint32_t* back_all = c.arena.alloc_n<uint32_t>(cc * NL);
  uint32_t* counts = c.arena.alloc_n<uint32_t>(cc * m_loc);
  std::vector<uint32_t*> recv(cc), ret(cc), back(cc);
  {
    size_t off = 0;
    for (size_t q = 0; q < cc; q++) recv[q] = recv_all + off, off += recv_total[q], ret[q] = ret_all + q * recv_max, back[q] = back_all + q * NL;
  }
  std::vector<const void*> csend(send.begin(), send.end()), cret(ret.begin(), ret.end());
  std::vector<void*> vrecv(recv.begin(), recv.end()), vback(back.begin(), back.end());
  comm_all_to_all_multi(c, cc, csend.data(), s_off.data(), s_cnt.data(), vrecv.data(), r_off.data(), r_cnt.data(), peer_off.data(),
                        send_all, NL, sizeof(uint32_t));
  // the owner's side: rank inside the address run, per-address totals
  std::vector<const uint32_t*> crecv(recv.begin(), recv.end());
  k_cs_rank(c, crecv.data(), recv_total.data(), cc, hi_bits, a_bits, m_loc, ret.data(), counts);
  // the ranks go back along the same segments (the send side of the way back is this rank's receive layout)
  comm_all_to_all_multi(c, cc, cret.data(), r_off.data(), r_cnt.data(), vback.data(), s_off.data(), s_cnt.data(),
                        back_peer_off.data(), ret_all, recv_max, sizeof(uint32_t));
  std::vector<const uint32_t*> cback(back.begin(), back.end()), csidx(sidx.begin(), sidx.end());
  k_cs_scatter(c, cback.data(), csidx.data(), cc, NL, rts);
  uint32_t* all_counts = c.arena.alloc_n<uint32_t>(cc * m_loc * R);
  comm_all_gather_dev(c, counts, all_counts, cc * m_loc * sizeof(uint32_t));
  k_cs_final(c, all_counts, cc, M, (unsigned)sh.rho, m_loc, fcs);
  c.sync();  // (the scope's buffers are released)
  c.route.v[RouteStats::SHARD_EXCHANGES] += 3;
}

// witness: access counters, subtable reads and (optionally) the lookup outputs a = g(E); arena memory of the caller's scope.
// Inside a sharded proof d_dims are this rank's shards of the lookup columns and so are read_ts / E / a; final_cts (2^l
// entries, l <= shard_bit + rho) is replicated.
LassoColumns lasso_witness_columns(Ctx& c, const lh_lasso_table& tb, size_t n, const uint32_t* const* d_dims, Fr** a_out,
                                   uint32_t** a_small_out, bool keep_sorted) {
  const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  const Shard sh(c);
  const size_t N = (size_t)1 << (n - sh.rho), M = (size_t)1 << l;  // N: entries of this rank's columns
  LassoColumns w;
  w.rts.resize(cc), w.fcs.resize(cc), w.E.resize(alpha);
  for (size_t j = 0; j < cc; j++) {
    w.rts[j] = c.arena.alloc_n<uint32_t>(N);
    w.fcs[j] = c.arena.alloc_n<uint32_t>(M);
  }
  // LH_SHARDED_COUNTERS_MIN_R=1 (tests): a world of one takes the repartitioned counters too (the personalised exchange
  // then has one peer, itself: the transport's point-to-point path on a one-GPU box)
  static const size_t counters_min_r = [] {
    const char* e = getenv("LH_SHARDED_COUNTERS_MIN_R");
    return e && atoi(e) >= 1 ? (size_t)atoi(e) : (size_t)2;
  }();
  keep_sorted = keep_sorted && !(sh.on && sh.R >= counters_min_r);
  if (sh.on && sh.R >= counters_min_r) {
    lasso_counters_sharded(c, sh, d_dims, cc, n, l, w.rts.data(), w.fcs.data());
  } else {
    if (keep_sorted)
      for (size_t j = 0; j < cc; j++) {
        w.dim_sorted.push_back(c.arena.alloc_n<uint32_t>(N));
        w.dim_index.push_back(c.arena.alloc_n<uint32_t>(N));
      }
    // all chunk columns in one launch set, one bad-index readback (kernels_poly.hip)
    k_lasso_counters(c, d_dims, cc, N, M, w.rts.data(), w.fcs.data(), keep_sorted ? w.dim_sorted.data() : nullptr,
                     keep_sorted ? w.dim_index.data() : nullptr);
  }
  LassoG g;
  memset(&g, 0, sizeof(g));
  for (size_t i = 0; i < alpha; i++) {
    if (tb.memory_subtable[i] == LH_SUBTABLE_IDENTITY) {
      // E = dim entry by entry: the same column (read-only everywhere; the batch opening merges columns it is handed twice)
      w.E[i] = const_cast<uint32_t*>(d_dims[tb.memory_chunk[i]]);
    } else {
      w.E[i] = c.arena.alloc_n<uint32_t>(N);
      k_lasso_subtable_read(c, (int)tb.memory_subtable[i], (uint32_t)l, d_dims[tb.memory_chunk[i]], N, w.E[i]);
    }
    g.e[i] = w.E[i];
  }
  if (a_small_out) {
    // a 32-bit output column when g = sum coeff_t E_t with small coefficients and the largest value fits 32 bits
    *a_small_out = nullptr;
    LassoGSmall gs;
    memset(&gs, 0, sizeof(gs));
    gs.num_terms = tb.num_terms;
    bool ok = true;
    uint64_t max_val = 0;
    for (uint32_t m = 0; m < tb.num_terms && ok; m++) {
      HFr co;
      memcpy(&co, &tb.g_coeff[m], 32);
      uint64_t canon[4];
      co.to_canonical(canon);
      const size_t i = tb.g_factor[m][0];
      const size_t ebits = tb.memory_subtable[i] == LH_SUBTABLE_IDENTITY ? l : l / 2;
      ok = tb.g_num_factors[m] == 1 && !canon[1] && !canon[2] && !canon[3] && canon[0] <= 0xffffffffull && ebits <= 32;
      if (!ok) break;
      max_val += canon[0] * (((uint64_t)1 << ebits) - 1);
      ok = max_val <= 0xffffffffull;
      gs.coeff[m] = (uint32_t)canon[0];
      gs.fac[m] = (uint8_t)i;
    }
    if (ok) {
      for (size_t i = 0; i < alpha; i++) gs.e[i] = w.E[i];
      *a_small_out = c.arena.alloc_n<uint32_t>(N);
      k_lasso_output_small(c, gs, N, *a_small_out);
      if (a_out) *a_out = nullptr;
      return w;
    }
  }
  if (a_out) {
    g.num_terms = tb.num_terms;
    for (uint32_t m = 0; m < tb.num_terms; m++) {
      memcpy(&g.coeff[m], &tb.g_coeff[m], 32);
      g.nfac[m] = tb.g_num_factors[m];
      for (int k = 0; k < LH_SC_MAX_FACTORS; k++) g.fac[m][k] = tb.g_factor[m][k];
    }
    *a_out = c.arena.alloc_n<Fr>(N);
    k_lasso_output(c, g, N, *a_out);
  }
  return w;
}

// Steps 2-7 of the argument (oracle/pyref/lasso.py argue): Surge sum-check, memory-checking grand products,
// evaluations.  The Fr tables hold at least 2^n (fcs_fr: 2^l) entries; `lap` (optional) receives phase boundaries.
LassoClaims lasso_argue(Ctx& c, const lh_lasso_table& tb, size_t n, const LassoColumns& w, const uint32_t* const* d_dims,
                        const Fr* a, const Fr* const* E_fr, Transcript& tr, const std::function<void(int)>& lap,
                        const uint32_t* a_small) {
  const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  // inside a sharded proof (dev.hpp Shard) every n-variable column is this rank's shard (N entries); sums over a column -
  // evaluations, round messages - are partial sums added over the ranks; the 2^l-entry subtable side is replicated
  const Shard sh(c);
  const bool shn = sh.on;
  if (shn) LH_REQUIRE(sh.sharded(n) && l <= sh.j + sh.rho, LH_ERR_ARG, "lasso: shard geometry does not fit the table");
  const size_t N = (size_t)1 << (n - sh.rho), M = (size_t)1 << l;
  auto sum_ranks = [&](HFr* v, size_t count) {
    if (shn) comm_sum_fr(c, v, count);
  };
  LassoClaims cl;
  // ---- 2-4: Surge primary sum-check
  cl.r = tr.squeeze_challenges(n);
  c.host_stamp("argue:squeezed");
  bool linear_g = true;
  for (uint32_t m = 0; m < tb.num_terms; m++) linear_g = linear_g && tb.g_num_factors[m] == 1;
  Fr a_sums[4];  // (k_inner_products_small_quads: the claim's two halves - and rounds 0 and 1 of the Surge sum-check over a)
  bool have_a_sums = false;
  if (a_small && n >= 2) {
    // against the eq table of r[1..] (half the entries), which the Surge sum-check and the batch opening at r use as well
    const Fr* eq_half = eq_half_get(c, cl.r.data(), n, shn);
    if (!a && linear_g && !shn && n >= 3) {
      k_inner_products_small_quads(c, a_small, eq_half, N / 4, a_sums);
      const HFr r0 = cl.r[0];
      cl.v = (HFr::one() - r0) * hst(a_sums[0]) + r0 * hst(a_sums[1]);
      have_a_sums = true;
    } else {
      k_inner_products_small_half(c, &a_small, 1, eq_half, N / 2, dev(cl.r[0]), (Fr*)&cl.v);
    }
    sum_ranks(&cl.v, 1);
  } else if (a_small) {
    ArenaScope scope(c.arena);
    Fr* eq = c.arena.alloc_n<Fr>(N);
    k_eq_xy(c, (const Fr*)cl.r.data(), n, eq);
    k_inner_products_small(c, &a_small, 1, eq, N, (Fr*)&cl.v);
  } else {
    cl.v = evaluate_polys(c, &a, 1, n, cl.r.data(), shn)[0];
  }
  tr.write_field_element(cl.v);
  SumCheckResult sc;
  if (linear_g) {
    // g linear (range / AND / XOR): the summand eq * sum_m coeff_m E_m IS eq * a entry by entry, and binding is linear
    // too, so the round messages of the sum-check over the alpha subtable-read columns are those of the sum-check over
    // the single output column a.  One table instead of alpha in every round; the evaluations E_i(r_z) the transcript
    // wants next are inner products of the 32-bit columns with eq(r_z).
    {
      ArenaScope scope(c.arena);
      const Fr* a_tab = a;
      struct HintGuard {  // the hint is this sum-check's alone, whatever way it ends
        Ctx& c;
        ~HintGuard() { c.sc_u32 = Ctx::ScU32(); }
      } hint_guard{c};
      if (!a_tab) {
        // no field-element view of the output column: the sum-check runs its first three rounds from the 32-bit column where
        // it can (a_sums above are rounds 0 and 1) and fills this table itself where it cannot (dev.hpp Ctx::sc_u32)
        a_tab = c.arena.alloc_n<Fr>(N);
        c.sc_u32.col = a_small;
        c.sc_u32.have_sums = have_a_sums;
        if (have_a_sums) c.sc_u32.odd = a_sums[1], c.sc_u32.s2 = a_sums[2], c.sc_u32.s3 = a_sums[3];
      }
      lh_sop one_term;
      memset(&one_term, 0, sizeof(one_term));
      one_term.global_eq = 0;
      one_term.num_terms = 1;
      const HFr one = HFr::one();
      memcpy(&one_term.coeff[0], &one, 32);
      one_term.num_factors[0] = 1;
      one_term.factor[0][0] = 0;
      sc = sum_check_prove(c, LH_SC_EVALUATIONS, n, one_term, &a_tab, 1, cl.r.data(), 1, cl.v, tr, true, nullptr, shn);
    }
    std::vector<const uint32_t*> cols(w.E.begin(), w.E.end());
    sc.evals.assign(alpha, HFr::zero());
    if (n >= 2) {  // (the table of r_z[1..] stays for the batch opening at r_z)
      const Fr* eq_half = eq_half_get(c, sc.challenges.data(), n, shn);
      k_inner_products_small_half(c, cols.data(), alpha, eq_half, N / 2, dev(sc.challenges[0]), (Fr*)sc.evals.data());
      sum_ranks(sc.evals.data(), alpha);
    } else {
      ArenaScope scope(c.arena);
      Fr* eq = c.arena.alloc_n<Fr>(N);
      k_eq_xy(c, (const Fr*)sc.challenges.data(), n, eq);
      k_inner_products_small(c, cols.data(), alpha, eq, N, (Fr*)sc.evals.data());
    }
  } else {
    lh_sop surge;
    memset(&surge, 0, sizeof(surge));
    surge.global_eq = 0;
    surge.num_terms = tb.num_terms;
    for (uint32_t m = 0; m < tb.num_terms; m++) {
      surge.coeff[m] = tb.g_coeff[m];
      surge.num_factors[m] = tb.g_num_factors[m];
      for (int k = 0; k < LH_SC_MAX_FACTORS; k++) surge.factor[m][k] = tb.g_factor[m][k];
    }
    sc = sum_check_prove(c, LH_SC_EVALUATIONS, n, surge, E_fr, alpha, cl.r.data(), 1, cl.v, tr, true, nullptr, shn);
  }
  cl.r_z = sc.challenges;
  cl.e_rz = sc.evals;
  tr.write_field_elements(sc.evals);
  if (lap) lap(2);

  // ---- 5/6: memory-checking fingerprints and product trees
  HFr gamma = tr.squeeze_challenge();
  HFr tau = tr.squeeze_challenge();
  HFr gamma2 = gamma * gamma;
  {
    ArenaScope scope(c.arena);
    std::vector<const Fr*> leaves(4 * alpha), level_up(4 * alpha, nullptr);
    std::vector<size_t> depths(4 * alpha);
    std::vector<uint8_t> plus_one(4 * alpha, 0);
    // write leaf = read leaf + 1: when the n-variable trees are alone at their leaf layer (n > l) that layer runs over the
    // read-set tables only (prove_grand_product: plus_one) and the write-set leaves are never stored
    // (sharded: the level above the leaves must itself be held in shards, else it is exchanged by the tree builder)
    const bool fused_up = n >= 12 && (!shn || sh.sharded(n - 1)), ws_implicit = fused_up && n > l;
    for (size_t i = 0; i < alpha; i++) {
      size_t j = tb.memory_chunk[i];
      Fr* rs = c.arena.alloc_n<Fr>(N);
      Fr* ws = ws_implicit ? nullptr : c.arena.alloc_n<Fr>(N);
      Fr* in = c.arena.alloc_n<Fr>(M);
      Fr* fi = c.arena.alloc_n<Fr>(M);
      if (fused_up) {  // the level above the leaves comes with them (no tree_up pass over the largest level)
        Fr* rs_up = c.arena.alloc_n<Fr>(N / 2);
        Fr* ws_up = c.arena.alloc_n<Fr>(N / 2);
        k_lasso_rw_leaves_up(c, d_dims[j], w.E[i], w.rts[j], N, dev(gamma), dev(gamma2), dev(tau), rs, ws, rs_up, ws_up);
        level_up[2 * i] = rs_up, level_up[2 * i + 1] = ws_up;
      } else {
        k_lasso_rw_leaves(c, d_dims[j], w.E[i], w.rts[j], N, dev(gamma), dev(gamma2), dev(tau), rs, ws);
      }
      k_lasso_if_leaves(c, (int)tb.memory_subtable[i], (uint32_t)l, w.fcs[j], M, dev(gamma), dev(gamma2), dev(tau), in, fi);
      leaves[2 * i] = rs, leaves[2 * i + 1] = ws;
      plus_one[2 * i + 1] = 1;
      leaves[2 * alpha + 2 * i] = in, leaves[2 * alpha + 2 * i + 1] = fi;
      depths[2 * i] = depths[2 * i + 1] = n;
      depths[2 * alpha + 2 * i] = depths[2 * alpha + 2 * i + 1] = l;
    }
    if (lap) lap(3);
    GrandProductResult gp = prove_grand_product(c, 4 * alpha, leaves.data(), depths.data(), tr, level_up.data(), plus_one.data());
    cl.r_N = gp.points[0];
    cl.r_M = gp.points[2 * alpha];
  }
  if (lap) lap(4);

  // ---- 7: evaluations at r_N / r_M: dim | read_ts | E, then final_cts - straight from the u32 columns (no
  // field-element views: 4 bytes read and 8 multiply-adds per entry); at r_N against the eq table of r_N[1..], which the
  // batch opening at r_N uses as well
  const Fr* eq_half_n = n >= 2 ? eq_half_get(c, cl.r_N.data(), n, shn) : nullptr;
  {
    ArenaScope scope(c.arena);
    std::vector<const uint32_t*> at_n;
    for (size_t j = 0; j < cc; j++) at_n.push_back(d_dims[j]);
    for (size_t j = 0; j < cc; j++) at_n.push_back(w.rts[j]);
    for (size_t i = 0; i < alpha; i++) at_n.push_back(w.E[i]);
    cl.ev_n.resize(at_n.size());
    Fr* eq = c.arena.alloc_n<Fr>(eq_half_n ? M : std::max(N, M));
    if (eq_half_n) {
      k_inner_products_small_half(c, at_n.data(), at_n.size(), eq_half_n, N / 2, dev(cl.r_N[0]), (Fr*)cl.ev_n.data());
      sum_ranks(cl.ev_n.data(), cl.ev_n.size());
    } else {
      k_eq_xy(c, (const Fr*)cl.r_N.data(), n, eq);
      k_inner_products_small(c, at_n.data(), at_n.size(), eq, N, (Fr*)cl.ev_n.data());
    }
    std::vector<const uint32_t*> at_l(w.fcs.begin(), w.fcs.end());
    cl.ev_l.resize(cc);
    k_eq_xy(c, (const Fr*)cl.r_M.data(), l, eq);
    k_inner_products_small(c, at_l.data(), cc, eq, M, (Fr*)cl.ev_l.data());
  }
  tr.write_field_elements(cl.ev_n);
  tr.write_field_elements(cl.ev_l);
  if (lap) lap(5);
  return cl;
}

void lasso_prove(Ctx& c, const LassoPcs& pcs, const lh_lasso_table& tb, size_t n, const uint32_t* const* d_dims,
                 Transcript& tr) {
  lasso_check_table(tb);
  const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  LH_REQUIRE(n >= 1 && n < 31, LH_ERR_ARG, "lasso: need at least one variable");
  if (n > pcs.max_vars || l > pcs.max_vars)
    throw Error(LH_ERR_INVALID_PCS_PARAM, "Too many variates of poly to commit");
  // every committed poly is zero-padded to nv = max(n, l) variables (spec step 1)
  // Inside a sharded proof (dev.hpp Shard, lasso_prove_sharded below) d_dims are this rank's shards of the lookup columns
  // and every n-variable column below is a shard of N = 2^(n - rho) entries; what crosses ranks: the access counters'
  // exchange, partial commitments, partial sums per round, residual tables (sumcheck.cpp, gkr.cpp, mkzg.cpp).  World of one: nothing.
  const Shard sh(c);
  const bool shn = sh.on;
  if (shn) {
    LH_REQUIRE(pcs.shard_bases != nullptr, LH_ERR_ARG, "sharded prove: implemented for multilinear KZG");
    LH_REQUIRE(sh.j >= 1 && l <= sh.j + sh.rho, LH_ERR_ARG,
               "sharded prove: need shard_bit + rho >= chunk_bits (subtables replicated)");
    LH_REQUIRE(sh.sharded(n) && n >= l, LH_ERR_ARG, "sharded prove: 2^num_vars lookups are too few to shard");
  }
  const size_t nv = std::max(n, l), NV = (size_t)1 << (nv - sh.rho);
  const size_t N = (size_t)1 << (n - sh.rho), M = (size_t)1 << l;
  // phase boundaries are EVENTS on the prover's stream (a hipStreamSynchronize at each of them drained the queue seven
  // times per proof for the sake of a number nobody reads in production): lh_lasso_last_timing turns them into phase
  // times when asked (Ctx::phase_times_resolve); lasso_ms[8], the total, is the host's wall clock
  const double t0 = now_ms();
  double* ph = c.lasso_ms;
  for (int k = 0; k < 8; k++)
    if (!c.phase_ev[k]) LH_HIP(hipEventCreate(&c.phase_ev[k]));
  c.phase_ev_pending = false;
  LH_HIP(hipEventRecord(c.phase_ev[0], c.stream));
  c.comm_phase = 0;
  struct PhaseGuard {  // (collectives issued after the prove - or after it failed - count as outside)
    Ctx& c;
    ~PhaseGuard() { c.comm_phase = 7; }
  } phase_guard{c};
  auto lap = [&](int idx) {
    LH_HIP(hipEventRecord(c.phase_ev[idx + 1], c.stream));
    c.comm_phase = idx + 1;
  };

  ArenaScope scope(c.arena);
  EqHalfScope eq_scope(c);  // (the shared eq tables are arena memory of this scope)
  c.route = RouteStats();
  std::vector<uint32_t> count_ors(cc, 0);  // OR of every final_cts column (bounds its read_ts column) when computed
  // ---- witness: counters, subtable reads, lookup outputs
  Fr* a = nullptr;
  uint32_t* a_small = nullptr;
  // (the sorted dim columns only pay off where the MSM sorts slab by slab: msm.hip LH_MSM_SLAB_LOG)
  LassoColumns w = lasso_witness_columns(c, tb, n, d_dims, &a, &a_small, (int)n >= msm_slab_log());
  std::vector<uint32_t*>&rts = w.rts, &fcs = w.fcs, &E = w.E;
  lap(0);
  // ---- 0/1: domain separation + commitments (one batched MSM)
  tr.common_field_element(HFr::from_u64(n));
  tr.common_field_element(HFr::from_u64(l));
  tr.common_field_element(HFr::from_u64(cc));
  tr.common_field_element(HFr::from_u64(alpha));
  {
    // zero padding adds nothing to an MSM: commit the unpadded columns against the first entries of the bases.
    // Commitments are linear, so columns that are linear in others need no MSM of their own (same group elements,
    // same proof bytes): E_i = dim_j for an identity subtable, and a = sum_m coeff_m E_{f(m)} when g is linear
    // (range / AND / XOR tables): 4 of 9 column MSMs instead of 9 for the range check.
    // (sharded: this rank's share of the bases; every MSM below is the commitment of a shard - the chunk-split-then-sum
    // of util/arithmetic/msm.rs:101-114 with the shards as chunks - and the partial commitments are added at the end)
    const G1Affine* bases = shn ? pcs.shard_bases(nv) : pcs.commit_bases(nv);
    bool linear_g = true;
    for (uint32_t m = 0; m < tb.num_terms; m++) linear_g = linear_g && tb.g_num_factors[m] == 1;
    const size_t total = 1 + 3 * cc + alpha;
    std::vector<MsmJob> jobs;
    std::vector<size_t> slot;  // position of each job's result in `comms`
    auto add_job = [&](size_t pos, const void* col, bool u32, size_t len) {
      jobs.push_back(MsmJob{col, u32, bases, len});
      slot.push_back(pos);
    };
    if (!linear_g) add_job(0, a, false, N);
    std::vector<size_t> dim_job(cc);
    for (size_t j = 0; j < cc; j++) {
      dim_job[j] = jobs.size(), add_job(1 + j, d_dims[j], true, N);
      jobs.back().known_bits = (uint32_t)l;  // (the access counters rejected any index >= 2^l)
      if (!w.dim_sorted.empty()) {
        jobs.back().sorted_scalars = w.dim_sorted[j], jobs.back().sorted_index = w.dim_index[j];
        c.route.v[RouteStats::SORTED_REUSE]++;
      }
    }
    // read_ts columns are small (a cell's access count): two of them share one pass over the points when their bit
    // lengths allow (MsmJob::pack_shift; the access counts bound the timestamps)
    std::vector<HG1> second(cc);
    std::vector<size_t> second_of;  // chunks whose commitment comes back through `second`
    {
      std::vector<uint32_t>& ors = count_ors;
      std::vector<const uint32_t*> cols(fcs.begin(), fcs.end());
      const bool pack_on = c.opt.lasso_pack_ts != 0;  // 0: one MSM pass per read_ts column (A/B measurements)
      if (pack_on && cc >= 2 && N >= ((size_t)1 << 12)) k_or_u32(c, cols.data(), cc, M, ors.data());
      auto bits_of = [](uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; };
      for (size_t j = 0; j < cc; j++) {
        const uint32_t b0 = std::max(bits_of(ors[j]), 4u), b1 = j + 1 < cc ? bits_of(ors[j + 1]) : 0;
        if (ors[j] && j + 1 < cc && ors[j + 1] && b0 + b1 <= MSM_PACK_MAX_BITS &&
            N >= ((size_t)MSM_PACK_MIN_POINTS_PER_BUCKET << (b0 + b1))) {  // (worth it while the points outnumber the buckets)
          uint32_t* packed = c.arena.alloc_n<uint32_t>(N);
          k_pack_u32(c, rts[j], rts[j + 1], b0, N, packed);
          add_job(1 + cc + j, packed, true, N);
          jobs.back().known_bits = b0 + b1;
          jobs.back().pack_shift = b0;
          jobs.back().out_second = (G1Affine*)&second[j + 1];
          second_of.push_back(j + 1);
          c.route.v[RouteStats::PACKED_TS]++;
          j++;
        } else {
          add_job(1 + cc + j, rts[j], true, N);
          jobs.back().known_bits = bits_of(ors[j]);  // (0 when the counts were not looked at: measured then)
        }
      }
    }
    // E_i = T[dim_j]: commit(E_i) = sum_d T[d] * B_d over the BUCKET sums B_d of dim_j's commitment - no second pass
    // over the N points (msm.hip, MsmJob::derived_parent; falls back to an ordinary column when the window shape of
    // dim_j does not allow it)
    SubtableOrders orders(c, l);
    for (size_t i = 0; i < alpha; i++)
      if (tb.memory_subtable[i] != LH_SUBTABLE_IDENTITY) {
        add_job(1 + 2 * cc + i, E[i], true, N);
        if (l <= 20) {
          MsmJob& jb = jobs.back();
          jb.derived_parent = (int)dim_job[tb.memory_chunk[i]];
          orders.get((int)tb.memory_subtable[i], &jb.d_table, &jb.d_order);
          jb.table_in_bits = (uint32_t)l, jb.table_out_bits = (uint32_t)(l / 2);
          c.route.v[RouteStats::DERIVED]++;
        }
        jobs.back().known_bits = (uint32_t)(l / 2);  // AND / XOR of two (l/2)-bit halves
      }
    // final_cts (2^l entries) is replicated - but an MSM is additive over point ranges (the chunk-then-sum of
    // util/arithmetic/msm.rs:101-114): rank s commits the counts [s 2^l / R, (s + 1) 2^l / R) against the same range of
    // the level's first bases, and the partial commitments join the exchange below - no rank repeats another's additions
    const ReplicatedRange fc_range(sh, M);
    for (size_t j = 0; j < cc; j++) {
      add_job(1 + 2 * cc + alpha + j, fcs[j] + fc_range.first, true, fc_range.count);
      jobs.back().bases = pcs.commit_bases(nv) + fc_range.first;
      jobs.back().known_bits = count_ors[j] ? 32u - (uint32_t)__builtin_clz(count_ors[j]) : 0u;
    }
    std::vector<HG1> part(jobs.size()), comms(total);
    msm_batch(c, jobs.data(), jobs.size(), (G1Affine*)part.data());
    c.host_stamp("commit:msm_done");
    if (shn) {
      // partial commitments of the shards (the second outputs of packed jobs included) -> their sums, one exchange
      std::vector<HG1> sums(part);
      for (size_t j : second_of) sums.push_back(second[j]);
      comm_sum_points(c, sums.data(), sums.size());
      for (size_t k = 0; k < part.size(); k++) part[k] = sums[k];
      for (size_t q = 0; q < second_of.size(); q++) second[second_of[q]] = sums[part.size() + q];
    }
    c.host_stamp("commit:summed");
    for (size_t k = 0; k < jobs.size(); k++) comms[slot[k]] = part[k];
    for (size_t j : second_of) comms[1 + cc + j] = second[j];
    for (size_t i = 0; i < alpha; i++)
      if (tb.memory_subtable[i] == LH_SUBTABLE_IDENTITY) comms[1 + 2 * cc + i] = comms[1 + tb.memory_chunk[i]];
    if (linear_g) {
      std::vector<host::G1Xyzz> terms(tb.num_terms);
      // (the coefficients of the range / AND / XOR tables are powers of two below 2^64: a few dozen doublings each, done
      // here - waking the host pool for them cost more than they do; wide coefficients go to the pool)
      bool small_coeffs = true;
      for (uint32_t m = 0; m < tb.num_terms && small_coeffs; m++) {
        HFr co;
        memcpy(&co, &tb.g_coeff[m], 32);
        uint64_t canon[4];
        co.to_canonical(canon);
        small_coeffs = !canon[1] && !canon[2] && !canon[3];
      }
      auto term = [&](size_t m) {
        HFr co;
        memcpy(&co, &tb.g_coeff[m], 32);
        terms[m] = host::g1_mul(host::g1_from_affine(comms[1 + 2 * cc + tb.g_factor[m][0]]), co);
      };
      if (small_coeffs)
        for (size_t m = 0; m < tb.num_terms; m++) term(m);
      else
        host_parallel_for(tb.num_terms, term);
      host::G1Xyzz acc = host::G1Xyzz::identity();
      for (auto& t : terms) acc = host::g1_add(acc, t);
      comms[0] = host::g1_to_affine(acc);
    }
    c.host_stamp("commit:linear");
    lasso_write_commitments(tr, comms);
    c.host_stamp("commit:written");
  }
  lap(1);

  // ---- field-element views only where a sum-check needs tables (E for Surge when g is not linear); dim / read_ts /
  // final_cts (and E under a linear g) stay u32 all the way: fingerprints, evaluations and the batch opening's merge
  // read the 4-byte columns
  const size_t num_n = 1 + 2 * cc + alpha;
  std::vector<const Fr*> polys_n(num_n, nullptr), polys_l(cc, nullptr);
  std::vector<SmallPoly> small(num_n + cc);
  auto fr_view = [&](const uint32_t* src, size_t len) {
    Fr* d = c.arena.alloc_n<Fr>(NV);
    k_fr_from_u32(c, src, len, d);
    if (len < NV) LH_HIP(hipMemsetAsync(d + len, 0, (NV - len) * sizeof(Fr), c.stream));
    return d;
  };
  SmallLinear a_linear;  // a = sum_m coeff_m E_{f(m)} entry by entry (linear g)
  if (a_small) {
    for (uint32_t m = 0; m < tb.num_terms; m++) {
      HFr co;
      memcpy(&co, &tb.g_coeff[m], 32);
      a_linear.poly.push_back(1 + 2 * cc + tb.g_factor[m][0]);
      a_linear.coeff.push_back(co);
    }
    small[0] = SmallPoly{a_small, N, 0, &a_linear};
  } else if (N < NV) {  // l > n: the output column needs the padding too
    Fr* ap = c.arena.alloc_n<Fr>(NV);
    LH_HIP(hipMemcpyAsync(ap, a, N * sizeof(Fr), hipMemcpyDeviceToDevice, c.stream));
    LH_HIP(hipMemsetAsync(ap + N, 0, (NV - N) * sizeof(Fr), c.stream));
    polys_n[0] = ap;
  } else {
    polys_n[0] = a;
  }
  for (size_t j = 0; j < cc; j++) {
    small[1 + j] = SmallPoly{d_dims[j], N, (uint32_t)l};
    small[1 + cc + j] = SmallPoly{rts[j], N, count_ors[j] ? 32u - (uint32_t)__builtin_clz(count_ors[j]) : 0u};
    small[num_n + j] = SmallPoly{fcs[j], M, 0};
    if (shn) {
      // the zero-padded n-variable final_cts column lives in the index range [0, 2^l), l <= shard_bit + rho: this rank's
      // shard of it is the slice [rank * 2^shard_bit, (rank + 1) * 2^shard_bit) at its local indices [0, 2^shard_bit)
      const size_t first = sh.rank << sh.j;
      small[num_n + j] = first < M ? SmallPoly{fcs[j] + first, std::min(M - first, (size_t)1 << sh.j), 0} : SmallPoly{fcs[j], 0, 0};
    }
  }
  bool linear_surge = true;  // (lasso_argue: the Surge sum-check then runs over the output column alone)
  for (uint32_t m = 0; m < tb.num_terms; m++) linear_surge = linear_surge && tb.g_num_factors[m] == 1;
  for (size_t i = 0; i < alpha; i++) {
    if (!linear_surge) polys_n[1 + 2 * cc + i] = fr_view(E[i], N);
    small[1 + 2 * cc + i] = SmallPoly{E[i], N, (uint32_t)(tb.memory_subtable[i] == LH_SUBTABLE_IDENTITY ? l : l / 2)};
  }
  const Fr* const* E_fr = polys_n.data() + 1 + 2 * cc;

  // (poly, point) pairs of the batch opening at the end (step 8): r, r_z, r_N, r_M are points 0..3
  std::vector<lh_evaluation> evs;
  {
    auto push = [&](size_t poly, size_t point) {
      lh_evaluation e;
      memset(&e, 0, sizeof(e));
      e.poly = (uint32_t)poly;
      e.point = (uint32_t)point;
      evs.push_back(e);
    };
    push(0, 0);
    for (size_t i = 0; i < alpha; i++) push(1 + 2 * cc + i, 1);
    for (size_t j = 0; j < cc; j++) push(1 + j, 2);
    for (size_t j = 0; j < cc; j++) push(1 + cc + j, 2);
    for (size_t i = 0; i < alpha; i++) push(1 + 2 * cc + i, 2);
    for (size_t j = 0; j < cc; j++) push(num_n + j, 3);
  }
  // the opening's column-wise commitments depend on the witness columns alone: on the helper ctx from here on, beside
  // the sum-checks (Options::open_precommit)
  struct PrecommitGuard {  // (a prove that fails on the way drops what was started)
    Ctx& c;
    ~PrecommitGuard() {
      c.gkr_hook = nullptr;
      open_precommit_cancel(c);
    }
  } precommit_guard{c};
  // (started when the memory-checking argument has built its product trees - Ctx::gkr_hook: beside the Surge rounds and the
  // tree kernels, which stream at the HBM bound, the helper's sorts only get in the way: tree_up 1.2 -> 5.0 ms per proof)
  if (pcs.precommit)
    c.gkr_hook = [&] { pcs.precommit(nv, small.data(), small.size(), evs.data(), evs.size()); };

  // ---- 2-7: Surge, memory checking, evaluations
  c.host_stamp("argue:start");
  LassoClaims cl = lasso_argue(c, tb, n, w, d_dims, polys_n[0], E_fr, tr, lap, a_small);
  c.host_stamp("argue:end");
  const std::vector<HFr>&r = cl.r, &r_z = cl.r_z, &r_N = cl.r_N, &r_M = cl.r_M, &ev_n = cl.ev_n, &ev_l = cl.ev_l;
  const HFr& v = cl.v;

  // ---- 8: ONE batch opening of all committed polys (nv variables) at r, r_z, r_N, r_M (zero-padded)
  {
    std::vector<HFr> points(4 * nv, HFr::zero());
    std::copy(r.begin(), r.end(), points.begin());
    std::copy(r_z.begin(), r_z.end(), points.begin() + nv);
    std::copy(r_N.begin(), r_N.end(), points.begin() + 2 * nv);
    std::copy(r_M.begin(), r_M.end(), points.begin() + 3 * nv);
    size_t next = 0;
    auto value = [&](const HFr& val) { memcpy(&evs[next++].value, &val, 32); };  // (in the order the pairs were listed)
    value(v);
    for (size_t i = 0; i < alpha; i++) value(cl.e_rz[i]);
    for (size_t j = 0; j < cc; j++) value(ev_n[j]);
    for (size_t j = 0; j < cc; j++) value(ev_n[cc + j]);
    for (size_t i = 0; i < alpha; i++) value(ev_n[2 * cc + i]);
    for (size_t j = 0; j < cc; j++) value(ev_l[j]);
    LH_REQUIRE(next == evs.size(), LH_ERR_ARG, "lasso: evaluation list out of step");
    std::vector<const Fr*> all(polys_n);
    all.insert(all.end(), polys_l.begin(), polys_l.end());
    pcs.batch_open(nv, all.data(), all.size(), points.data(), 4, evs.data(), evs.size(), tr, small.data());
  }
  lap(6);
  c.host_stamp("open:end");
  c.host_stamps_print();
  ph[7] = 0;
  ph[8] = now_ms() - t0;
  c.phase_ev_pending = true;
}

// ONE proof over the 2^rho ranks of the ctx's communicator: the same prover, with every table a shard
// (dev.hpp Shard).  Same transcript, same proof bytes on every rank as lasso_prove on one GPU.
void lasso_prove_sharded(Ctx& c, const Srs& srs, const lh_lasso_table& tb, size_t n, const uint32_t* const* d_dims_local,
                         Transcript& tr) {
  LH_REQUIRE(c.has_comm, LH_ERR_ARG, "lasso_prove_sharded: no communicator attached");
  const size_t R = (size_t)c.comm.size;
  LH_REQUIRE(R >= 1 && (R & (R - 1)) == 0, LH_ERR_ARG, "sharded prove: the number of ranks must be a power of two");
  struct Active {
    Ctx& c;
    explicit Active(Ctx& c_) : c(c_) { c.shard_active = true; }
    ~Active() { c.shard_active = false; }
  } active(c);
  lasso_prove(c, lasso_mkzg_pcs(c, srs), tb, n, d_dims_local, tr);
}

}  // namespace lh
