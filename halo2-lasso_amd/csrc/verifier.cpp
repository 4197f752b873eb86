// The verify half of the trait surface, host only (no GPU work: a verifier reads a few KB of proof and does
// O(num_vars) field operations and pairings).
//   SumCheck::verify                       piop/sum_check/classic.rs:175-193,242-272, eval.rs:34-57, coeff.rs:19-39
//   MultilinearKzg::{verify, batch_verify} pcs/multilinear/kzg.rs:330-375, pcs/multilinear.rs:237-276
//   HyperPlonk::verify                     backend/hyperplonk.rs:293-362, hyperplonk/verifier.rs:39-182,
//                                          piop/sum_check.rs:60-125, poly/multilinear.rs:433-475
//   Lasso verify                           oracle/pyref/lasso.py (the build's own protocol; no reference code)
#include <map>
#include <set>
#include <string>
#include "host.hpp"
#include "expr.hpp"
#include "pairing.hpp"

namespace lh {

using host::G2Affine;

struct VerifierParams {
  HG1 g1;
  G2Affine g2;
  std::vector<G2Affine> ss;
};

static G2Affine g2_from_c(const lh_g2& p) {
  G2Affine a;
  static_assert(sizeof(lh_g2) == sizeof(G2Affine), "lh_g2 layout");
  memcpy(&a, &p, sizeof(a));
  return a;
}
static lh_g2 g2_to_c(const G2Affine& a) {
  lh_g2 p;
  memcpy(&p, &a, sizeof(p));
  return p;
}

VerifierParams* mkzg_vp_setup(const HFr* ss, size_t num_vars) {
  auto* vp = new VerifierParams();
  vp->g1 = HG1{host::Fq::from_u64(1), host::Fq::from_u64(2)};
  vp->g2 = host::g2_generator();
  host::G2Xyzz g = host::g2_from_affine(vp->g2);
  for (size_t i = 0; i < num_vars; i++) vp->ss.push_back(host::g2_to_affine(host::g2_mul(g, ss[i])));
  return vp;
}
VerifierParams* mkzg_vp_new(const lh_g1& g1, const lh_g2& g2, const lh_g2* ss, size_t num_vars) {
  auto* vp = new VerifierParams();
  memcpy(&vp->g1, &g1, sizeof(HG1));
  vp->g2 = g2_from_c(g2);
  bool ok = host::g2_is_on_curve(vp->g2);
  for (size_t i = 0; i < num_vars; i++) {
    vp->ss.push_back(g2_from_c(ss[i]));
    ok = ok && host::g2_is_on_curve(vp->ss.back());
  }
  if (!ok) {
    delete vp;
    throw Error(LH_ERR_SERIALIZATION, "verifier params: G2 point not on the curve");
  }
  return vp;
}
void mkzg_vp_export(const VerifierParams& vp, lh_g1* g1, lh_g2* g2, lh_g2* ss) {
  if (g1) memcpy(g1, &vp.g1, sizeof(HG1));
  if (g2) *g2 = g2_to_c(vp.g2);
  if (ss)
    for (size_t i = 0; i < vp.ss.size(); i++) ss[i] = g2_to_c(vp.ss[i]);
}
size_t mkzg_vp_num_vars(const VerifierParams& vp) { return vp.ss.size(); }
void mkzg_vp_free(VerifierParams* vp) { delete vp; }

bool pairing_check(const lh_g1* ps, const lh_g2* qs, size_t n) {
  std::vector<std::pair<HG1, G2Affine>> pairs(n);
  for (size_t i = 0; i < n; i++) {
    memcpy(&pairs[i].first, &ps[i], sizeof(HG1));
    pairs[i].second = g2_from_c(qs[i]);
    LH_REQUIRE(host::g2_is_on_curve(pairs[i].second), LH_ERR_ARG, "pairing: G2 point not on the curve");
  }
  return host::pairings_product_is_identity(pairs);
}

// ------------------------------------------------------------------ SumCheck::verify
std::pair<HFr, std::vector<HFr>> sum_check_verify(int prover_kind, size_t num_vars, size_t degree, const HFr& sum,
                                                  Transcript& tr) {
  LH_REQUIRE(prover_kind == LH_SC_EVALUATIONS || prover_kind == LH_SC_COEFFICIENTS, LH_ERR_ARG, "bad prover kind");
  std::vector<std::vector<HFr>> msgs;
  std::vector<HFr> challenges;
  for (size_t r = 0; r < num_vars; r++) {  // classic.rs:250-258: all messages first, then the consistency pass
    msgs.push_back(tr.read_field_elements(degree + 1));
    challenges.push_back(tr.squeeze_challenge());
  }
  HFr s = sum;
  for (size_t r = 0; r < num_vars; r++) {
    const std::vector<HFr>& m = msgs[r];
    HFr msg_sum;
    if (prover_kind == LH_SC_EVALUATIONS) {
      msg_sum = m[0] + (m.size() > 1 ? m[1] : HFr::zero());
    } else {
      msg_sum = m[0].dbl();
      for (size_t i = 1; i < m.size(); i++) msg_sum += m[i];
    }
    if (s != msg_sum)
      throw Error(LH_ERR_INVALID_SUMCHECK,
                  r == 0 ? std::string("Expect sum to match the first round message")
                         : "Consistency failure at round " + std::to_string(r));
    s = prover_kind == LH_SC_EVALUATIONS ? interpolate_evals(m, challenges[r]) : horner(m, challenges[r]);
  }
  return {s, challenges};
}

// ------------------------------------------------------------------ MultilinearKzg::verify / batch_verify
void mkzg_verify(const VerifierParams& vp, const HG1& comm, const HFr* point, size_t num_vars, const HFr& eval,
                 Transcript& tr) {
  if (num_vars > vp.ss.size())
    throw Error(LH_ERR_INVALID_PCS_PARAM, "Too many variates of poly to verify (param supports variates up to " +
                                              std::to_string(vp.ss.size()) + " but got " + std::to_string(num_vars) + ")");
  std::vector<HG1> quotients = tr.read_commitments(num_vars);
  std::vector<std::pair<HG1, G2Affine>> pairs;
  host::G1Xyzz lhs0 = host::g1_add(host::g1_from_affine(comm),
                                   host::g1_mul(host::g1_from_affine(HG1{vp.g1.x, -vp.g1.y}), eval));
  pairs.push_back({host::g1_to_affine(lhs0), host::g2_neg(vp.g2)});
  host::G2Xyzz g2 = host::g2_from_affine(vp.g2);
  for (size_t i = 0; i < num_vars; i++) {
    host::G2Xyzz rhs = host::g2_add(host::g2_from_affine(vp.ss[i]), host::g2_mul(g2, -point[i]));
    pairs.push_back({quotients[i], host::g2_to_affine(rhs)});
  }
  if (!host::pairings_product_is_identity(pairs)) throw Error(LH_ERR_INVALID_PCS_OPEN, "Invalid multilinear KZG open");
}

// additive::batch_verify (pcs/multilinear.rs:237-276), generic over the PCS
static void additive_batch_verify(size_t num_vars, const HG1* comms, size_t num_comms, const HFr* points,
                                  size_t num_points, const lh_evaluation* evals, size_t num_evals, Transcript& tr,
                                  const std::function<void(const HG1&, const HFr*, const HFr&)>& verify) {
  for (size_t i = 0; i < num_evals; i++)
    LH_REQUIRE(evals[i].poly < num_comms && evals[i].point < num_points, LH_ERR_ARG, "batch verify: bad evaluation");
  size_t ell = 0;
  while (((size_t)1 << ell) < num_evals) ell++;
  std::vector<HFr> t = tr.squeeze_challenges(ell);
  std::vector<HFr> eq_xt = host_eq_xy(t);
  HFr tilde_gs_sum = HFr::zero();
  for (size_t i = 0; i < num_evals; i++) {
    HFr v;
    memcpy(&v, &evals[i].value, 32);
    tilde_gs_sum += v * eq_xt[i];
  }
  auto res = sum_check_verify(LH_SC_COEFFICIENTS, num_vars, 2, tilde_gs_sum, tr);
  const std::vector<HFr>& x = res.second;
  std::vector<HFr> eq_evals(num_points);
  for (size_t j = 0; j < num_points; j++) eq_evals[j] = host_eq_xy_eval(x.data(), points + j * num_vars, num_vars);
  host::G1Xyzz acc = host::G1Xyzz::identity();  // sum_with_scalar (kzg.rs:138-149)
  for (size_t i = 0; i < num_evals; i++)
    acc = host::g1_add(acc, host::g1_mul(host::g1_from_affine(comms[evals[i].poly]), eq_evals[evals[i].point] * eq_xt[i]));
  verify(host::g1_to_affine(acc), x.data(), res.first);
}

void mkzg_batch_verify(const VerifierParams& vp, size_t num_vars, const HG1* comms, size_t num_comms,
                       const HFr* points, size_t num_points, const lh_evaluation* evals, size_t num_evals,
                       Transcript& tr) {
  if (num_vars > vp.ss.size())
    throw Error(LH_ERR_INVALID_PCS_PARAM,
                "Too many variates of poly to batch verify (param supports variates up to " +
                    std::to_string(vp.ss.size()) + " but got " + std::to_string(num_vars) + ")");
  additive_batch_verify(num_vars, comms, num_comms, points, num_points, evals, num_evals, tr,
                        [&](const HG1& comm, const HFr* x, const HFr& eval) { mkzg_verify(vp, comm, x, num_vars, eval, tr); });
}

// ------------------------------------------------------------------ Zeromorph::verify (zeromorph.rs:215-256)
struct ZmVerifierParams {
  HG1 g1;
  G2Affine g2, s_g2, s_offset_g2;
};
ZmVerifierParams* zeromorph_vp_setup(const HFr& s, size_t param_size, size_t poly_size) {
  LH_REQUIRE(poly_size >= 1 && poly_size <= param_size, LH_ERR_INVALID_PCS_PARAM, "Too large poly_size to trim to");
  auto* vp = new ZmVerifierParams();
  vp->g1 = HG1{host::Fq::from_u64(1), host::Fq::from_u64(2)};
  vp->g2 = host::g2_generator();
  host::G2Xyzz g = host::g2_from_affine(vp->g2);
  vp->s_g2 = host::g2_to_affine(host::g2_mul(g, s));
  HFr so = HFr::one();  // s^(param_size - poly_size)
  for (size_t i = 0; i < param_size - poly_size; i++) so *= s;
  vp->s_offset_g2 = host::g2_to_affine(host::g2_mul(g, so));
  return vp;
}
ZmVerifierParams* zeromorph_vp_new(const lh_g1& g1, const lh_g2& g2, const lh_g2& s_g2, const lh_g2& s_offset_g2) {
  auto* vp = new ZmVerifierParams();
  memcpy(&vp->g1, &g1, sizeof(HG1));
  vp->g2 = g2_from_c(g2), vp->s_g2 = g2_from_c(s_g2), vp->s_offset_g2 = g2_from_c(s_offset_g2);
  if (!host::g2_is_on_curve(vp->g2) || !host::g2_is_on_curve(vp->s_g2) || !host::g2_is_on_curve(vp->s_offset_g2)) {
    delete vp;
    throw Error(LH_ERR_SERIALIZATION, "verifier params: G2 point not on the curve");
  }
  return vp;
}
void zeromorph_vp_export(const ZmVerifierParams& vp, lh_g1* g1, lh_g2* g2, lh_g2* s_g2, lh_g2* s_offset_g2) {
  if (g1) memcpy(g1, &vp.g1, sizeof(HG1));
  if (g2) *g2 = g2_to_c(vp.g2);
  if (s_g2) *s_g2 = g2_to_c(vp.s_g2);
  if (s_offset_g2) *s_offset_g2 = g2_to_c(vp.s_offset_g2);
}
void zeromorph_vp_free(ZmVerifierParams* vp) { delete vp; }

void zeromorph_verify(const ZmVerifierParams& vp, const HG1& comm, const HFr* point, size_t num_vars, const HFr& eval,
                      Transcript& tr) {
  std::vector<HG1> q_comms = tr.read_commitments(num_vars);
  const HFr y = tr.squeeze_challenge();
  const HG1 q_hat_comm = tr.read_commitment();
  const HFr x = tr.squeeze_challenge(), z = tr.squeeze_challenge();
  auto sc = zeromorph_scalars(y, x, z, point, num_vars);
  host::G1Xyzz c = host::g1_from_affine(q_hat_comm);
  c = host::g1_add(c, host::g1_mul(host::g1_from_affine(comm), z));
  c = host::g1_add(c, host::g1_mul(host::g1_from_affine(vp.g1), sc.first * eval));
  for (size_t k = 0; k < num_vars; k++) c = host::g1_add(c, host::g1_mul(host::g1_from_affine(q_comms[k]), sc.second[k]));
  const HG1 pi = tr.read_commitment();
  host::G2Xyzz rhs = host::g2_add(host::g2_from_affine(vp.s_g2), host::g2_mul(host::g2_from_affine(vp.g2), -x));
  if (!host::pairings_product_is_identity({{host::g1_to_affine(c), host::g2_neg(vp.s_offset_g2)},
                                            {pi, host::g2_to_affine(rhs)}}))
    throw Error(LH_ERR_INVALID_PCS_OPEN, "Invalid Zeromorph KZG open");
}
void zeromorph_batch_verify(const ZmVerifierParams& vp, size_t num_vars, const HG1* comms, size_t num_comms,
                            const HFr* points, size_t num_points, const lh_evaluation* evals, size_t num_evals,
                            Transcript& tr) {
  additive_batch_verify(num_vars, comms, num_comms, points, num_points, evals, num_evals, tr,
                        [&](const HG1& comm, const HFr* x, const HFr& eval) { zeromorph_verify(vp, comm, x, num_vars, eval, tr); });
}

// ------------------------------------------------------------------ expressions on the host
static void check_expr(const lh_expr& e) {
  LH_REQUIRE(e.nodes && e.num_nodes, LH_ERR_ARG, "empty expression");
  for (size_t i = 0; i < e.num_nodes; i++) {
    const lh_expr_node& nd = e.nodes[i];
    LH_REQUIRE(nd.op <= LH_EX_SCALED, LH_ERR_ARG, "expression: bad op");
    if (nd.op >= LH_EX_NEGATED) LH_REQUIRE(nd.a >= 0 && (size_t)nd.a < i, LH_ERR_ARG, "expression: bad child");
    if (nd.op == LH_EX_SUM || nd.op == LH_EX_PRODUCT)
      LH_REQUIRE(nd.b >= 0 && (size_t)nd.b < i, LH_ERR_ARG, "expression: bad child");
  }
}
static size_t expr_degree(const lh_expr& e) {  // expression.rs:171-182
  std::vector<size_t> d(e.num_nodes);
  for (size_t i = 0; i < e.num_nodes; i++) {
    const lh_expr_node& nd = e.nodes[i];
    switch (nd.op) {
      case LH_EX_CONSTANT:
      case LH_EX_CHALLENGE: d[i] = 0; break;
      case LH_EX_NEGATED:
      case LH_EX_SCALED: d[i] = d[nd.a]; break;
      case LH_EX_SUM: d[i] = std::max(d[nd.a], d[nd.b]); break;
      case LH_EX_PRODUCT: d[i] = d[nd.a] + d[nd.b]; break;
      default: d[i] = 1;
    }
  }
  return d.back();
}
struct EvalCtx {
  HFr identity;
  std::map<int, HFr> lagrange;
  std::vector<HFr> eq_xys;
  std::map<std::pair<size_t, int>, HFr> evals;
  std::vector<HFr> challenges;
};
static HFr eval_expr(const lh_expr& e, const EvalCtx& cx) {  // piop/sum_check.rs:60-98
  std::vector<HFr> v(e.num_nodes);
  for (size_t i = 0; i < e.num_nodes; i++) {
    const lh_expr_node& nd = e.nodes[i];
    HFr sc;
    memcpy(&sc, &nd.scalar, 32);
    switch (nd.op) {
      case LH_EX_CONSTANT: v[i] = sc; break;
      case LH_EX_IDENTITY: v[i] = cx.identity; break;
      case LH_EX_LAGRANGE: v[i] = cx.lagrange.at(nd.a); break;
      case LH_EX_EQ_XY:
        LH_REQUIRE(nd.a >= 0 && (size_t)nd.a < cx.eq_xys.size(), LH_ERR_ARG, "expression: eq_xy index");
        v[i] = cx.eq_xys[nd.a];
        break;
      case LH_EX_POLYNOMIAL: {
        auto it = cx.evals.find({(size_t)nd.a, nd.b});
        LH_REQUIRE(it != cx.evals.end(), LH_ERR_ARG, "expression: query without evaluation");
        v[i] = it->second;
        break;
      }
      case LH_EX_CHALLENGE:
        LH_REQUIRE(nd.a >= 0 && (size_t)nd.a < cx.challenges.size(), LH_ERR_ARG, "expression: challenge index");
        v[i] = cx.challenges[nd.a];
        break;
      case LH_EX_NEGATED: v[i] = -v[nd.a]; break;
      case LH_EX_SUM: v[i] = v[nd.a] + v[nd.b]; break;
      case LH_EX_PRODUCT: v[i] = v[nd.a] * v[nd.b]; break;
      default: v[i] = v[nd.a] * sc;
    }
  }
  return v.back();
}

static HFr identity_eval(const std::vector<HFr>& x) {  // sum_check.rs:123-125
  HFr acc = HFr::zero(), p = HFr::one();
  for (auto& xi : x) {
    acc += xi * p;
    p = p.dbl();
  }
  return acc;
}
static HFr lagrange_eval(const std::vector<HFr>& x, size_t b) {  // sum_check.rs:100-111
  HFr acc = HFr::one();
  for (size_t i = 0; i < x.size(); i++) acc *= ((b >> i) & 1) ? x[i] : HFr::one() - x[i];
  return acc;
}
// the row of BooleanHypercube::iter() at position i.rem_euclid(2^num_vars)
static size_t bh_row(size_t num_vars, long long i) {
  const long long n = (long long)1 << num_vars;
  long long m = i % n;
  if (m < 0) m += n;
  return bh_nth(num_vars, (size_t)m);
}

// poly/multilinear.rs:435-475,528-549
static std::vector<size_t> coeff_pattern(bool next, size_t num_vars, size_t distance) {
  const size_t rem = next ? (size_t)bh_primitive(num_vars) - ((size_t)1 << num_vars) : (size_t)bh_x_inv(num_vars) << distance;
  std::vector<size_t> pat((size_t)1 << (distance - 1), 0);
  for (size_t depth = 0; depth + 1 < distance; depth++) {
    size_t step = (size_t)1 << (distance - depth - 1);
    for (size_t e = 0; e < pat.size(); e += step) {
      size_t o = e + step / 2;
      size_t rot = next ? pat[e] << 1 : pat[e] >> 1;
      pat[o] = rot ^ rem;
      pat[e] = rot;
    }
  }
  return pat;
}
static HFr rotation_eval(const std::vector<HFr>& x, int rotation, const std::vector<HFr>& evals_for_rotation) {
  if (rotation == 0) return evals_for_rotation[0];
  const size_t n = x.size(), distance = (size_t)std::abs(rotation);
  LH_REQUIRE(distance <= n && evals_for_rotation.size() == ((size_t)1 << distance), LH_ERR_ARG, "rotation_eval: shape");
  std::vector<size_t> pat = coeff_pattern(rotation > 0, n, distance), nths(distance);
  std::vector<HFr> xs(distance);
  for (size_t i = 0; i < distance; i++) {
    if (rotation < 0) {
      nths[i] = distance - i;
      xs[i] = x[distance - 1 - i];
    } else {
      nths[i] = n - 1 + i;
      xs[i] = x[n - distance + i];
    }
  }
  std::vector<HFr> evals = evals_for_rotation;
  for (size_t idx = 0; idx < distance; idx++) {
    std::vector<HFr> next(evals.size() / 2);
    for (size_t k = 0; k < next.size(); k++) {
      const bool flip = (pat[k << idx] >> nths[idx]) & 1;
      const HFr &e0 = evals[2 * k], &e1 = evals[2 * k + 1];
      next[k] = flip ? (e0 - e1) * xs[idx] + e1 : (e1 - e0) * xs[idx] + e0;
    }
    evals.swap(next);
  }
  return evals[0];
}

// ------------------------------------------------------------------ HyperPlonk::verify
static void lasso_verify_check_table(const lh_lasso_table& tb, size_t n);

void hyperplonk_verify(const PcsBatchVerify& batch_verify, const lh_hp_vparam& vp, const HFr* const* instances,
                       Transcript& tr) {
  hyperplonk_verify_phases(batch_verify, vp, {vp.num_witness_polys}, {vp.num_challenges}, instances, tr);
}

void hyperplonk_verify_phases(const PcsBatchVerify& batch_verify, const lh_hp_vparam& vp,
                              const std::vector<size_t>& phase_witness_polys, const std::vector<size_t>& phase_challenges,
                              const HFr* const* instances, Transcript& tr) {
  const size_t nv = vp.num_vars;
  LH_REQUIRE(phase_witness_polys.size() == phase_challenges.size(), LH_ERR_ARG, "hyperplonk: phases are malformed");
  {
    size_t w = 0, ch = 0;
    for (size_t v : phase_witness_polys) w += v;
    for (size_t v : phase_challenges) ch += v;
    LH_REQUIRE(w == vp.num_witness_polys && ch == vp.num_challenges, LH_ERR_ARG,
               "hyperplonk: phases do not add up to num_witness_polys / num_challenges");
  }
  LH_REQUIRE(nv >= 1 && nv < 32, LH_ERR_ARG, "hyperplonk: bad num_vars");
  check_expr(vp.expression);
  for (size_t i = 0; i < vp.num_instance_polys; i++)
    for (size_t k = 0; k < vp.num_instances[i]; k++) tr.common_field_element(instances[i][k]);
  // rounds 0..n (hyperplonk.rs:309-316): per phase read the witness commitments, squeeze the phase's challenges
  std::vector<HG1> witness_comms;
  std::vector<HFr> challenges;
  for (size_t r = 0; r < phase_witness_polys.size(); r++) {
    std::vector<HG1> cm = tr.read_commitments(phase_witness_polys[r]);
    witness_comms.insert(witness_comms.end(), cm.begin(), cm.end());
    std::vector<HFr> ch = tr.squeeze_challenges(phase_challenges[r]);
    challenges.insert(challenges.end(), ch.begin(), ch.end());
  }
  HFr beta = tr.squeeze_challenge();
  std::vector<HG1> m_comms = tr.read_commitments(vp.num_lookups);
  // Lasso lookups (oracle/pyref/hyperplonk.py LassoLookup): read_ts | E | final_cts per lookup, identity-mask framing
  std::vector<HG1> lasso_comms;
  if (vp.num_lasso_lookups) {
    LH_REQUIRE(vp.lasso_lookups != nullptr, LH_ERR_ARG, "hyperplonk: lasso_lookups is null");
    size_t count = 0;
    for (size_t k = 0; k < vp.num_lasso_lookups; k++) {
      const lh_hp_lasso_lookup& lk = vp.lasso_lookups[k];
      lasso_verify_check_table(lk.table, nv);
      if (lk.table.chunk_bits > nv) throw Error(LH_ERR_INVALID_SNARK, "Lasso subtable larger than the circuit");
      count += 2 * lk.table.num_chunks + lk.table.num_memories;
    }
    lasso_comms = lasso_read_commitments(tr, count);
  }
  HFr gamma = tr.squeeze_challenge();
  std::vector<HG1> hz_comms = tr.read_commitments(vp.num_lookups + vp.num_permutation_z_polys);
  HFr alpha = tr.squeeze_challenge();
  std::vector<HFr> y = tr.squeeze_challenges(nv);
  challenges.push_back(beta);
  challenges.push_back(gamma);
  challenges.push_back(alpha);

  // verify_sum_check (verifier.rs:39-90), zero-check: sum = 0
  auto res = sum_check_verify(LH_SC_EVALUATIONS, nv, expr_degree(vp.expression), HFr::zero(), tr);
  const std::vector<HFr>& x = res.second;
  std::set<std::pair<size_t, int>> pcs_query, inst_query;
  std::set<int> lag_used;
  for (size_t i = 0; i < vp.expression.num_nodes; i++) {
    const lh_expr_node& nd = vp.expression.nodes[i];
    if (nd.op == LH_EX_POLYNOMIAL) {
      LH_REQUIRE(nd.a >= 0 && (size_t)std::abs(nd.b) <= nv, LH_ERR_ARG, "expression: bad query");
      ((size_t)nd.a >= vp.num_instance_polys ? pcs_query : inst_query).insert({(size_t)nd.a, nd.b});
    } else if (nd.op == LH_EX_LAGRANGE) {
      lag_used.insert(nd.a);
    }
  }
  EvalCtx cx;
  std::vector<std::vector<HFr>> evals_for_rotation;
  for (auto& q : pcs_query) {
    evals_for_rotation.push_back(tr.read_field_elements((size_t)1 << std::abs(q.second)));
    cx.evals[q] = rotation_eval(x, q.second, evals_for_rotation.back());
  }
  {  // instance_evals (verifier.rs:92-145): instance polys are known to the verifier through Lagrange evals
    long long lo = 0, hi = 0;
    for (auto& q : inst_query) {
      long long i = -(long long)q.second;
      lo = std::min(lo, i);
      hi = std::max(hi, i + (long long)vp.num_instances[q.first]);
    }
    if (lo < 0) lo -= 1;
    if (hi > 0) hi += 1;
    std::map<long long, HFr> lag;
    for (long long i = lo; i < hi; i++)
      if (i != 0) lag[i] = lagrange_eval(x, bh_row(nv, i));
    for (auto& q : inst_query) {
      const long long cnt = (long long)vp.num_instances[q.first], rot = q.second;
      std::vector<long long> idxs;
      if (rot > 0) {
        for (long long i = -rot; i < 0; i++) idxs.push_back(i);
        for (long long i = 1; i <= cnt; i++) idxs.push_back(i);
        idxs.resize((size_t)cnt);
      } else {
        for (long long i = 1 - rot; i < 1 - rot + cnt; i++) idxs.push_back(i);
      }
      HFr acc = HFr::zero();
      for (long long k = 0; k < cnt; k++) acc += instances[q.first][k] * lag.at(idxs[(size_t)k]);
      cx.evals[q] = acc;
    }
  }
  cx.identity = identity_eval(x);
  for (int i : lag_used) cx.lagrange[i] = lagrange_eval(x, bh_row(nv, i));
  cx.eq_xys.push_back(host_eq_xy_eval(x.data(), y.data(), nv));
  cx.challenges = challenges;
  if (eval_expr(vp.expression, cx) != res.first)
    throw Error(LH_ERR_INVALID_SNARK, "Unmatched between sum_check output and query evaluation");

  // points / evaluations in pcs_query order (verifier.rs:76-89,147-180)
  std::set<int> rots;
  for (auto& q : pcs_query) rots.insert(q.second);
  std::map<int, size_t> point_off;
  std::vector<HFr> points;
  size_t num_points = 0;
  for (int r : rots) {
    point_off[r] = num_points;
    for (auto& pt : rotation_eval_points(x, r)) {
      points.insert(points.end(), pt.begin(), pt.end());
      num_points++;
    }
  }
  std::vector<lh_evaluation> evals;
  size_t qi = 0;
  for (auto& q : pcs_query) {
    const std::vector<HFr>& efr = evals_for_rotation[qi++];
    for (size_t k = 0; k < efr.size(); k++) {
      lh_evaluation e;
      e.poly = (uint32_t)q.first;
      e.point = (uint32_t)(point_off[q.second] + k);
      memcpy(&e.value, &efr[k], 32);
      evals.push_back(e);
    }
  }
  std::vector<HG1> comms(vp.num_instance_polys, HG1{host::Fq::zero(), host::Fq::zero()});  // Commitment::default()
  for (size_t i = 0; i < vp.num_preprocess_polys; i++) {
    HG1 p;
    memcpy(&p, &vp.preprocess_comms[i], sizeof(p));
    comms.push_back(p);
  }
  comms.insert(comms.end(), witness_comms.begin(), witness_comms.end());
  for (size_t i = 0; i < vp.num_permutation_polys; i++) {
    HG1 p;
    memcpy(&p, &vp.permutation_comms[i], sizeof(p));
    comms.push_back(p);
  }
  comms.insert(comms.end(), m_comms.begin(), m_comms.end());
  comms.insert(comms.end(), hz_comms.begin(), hz_comms.end());
  // Lasso lookups: the argument's checks, then its claims join the one batch verification
  size_t base = comms.size();
  comms.insert(comms.end(), lasso_comms.begin(), lasso_comms.end());
  for (size_t k = 0; k < vp.num_lasso_lookups; k++) {
    const lh_hp_lasso_lookup& lk = vp.lasso_lookups[k];
    const lh_lasso_table& tb = lk.table;
    const size_t cc = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
    // the prover's range (hyperplonk.cpp): preprocess and witness polys - instance polys have no commitment to open, the
    // polys after the witness do not exist yet when the lookup's columns are read
    const size_t first_committed = vp.num_instance_polys;
    const size_t end_committed = vp.num_instance_polys + vp.num_preprocess_polys + witness_comms.size();
    LH_REQUIRE(lk.output_poly >= first_committed && lk.output_poly < end_committed, LH_ERR_ARG, "hyperplonk: lasso output poly out of range");
    for (size_t j = 0; j < cc; j++)
      LH_REQUIRE(lk.chunk_polys[j] >= first_committed && lk.chunk_polys[j] < end_committed, LH_ERR_ARG,
                 "hyperplonk: lasso chunk poly out of range");
    for (size_t v : {nv, l, cc, alpha}) tr.common_field_element(HFr::from_u64(v));
    LassoClaims cl = lasso_check(tb, nv, tr);
    const size_t p0 = num_points;
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
    push(lk.output_poly, p0, cl.v);
    for (size_t i = 0; i < alpha; i++) push(base + cc + i, p0 + 1, cl.e_rz[i]);
    for (size_t j = 0; j < cc; j++) push(lk.chunk_polys[j], p0 + 2, cl.ev_n[j]);
    for (size_t j = 0; j < cc; j++) push(base + j, p0 + 2, cl.ev_n[cc + j]);
    for (size_t i = 0; i < alpha; i++) push(base + cc + i, p0 + 2, cl.ev_n[2 * cc + i]);
    for (size_t j = 0; j < cc; j++) push(base + cc + alpha + j, p0 + 3, cl.ev_l[j]);
    base += 2 * cc + alpha;
  }
  batch_verify(nv, comms.data(), comms.size(), points.data(), num_points, evals.data(), evals.size(), tr);
}

// ------------------------------------------------------------------ Lasso verify (oracle/pyref/lasso.py:219-261)
struct GpClaim {
  HFr claim;
  std::vector<HFr> point;
};
// oracle/pyref/gkr.py:197-227 (layer schedule of fractional_sum_check.rs:193-270 with p dropped)
static std::vector<HFr> verify_grand_product(const std::vector<size_t>& depth, Transcript& tr,
                                             std::vector<GpClaim>& out) {
  const size_t B = depth.size();
  size_t max_depth = 0;
  for (size_t d : depth) max_depth = std::max(max_depth, d);
  std::vector<HFr> roots = tr.read_field_elements(B), claims = roots, y;
  out.assign(B, GpClaim());
  for (size_t h = 0; h < max_depth; h++) {
    std::vector<size_t> active;
    for (size_t b = 0; b < B; b++)
      if (depth[b] > h) active.push_back(b);
    std::vector<HFr> evals, x;
    if (h == 0) {
      evals = tr.read_field_elements(2 * active.size());
      for (size_t k = 0; k < active.size(); k++)
        if (claims[active[k]] != evals[2 * k] * evals[2 * k + 1])
          throw Error(LH_ERR_INVALID_SUMCHECK, "grand product: root mismatch");
    } else {
      HFr lam = tr.squeeze_challenge(), claim = HFr::zero(), power = HFr::one();
      for (size_t b : active) {
        claim += claims[b] * power;
        power *= lam;
      }
      auto res = sum_check_verify(LH_SC_EVALUATIONS, h, 3, claim, tr);
      x = res.second;
      evals = tr.read_field_elements(2 * active.size());
      HFr acc = HFr::zero();
      power = HFr::one();
      for (size_t k = 0; k < active.size(); k++) {
        acc += evals[2 * k] * evals[2 * k + 1] * power;
        power *= lam;
      }
      if (res.first != acc * host_eq_xy_eval(x.data(), y.data(), h))
        throw Error(LH_ERR_INVALID_SUMCHECK, "grand product: layer " + std::to_string(h) + " mismatch");
    }
    HFr mu = tr.squeeze_challenge();
    y = x;
    y.push_back(mu);
    for (size_t k = 0; k < active.size(); k++) {
      const size_t b = active[k];
      claims[b] = evals[2 * k] + mu * (evals[2 * k + 1] - evals[2 * k]);
      if (depth[b] == h + 1) out[b] = GpClaim{claims[b], y};
    }
  }
  return roots;
}

static HFr subtable_mle_eval(uint32_t kind, const std::vector<HFr>& point) {  // lasso.py:55-72
  if (kind == LH_SUBTABLE_IDENTITY) return identity_eval(point);
  const size_t h = point.size() / 2;
  HFr acc = HFr::zero(), p = HFr::one();
  for (size_t i = 0; i < h; i++) {
    const HFr &yi = point[i], &xi = point[h + i];
    HFr xy = xi * yi;
    acc += (kind == LH_SUBTABLE_AND ? xy : xi + yi - xy.dbl()) * p;
    p = p.dbl();
  }
  return acc;
}

static void lasso_verify_check_table(const lh_lasso_table& tb, size_t n) {
  const size_t c = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  LH_REQUIRE(c >= 1 && c <= LH_LASSO_MAX_CHUNKS && alpha >= 1 && alpha <= LH_LASSO_MAX_MEMORIES &&
                 tb.num_terms <= LH_LASSO_MAX_TERMS,
             LH_ERR_ARG, "lasso: bad table");
  LH_REQUIRE(n >= 1 && l >= 1 && n < 32 && l < 32, LH_ERR_ARG, "lasso: need at least one variable");
  for (size_t i = 0; i < alpha; i++)
    LH_REQUIRE(tb.memory_chunk[i] < c && tb.memory_subtable[i] <= LH_SUBTABLE_XOR, LH_ERR_ARG, "lasso: bad memory");
  LH_REQUIRE(tb.num_terms >= 1, LH_ERR_ARG, "lasso: bad g term count");
  for (size_t m = 0; m < tb.num_terms; m++)
    LH_REQUIRE(tb.g_num_factors[m] >= 1 && tb.g_num_factors[m] <= LH_SC_MAX_FACTORS, LH_ERR_ARG, "lasso: bad g term");
}

// the verifier's side of lasso_argue (oracle/pyref/lasso.py check): reads the messages, checks Surge and the
// memory-checking identities, returns the points and claimed evaluations left to check against the commitments
LassoClaims lasso_check(const lh_lasso_table& tb, size_t n, Transcript& tr) {
  const size_t c = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  auto g_eval = [&](const std::vector<HFr>& vals) {
    HFr acc = HFr::zero();
    for (size_t m = 0; m < tb.num_terms; m++) {
      HFr t;
      memcpy(&t, &tb.g_coeff[m], 32);
      for (size_t k = 0; k < tb.g_num_factors[m]; k++) {
        LH_REQUIRE(tb.g_factor[m][k] < alpha, LH_ERR_ARG, "lasso: bad g factor");
        t *= vals[tb.g_factor[m][k]];
      }
      acc += t;
    }
    return acc;
  };
  size_t g_degree = 0;
  for (size_t m = 0; m < tb.num_terms; m++) g_degree = std::max<size_t>(g_degree, tb.g_num_factors[m]);

  LassoClaims cl;
  cl.r = tr.squeeze_challenges(n);
  cl.v = tr.read_field_element();
  auto surge = sum_check_verify(LH_SC_EVALUATIONS, n, g_degree + 1, cl.v, tr);
  cl.r_z = surge.second;
  cl.e_rz = tr.read_field_elements(alpha);
  if (surge.first != host_eq_xy_eval(cl.r_z.data(), cl.r.data(), n) * g_eval(cl.e_rz))
    throw Error(LH_ERR_INVALID_SNARK, "Surge sum-check final evaluation mismatch");

  HFr gamma = tr.squeeze_challenge(), tau = tr.squeeze_challenge();
  std::vector<size_t> depth(2 * alpha, n);
  depth.insert(depth.end(), 2 * alpha, l);
  std::vector<GpClaim> claims;
  std::vector<HFr> roots = verify_grand_product(depth, tr, claims);
  for (size_t i = 0; i < alpha; i++) {
    const HFr &rs = roots[2 * i], &ws = roots[2 * i + 1], &init = roots[2 * alpha + 2 * i],
              &fin = roots[2 * alpha + 2 * i + 1];
    if (init * ws != rs * fin)
      throw Error(LH_ERR_INVALID_SNARK, "memory " + std::to_string(i) + ": Init*WS != RS*Final");
  }
  cl.r_N = claims[0].point, cl.r_M = claims[2 * alpha].point;

  std::vector<HFr> vals = tr.read_field_elements(3 * c + alpha);
  const HFr *dim_e = &vals[0], *rts_e = &vals[c], *e_e = &vals[2 * c], *fc_e = &vals[2 * c + alpha];
  auto fingerprint = [&](const HFr& a, const HFr& val, const HFr& t) { return a * gamma * gamma + val * gamma + t - tau; };
  const HFr id_M = identity_eval(cl.r_M), one = HFr::one();
  for (size_t i = 0; i < alpha; i++) {
    const size_t j = tb.memory_chunk[i];
    HFr rs = fingerprint(dim_e[j], e_e[i], rts_e[j]);
    HFr init = fingerprint(id_M, subtable_mle_eval(tb.memory_subtable[i], cl.r_M), HFr::zero());
    if (claims[2 * i].claim != rs || claims[2 * i + 1].claim != rs + one || claims[2 * alpha + 2 * i].claim != init ||
        claims[2 * alpha + 2 * i + 1].claim != init + fc_e[j])
      throw Error(LH_ERR_INVALID_SNARK, "memory " + std::to_string(i) + ": leaf claim mismatch");
  }
  cl.ev_n.assign(vals.begin(), vals.begin() + 2 * c + alpha);
  cl.ev_l.assign(vals.begin() + 2 * c + alpha, vals.end());
  return cl;
}

void lasso_verify(const PcsBatchVerify& batch_verify, const lh_lasso_table& tb, size_t n, Transcript& tr) {
  lasso_verify_check_table(tb, n);
  const size_t c = tb.num_chunks, l = tb.chunk_bits, alpha = tb.num_memories;
  for (size_t v : {n, l, c, alpha}) tr.common_field_element(HFr::from_u64(v));
  const size_t nv = std::max(n, l);
  std::vector<HG1> comms = lasso_read_commitments(tr, 1 + 3 * c + alpha);
  LassoClaims cl = lasso_check(tb, n, tr);
  const HFr *dim_e = &cl.ev_n[0], *rts_e = &cl.ev_n[c], *e_e = &cl.ev_n[2 * c], *fc_e = &cl.ev_l[0];

  std::vector<lh_evaluation> evals;
  auto push = [&](size_t poly, size_t point, const HFr& val) {
    lh_evaluation e;
    e.poly = (uint32_t)poly, e.point = (uint32_t)point;
    memcpy(&e.value, &val, 32);
    evals.push_back(e);
  };
  push(0, 0, cl.v);
  for (size_t i = 0; i < alpha; i++) push(1 + 2 * c + i, 1, cl.e_rz[i]);
  for (size_t j = 0; j < c; j++) push(1 + j, 2, dim_e[j]);
  for (size_t j = 0; j < c; j++) push(1 + c + j, 2, rts_e[j]);
  for (size_t i = 0; i < alpha; i++) push(1 + 2 * c + i, 2, e_e[i]);
  for (size_t j = 0; j < c; j++) push(1 + 2 * c + alpha + j, 3, fc_e[j]);
  std::vector<HFr> points;
  const std::vector<HFr>* pts[4] = {&cl.r, &cl.r_z, &cl.r_N, &cl.r_M};
  for (const std::vector<HFr>* pt : pts) {
    points.insert(points.end(), pt->begin(), pt->end());
    points.insert(points.end(), nv - pt->size(), HFr::zero());
  }
  batch_verify(nv, comms.data(), comms.size(), points.data(), 4, evals.data(), evals.size(), tr);
}

}  // namespace lh
