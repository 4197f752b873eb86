// piop::gkr on the device: prove_fractional_sum_check (reference piop/gkr/fractional_sum_check.rs:89-190) and the
// product-only layered circuit of Lasso's memory checking (oracle/pyref/gkr.py::prove_grand_product): tree building on
// shards, the resident layers near the roots, one sum-check per larger layer.
#include <algorithm>
#include <functional>
#include <chrono>
#include <memory>
#include <thread>
#include "host.hpp"
#include "resident_host.hpp"

namespace lh {

// ------------------------------------------------------------------ prove_fractional_sum_check
// reference piop/gkr/fractional_sum_check.rs:89-190
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
  if (c.gkr_hook) {  // (the trees are built: from here on the small layers leave most of the chip idle)
    // (before the resident launch: what the hook starts on another stream waits for an event recorded HERE on this
    // ctx's stream - behind the resident kernel it would wait for the whole resident phase.  Later starts - at the layer
    // with 2^12 .. 2^21 entries - were measured in rounds 3 and 4 and are monotonically worse: profiles/README.md)
    std::function<void()> hook;
    hook.swap(c.gkr_hook);
    hook();
  }
  if (!resident_layers.empty()) resident.launch(resident_layers);
  for (size_t h = 0; h < max_depth; h++) {
    std::vector<size_t> active;
    for (size_t b = 0; b < B; b++)
      if (num_vars[b] > h) active.push_back(b);
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


}  // namespace lh
