// General plonkish expressions on the host: symbolic expansion of the reference's `Expression` AST
// (util/expression.rs:67-169) into a sum of monomials over atoms, which is what the device kernels
// evaluate (kernels_expr.hip).  Challenges and constants fold into the coefficients; field arithmetic is
// exact, so the expanded form takes the same value as the AST at every point and every sum-check message
// is the same field element as the reference's `ExpressionRegistry` evaluation (evaluator.rs:135-323).
#include <algorithm>
#include <map>
#include "host.hpp"
#include "expr.hpp"

namespace lh {

// BooleanHypercube tables (reference util/arithmetic/bh.rs:5-74)
static const uint32_t BH_PRIMITIVES[32] = {
    1, 3, 7, 11, 19, 37, 67, 131, 285, 529, 1033, 2053, 4179, 8219, 16427, 32771, 65581, 131081, 262183, 524327,
    1048585, 2097157, 4194307, 8388641, 16777243, 33554441, 67108935, 134217767, 268435465, 536870917, 1073741907,
    2147483657u};
static const uint32_t BH_X_INVS[32] = {
    0, 1, 3, 5, 9, 18, 33, 65, 142, 264, 516, 1026, 2089, 4109, 8213, 16385, 32790, 65540, 131091, 262163, 524292,
    1048578, 2097153, 4194320, 8388621, 16777220, 33554467, 67108883, 134217732, 268435458, 536870953, 1073741828};

uint32_t bh_primitive(size_t num_vars) { return BH_PRIMITIVES[num_vars]; }
uint32_t bh_x_inv(size_t num_vars) { return BH_X_INVS[num_vars]; }
size_t bh_next(size_t b, size_t num_vars) {
  b <<= 1;
  return b ^ ((b >> num_vars) * BH_PRIMITIVES[num_vars]);
}
// bh.iter().nth(i): 0, then 1, x, x^2, ...  (bh.rs:127-133)
size_t bh_nth(size_t num_vars, size_t i) {
  if (i == 0) return 0;
  size_t b = 1;
  for (size_t k = 1; k < i; k++) b = bh_next(b, num_vars);
  return b;
}

// ------------------------------------------------------------------ expansion
typedef std::map<std::vector<uint16_t>, HFr> PolyMap;  // sorted atom ids -> coefficient

static uint16_t atom_id(ExpandedExpr& out, const ExprAtom& a) {
  for (size_t i = 0; i < out.atoms.size(); i++)
    if (out.atoms[i].kind == a.kind && out.atoms[i].a == a.a && out.atoms[i].b == a.b) return (uint16_t)i;
  out.atoms.push_back(a);
  return (uint16_t)(out.atoms.size() - 1);
}

static void add_into(PolyMap& dst, const PolyMap& src, bool negate) {
  for (auto& kv : src) {
    HFr v = negate ? -kv.second : kv.second;
    auto it = dst.find(kv.first);
    if (it == dst.end()) dst.emplace(kv.first, v);
    else it->second += v;
  }
}

static ExpandedExpr expand_expr_nodes(const lh_expr& e, const HFr* challenges, size_t num_challenges, std::vector<PolyMap>* node_vals);
ExpandedExpr expand_expr(const lh_expr& e, const HFr* challenges, size_t num_challenges) {
  return expand_expr_nodes(e, challenges, num_challenges, nullptr);
}
// (node_vals: every node's expanded value, for find_eq_factor_shape)
static ExpandedExpr expand_expr_nodes(const lh_expr& e, const HFr* challenges, size_t num_challenges, std::vector<PolyMap>* node_vals) {
  LH_REQUIRE(e.nodes && e.num_nodes >= 1 && e.num_nodes < 100000, LH_ERR_ARG, "expression: empty or too large");
  ExpandedExpr out;
  std::vector<PolyMap> val(e.num_nodes);
  std::vector<int> deg(e.num_nodes, 0);
  auto child = [&](int32_t idx, size_t self) -> size_t {
    LH_REQUIRE(idx >= 0 && (size_t)idx < self, LH_ERR_ARG, "expression: node refers to a later node");
    return (size_t)idx;
  };
  for (size_t i = 0; i < e.num_nodes; i++) {
    const lh_expr_node& nd = e.nodes[i];
    HFr sc;
    memcpy(&sc, &nd.scalar, 32);
    switch (nd.op) {
      case LH_EX_CONSTANT:
        val[i].emplace(std::vector<uint16_t>{}, sc);
        break;
      case LH_EX_IDENTITY:
      case LH_EX_LAGRANGE:
      case LH_EX_EQ_XY:
      case LH_EX_POLYNOMIAL: {
        ExprAtom a{(uint8_t)nd.op, nd.op == LH_EX_IDENTITY ? 0 : nd.a, nd.op == LH_EX_POLYNOMIAL ? nd.b : 0};
        if (nd.op != LH_EX_IDENTITY && nd.op != LH_EX_LAGRANGE) LH_REQUIRE(nd.a >= 0, LH_ERR_ARG, "expression: negative index");
        val[i].emplace(std::vector<uint16_t>{atom_id(out, a)}, HFr::one());
        deg[i] = 1;
        break;
      }
      case LH_EX_CHALLENGE:
        LH_REQUIRE(nd.a >= 0 && (size_t)nd.a < num_challenges, LH_ERR_ARG, "expression: challenge index out of range");
        val[i].emplace(std::vector<uint16_t>{}, challenges[nd.a]);
        break;
      case LH_EX_NEGATED: {
        size_t a = child(nd.a, i);
        add_into(val[i], val[a], true);
        deg[i] = deg[a];
        break;
      }
      case LH_EX_SUM: {
        size_t a = child(nd.a, i), b = child(nd.b, i);
        val[i] = val[a];
        add_into(val[i], val[b], false);
        deg[i] = std::max(deg[a], deg[b]);
        break;
      }
      case LH_EX_PRODUCT: {
        size_t a = child(nd.a, i), b = child(nd.b, i);
        LH_REQUIRE(val[a].size() * val[b].size() < 200000, LH_ERR_ARG, "expression: expansion too large");
        for (auto& x : val[a])
          for (auto& y : val[b]) {
            std::vector<uint16_t> key(x.first);
            key.insert(key.end(), y.first.begin(), y.first.end());
            std::sort(key.begin(), key.end());
            HFr v = x.second * y.second;
            auto it = val[i].find(key);
            if (it == val[i].end()) val[i].emplace(std::move(key), v);
            else it->second += v;
          }
        deg[i] = deg[a] + deg[b];
        break;
      }
      case LH_EX_SCALED: {
        size_t a = child(nd.a, i);
        for (auto& x : val[a]) val[i].emplace(x.first, x.second * sc);
        deg[i] = deg[a];
        break;
      }
      default:
        throw Error(LH_ERR_ARG, "expression: unknown node op");
    }
  }
  out.degree = deg.back();  // Expression::degree() is structural (expression.rs:171-182)
  for (auto& kv : val.back())
    if (!kv.second.is_zero()) out.monos.push_back(ExprMono{kv.second, kv.first});
  if (node_vals) *node_vals = std::move(val);
  return out;
}

EqFactorShape find_eq_factor_shape(const lh_expr& e, const HFr* challenges, size_t num_challenges, const ExpandedExpr& ex) {
  EqFactorShape sh;
  // exactly one eq atom in the whole expression
  int eq_atom = -1;
  for (size_t a = 0; a < ex.atoms.size(); a++)
    if (ex.atoms[a].kind == LH_EX_EQ_XY) {
      if (eq_atom >= 0) return sh;
      eq_atom = (int)a;
    }
  if (eq_atom < 0) return sh;
  // the product node C * eq (the last such node: the one the composition puts on top of the constraints)
  int c_node = -1;
  for (size_t i = 0; i < e.num_nodes; i++) {
    const lh_expr_node& nd = e.nodes[i];
    if (nd.op != LH_EX_PRODUCT) continue;
    if (e.nodes[nd.b].op == LH_EX_EQ_XY) c_node = nd.a;
    else if (e.nodes[nd.a].op == LH_EX_EQ_XY) c_node = nd.b;
  }
  if (c_node < 0) return sh;
  std::vector<PolyMap> val;
  ExpandedExpr again = expand_expr_nodes(e, challenges, num_challenges, &val);
  // (atom ids are assigned in node order by the same routine: `again` numbers them as `ex` does)
  if (again.atoms.size() != ex.atoms.size()) return sh;
  const PolyMap& C = val[(size_t)c_node];
  const PolyMap& E = val.back();
  if (C.empty()) return sh;
  auto with_eq = [&](const std::vector<uint16_t>& m) {
    std::vector<uint16_t> k(m);
    k.push_back((uint16_t)eq_atom);
    std::sort(k.begin(), k.end());
    return k;
  };
  // kappa from one monomial, then every monomial of E accounted for
  HFr kappa = HFr::zero();
  int c_degree = 0;
  for (auto& kv : C) {
    if (kv.second.is_zero()) continue;
    for (uint16_t a : kv.first)
      if (a == (uint16_t)eq_atom) return sh;  // eq inside C: not this shape
    auto it = E.find(with_eq(kv.first));
    if (it == E.end()) return sh;
    if (kappa.is_zero()) kappa = it->second * kv.second.inv();
    if (it->second != kappa * kv.second) return sh;
    c_degree = std::max<int>(c_degree, (int)kv.first.size());
  }
  if (kappa.is_zero()) return sh;
  size_t eq_monos = 0, c_monos = 0;
  for (auto& kv : C) c_monos += kv.second.is_zero() ? 0 : 1;
  for (auto& kv : E) {
    if (kv.second.is_zero()) continue;
    const bool has_eq = std::find(kv.first.begin(), kv.first.end(), (uint16_t)eq_atom) != kv.first.end();
    if (has_eq) {
      eq_monos++;
      continue;
    }
    if (kv.first.size() != 1) return sh;  // the part beside eq * C must be linear in single atoms (no constant either)
    sh.lin.push_back({kv.first[0], kv.second});
  }
  if (eq_monos != c_monos || sh.lin.size() > (size_t)LIN_MAX_TABLES || c_degree < 1 || c_degree + 1 != ex.degree) return sh;
  sh.ok = true, sh.c_node = c_node, sh.eq_atom = (uint16_t)eq_atom, sh.kappa = kappa, sh.c_degree = c_degree;
  return sh;
}

// ------------------------------------------------------------------ expression -> register program
// Code generation for k_sc_round_prog: constants and challenges fold on the host, leaves are operands (no load
// instruction), inner nodes are emitted in Sethi-Ullman order so that the register file stays within
// PROG_MAX_REGS.  The program computes the same field value as the AST at every point.
struct Program {
  std::vector<uint32_t> code;
  std::vector<Fr> consts;
  uint32_t num_regs = 0, result_reg = 0;
  bool ok = false;
};
namespace {
struct Opnd {
  uint32_t kind, idx;
};
struct ProgBuilder {
  const lh_expr& e;
  std::vector<char> is_const;
  std::vector<HFr> cval;
  std::vector<int> need, table;  // table: leaf node -> table index
  Program prog;
  std::vector<uint32_t> free_regs;
  uint32_t regs_in_use = 0;
  bool failed = false;

  explicit ProgBuilder(const lh_expr& e_) : e(e_) {}
  uint32_t const_idx(const HFr& v) {
    Fr d = dev(v);
    for (size_t i = 0; i < prog.consts.size(); i++)
      if (memcmp(&prog.consts[i], &d, sizeof(Fr)) == 0) return (uint32_t)i;
    prog.consts.push_back(d);
    return (uint32_t)prog.consts.size() - 1;
  }
  uint32_t alloc_reg() {
    if (!free_regs.empty()) {
      uint32_t r = free_regs.back();
      free_regs.pop_back();
      return r;
    }
    if (regs_in_use >= (uint32_t)PROG_MAX_REGS) {
      failed = true;
      return 0;
    }
    prog.num_regs = std::max(prog.num_regs, regs_in_use + 1);
    return regs_in_use++;
  }
  void release(const Opnd& o) {
    if (o.kind == PROG_REG) free_regs.push_back(o.idx);
  }
  void emit(uint32_t op, uint32_t dst, const Opnd& a, const Opnd& b) {
    if (a.idx > 0xffff || b.idx > 0xffff || prog.code.size() > 2 * 4096) failed = true;
    prog.code.push_back(op | dst << 4 | a.kind << 8 | b.kind << 10);
    prog.code.push_back(a.idx | b.idx << 16);
  }
  Opnd binary(uint32_t op, const Opnd& a, const Opnd& b) {
    // reuse an operand register as the destination (the operands are read before the store)
    uint32_t dst = a.kind == PROG_REG ? a.idx : b.kind == PROG_REG ? b.idx : alloc_reg();
    emit(op, dst, a, b);
    if (a.kind == PROG_REG && a.idx != dst) release(a);
    if (b.kind == PROG_REG && b.idx != dst) release(b);
    return Opnd{PROG_REG, dst};
  }
  Opnd gen(int n) {
    const lh_expr_node& nd = e.nodes[n];
    if (is_const[n]) return Opnd{PROG_CONST, const_idx(cval[n])};
    switch (nd.op) {
      case LH_EX_IDENTITY:
      case LH_EX_LAGRANGE:
      case LH_EX_EQ_XY:
      case LH_EX_POLYNOMIAL: return Opnd{PROG_ATOM, (uint32_t)table[n]};
      case LH_EX_NEGATED: {
        Opnd a = gen(nd.a);
        uint32_t dst = a.kind == PROG_REG ? a.idx : alloc_reg();
        emit(PROG_NEG, dst, a, Opnd{PROG_REG, 0});
        return Opnd{PROG_REG, dst};
      }
      case LH_EX_SCALED: {
        HFr sc;
        memcpy(&sc, &nd.scalar, 32);
        Opnd a = gen(nd.a);
        if (sc == HFr::one()) return a;
        return binary(PROG_MUL, a, Opnd{PROG_CONST, const_idx(sc)});
      }
      default: {  // SUM / PRODUCT
        int l = nd.a, r = nd.b;
        uint32_t op = nd.op == LH_EX_SUM ? PROG_ADD : PROG_MUL;
        if (nd.op == LH_EX_SUM) {
          if (is_const[l] && cval[l].is_zero()) return gen(r);
          if (is_const[r] && cval[r].is_zero()) return gen(l);
          if (!is_const[r] && e.nodes[r].op == LH_EX_NEGATED) {  // a + (-b) -> a - b
            op = PROG_SUB;
            r = e.nodes[r].a;
          }
        } else {
          if (is_const[l] && cval[l] == HFr::one()) return gen(r);
          if (is_const[r] && cval[r] == HFr::one()) return gen(l);
        }
        Opnd a, b;
        if (need[r] > need[l]) {
          b = gen(r);
          a = gen(l);
        } else {
          a = gen(l);
          b = gen(r);
        }
        return binary(op, a, b);
      }
    }
  }
};
}  // namespace

// leaf_table(node index) -> table index of an atom leaf
// (pair_table >= 0: the result is multiplied by the per-pair entry of that table - PROG_PAIR - at the end)
static Program compile_program(const lh_expr& e, const HFr* challenges, size_t num_challenges,
                               const std::function<int(const lh_expr_node&)>& leaf_table, int pair_table = -1) {
  ProgBuilder pb(e);
  const size_t N = e.num_nodes;
  pb.is_const.assign(N, 0);
  pb.cval.assign(N, HFr::zero());
  pb.need.assign(N, 0);
  pb.table.assign(N, -1);
  for (size_t i = 0; i < N; i++) {
    const lh_expr_node& nd = e.nodes[i];
    HFr sc;
    memcpy(&sc, &nd.scalar, 32);
    auto both_const = [&]() { return pb.is_const[nd.a] && pb.is_const[nd.b]; };
    switch (nd.op) {
      case LH_EX_CONSTANT: pb.is_const[i] = 1, pb.cval[i] = sc; break;
      case LH_EX_CHALLENGE:
        LH_REQUIRE(nd.a >= 0 && (size_t)nd.a < num_challenges, LH_ERR_ARG, "expression: challenge index out of range");
        pb.is_const[i] = 1, pb.cval[i] = challenges[nd.a];
        break;
      case LH_EX_NEGATED:
        if (pb.is_const[nd.a]) pb.is_const[i] = 1, pb.cval[i] = -pb.cval[nd.a];
        else pb.need[i] = std::max(pb.need[nd.a], 1);
        break;
      case LH_EX_SCALED:
        if (pb.is_const[nd.a]) pb.is_const[i] = 1, pb.cval[i] = pb.cval[nd.a] * sc;
        else pb.need[i] = std::max(pb.need[nd.a], 1);
        break;
      case LH_EX_SUM:
      case LH_EX_PRODUCT:
        if (both_const()) {
          pb.is_const[i] = 1;
          pb.cval[i] = nd.op == LH_EX_SUM ? pb.cval[nd.a] + pb.cval[nd.b] : pb.cval[nd.a] * pb.cval[nd.b];
        } else if (nd.op == LH_EX_PRODUCT && ((pb.is_const[nd.a] && pb.cval[nd.a].is_zero()) ||
                                              (pb.is_const[nd.b] && pb.cval[nd.b].is_zero()))) {
          pb.is_const[i] = 1, pb.cval[i] = HFr::zero();
        } else {
          int na = pb.need[nd.a], nb = pb.need[nd.b];
          pb.need[i] = na == nb ? na + 1 : std::max(na, nb);
        }
        break;
      default: pb.table[i] = leaf_table(nd);
    }
  }
  Opnd res = pb.gen((int)N - 1);
  if (pair_table >= 0) res = pb.binary(PROG_MUL, res, Opnd{PROG_PAIR, (uint32_t)pair_table});
  if (res.kind != PROG_REG) {
    uint32_t dst = pb.alloc_reg();
    pb.emit(PROG_MOV, dst, res, Opnd{PROG_REG, 0});
    res = Opnd{PROG_REG, dst};
  }
  pb.prog.result_reg = res.idx;
  pb.prog.ok = !pb.failed && pb.prog.num_regs >= 1;
  return pb.prog;
}

// ------------------------------------------------------------------ ClassicSumCheck<EvaluationsProver> over an Expression
SumCheckResult sum_check_prove_expr(Ctx& c, size_t num_vars, const lh_expr& expr, const Fr* const* d_polys,
                                    size_t num_polys, const HFr* challenges, size_t num_challenges, const HFr* ys,
                                    size_t num_ys, const HFr& sum, Transcript& tr, bool sharded) {
  LH_REQUIRE(num_vars > 0 && num_vars < 32, LH_ERR_ARG, "sum-check needs 0 < num_vars < 32");  // classic.rs:42, bh.rs:85
  ExpandedExpr ex = expand_expr(expr, challenges, num_challenges);
  LH_REQUIRE(ex.degree >= 2 && ex.degree <= 8, LH_ERR_ARG, "EvaluationsProver: degree must be in 2..8");  // eval.rs:316
  // `sharded` (one proof over several GPUs, dev.hpp Shard): d_polys are this rank's shards and so is every table built
  // here - eq tables with the rank's factor, identity / Lagrange tables by global row; a ROTATED poly (classic.rs:104-126:
  // the rotation is a multiplication in GF(2^n), it mixes every index bit) is the one thing that needs the other ranks'
  // rows: the poly is gathered once, rotated, and this rank's rows are kept.  The rounds then run on the shards with
  // their partial sums added over the ranks (sum_check_loop).
  const Shard sh(c);
  if (sharded) LH_REQUIRE(sh.on && sh.sharded(num_vars) && sh.j >= 1, LH_ERR_ARG, "sharded sum-check outside a sharded proof");
  const size_t n_full = (size_t)1 << num_vars;
  const size_t n = sharded ? n_full >> sh.rho : n_full;  // entries of every (local) table
  ArenaScope scope(c.arena);

  // eq factoring of the zero-check shape E = linear part + kappa eq(y, .) C (host.hpp EqFactoring, expr.hpp EqFactorShape):
  // the streaming rounds evaluate C alone at one point fewer, weighted with ONE entry of a halving eq level per pair,
  // instead of E over a streamed and bound eq table; the host rebuilds the reference's message.  Single-GPU sum-checks
  // of >= 2^14 rows through the register program; LH_SC_EQ_FACTORING=0 / option sc_eq_factoring switch it off.
  static const bool use_prog = !(getenv("LH_EXPR_MONOMIALS") && atoi(getenv("LH_EXPR_MONOMIALS")));
  EqFactorShape shp;
  // (below 2^14 rows the rounds are latency-bound and the levels' set-up is not worth it; LH_EXPR_EF_MIN_VARS: tests)
  static const size_t ef_min_vars = [] {
    const char* e = getenv("LH_EXPR_EF_MIN_VARS");
    return e && atoi(e) >= 2 ? (size_t)atoi(e) : (size_t)14;
  }();
  if (c.opt.sc_eq_factoring != 0 && use_prog && !sharded && num_vars >= ef_min_vars)
    shp = find_eq_factor_shape(expr, challenges, num_challenges, ex);
  if (shp.ok) {  // (1 - y_j) must be invertible in every round
    const HFr* y = ys + (size_t)ex.atoms[shp.eq_atom].a * num_vars;
    LH_REQUIRE((size_t)ex.atoms[shp.eq_atom].a < num_ys, LH_ERR_ARG, "expression: eq_xy index out of range");
    for (size_t i = 0; i < num_vars; i++) shp.ok = shp.ok && !(HFr::one() - y[i]).is_zero();
  }

  // tables: every poly at the current rotation first (all of them are bound and reported), then one
  // table per remaining atom
  std::vector<const Fr*> tables(d_polys, d_polys + num_polys);
  std::vector<int> table_of(ex.atoms.size(), -1);
  for (size_t a = 0; a < ex.atoms.size(); a++) {
    const ExprAtom& at = ex.atoms[a];
    if (at.kind == LH_EX_POLYNOMIAL) {
      LH_REQUIRE((size_t)at.a < num_polys, LH_ERR_ARG, "expression: poly index out of range");
      LH_REQUIRE((size_t)std::abs(at.b) <= num_vars, LH_ERR_ARG, "expression: rotation distance > num_vars");  // classic.rs:42
      if (at.b == 0) {
        table_of[a] = at.a;
      } else {
        Fr* rot = c.arena.alloc_n<Fr>(n);
        if (sharded) {
          ArenaScope tmp(c.arena);
          Fr* full = c.arena.alloc_n<Fr>(n_full);
          Fr* full_rot = c.arena.alloc_n<Fr>(n_full);
          comm_gather_tables(c, d_polys[at.a], 1, n, (size_t)1 << sh.j, &full);
          c.route.v[RouteStats::SHARD_EXCHANGES]++;
          k_rotate_gather(c, full, num_vars, at.b, bh_primitive(num_vars), bh_x_inv(num_vars), full_rot);
          k_shard_extract(c, full_rot, n, sh.j, sh.rho, sh.rank, sizeof(Fr), rot);
          c.sync();  // (the temporaries are released with this scope)
        } else {
          k_rotate_gather(c, d_polys[at.a], num_vars, at.b, bh_primitive(num_vars), bh_x_inv(num_vars), rot);
        }
        table_of[a] = (int)tables.size();
        tables.push_back(rot);
      }
    } else if (at.kind == LH_EX_EQ_XY) {
      LH_REQUIRE((size_t)at.a < num_ys, LH_ERR_ARG, "expression: eq_xy index out of range");
      Fr* eq = nullptr;
      if (!shp.ok) {  // (factored: the slot is filled when - if - the rounds leave the factored form, sum_check_loop)
        eq = c.arena.alloc_n<Fr>(n);
        if (sharded) eq_xy_shard(c, sh, ys + (size_t)at.a * num_vars, num_vars, 0, eq);
        else k_eq_xy(c, (const Fr*)(ys + (size_t)at.a * num_vars), num_vars, eq);
      }
      table_of[a] = (int)tables.size();
      tables.push_back(eq);
    } else if (at.kind == LH_EX_IDENTITY) {
      Fr* id = c.arena.alloc_n<Fr>(n);
      if (sharded) k_identity_table_shard(c, n, sh.j, sh.rho, sh.rank, id);
      else k_identity_table(c, n, id);
      table_of[a] = (int)tables.size();
      tables.push_back(id);
    } else {  // Lagrange(i): 1 on row bh[i mod 2^n] (classic.rs:46-55)
      long long m = (long long)at.a % (long long)n_full;
      if (m < 0) m += (long long)n_full;
      Fr* l = c.arena.alloc_n<Fr>(n);
      const size_t hot = bh_nth(num_vars, (size_t)m);
      if (!sharded) {
        k_one_hot_table(c, n, hot, l);
      } else if (((hot >> sh.j) & (sh.R - 1)) == sh.rank) {  // this rank's row
        k_one_hot_table(c, n, ((hot >> (sh.j + sh.rho)) << sh.j) | (hot & (((size_t)1 << sh.j) - 1)), l);
      } else {
        LH_HIP(hipMemsetAsync(l, 0, n * sizeof(Fr), c.stream));
      }
      table_of[a] = (int)tables.size();
      tables.push_back(l);
    }
  }
  const size_t T = tables.size();
  LH_REQUIRE(T <= (size_t)SC_MAX_TABLES, LH_ERR_ARG, "expression: too many tables for one round kernel");
  LH_REQUIRE(!ex.monos.empty(), LH_ERR_ARG, "expression: identically zero");

  // monomial list -> device
  const uint32_t M = (uint32_t)ex.monos.size();
  std::vector<Fr> coeff(M);
  std::vector<uint8_t> is_one(M), fac, store;
  std::vector<uint32_t> off(M + 1, 0);
  std::vector<char> used(T, 0);
  const HFr one = HFr::one();
  for (uint32_t m = 0; m < M; m++) {
    coeff[m] = dev(ex.monos[m].coeff);
    is_one[m] = ex.monos[m].coeff == one;
    for (uint16_t a : ex.monos[m].atoms) {
      int t = table_of[a];
      fac.push_back((uint8_t)t);
      store.push_back(used[t] ? 0 : 1);
      used[t] = 1;
    }
    off[m + 1] = (uint32_t)fac.size();
  }
  if (fac.empty()) fac.push_back(0), store.push_back(0);
  Fr* d_coeff = c.arena.alloc_n<Fr>(M);
  uint8_t* d_is_one = c.arena.alloc_n<uint8_t>(M);
  uint32_t* d_off = c.arena.alloc_n<uint32_t>(M + 1);
  uint8_t* d_fac = c.arena.alloc_n<uint8_t>(fac.size());
  uint8_t* d_store = c.arena.alloc_n<uint8_t>(store.size());
  LH_HIP(hipMemcpyAsync(d_coeff, coeff.data(), M * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemcpyAsync(d_is_one, is_one.data(), M, hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemcpyAsync(d_off, off.data(), (M + 1) * 4, hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemcpyAsync(d_fac, fac.data(), fac.size(), hipMemcpyHostToDevice, c.stream));
  LH_HIP(hipMemcpyAsync(d_store, store.data(), store.size(), hipMemcpyHostToDevice, c.stream));
  c.sync();  // the host vectors go out of scope before the first round is queued otherwise

  // preferred path: the expression as a register program (bind pass + evaluation from the bound tables)
  if (use_prog) {
    auto leaf_table = [&](const lh_expr_node& nd) -> int {
      for (size_t a = 0; a < ex.atoms.size(); a++)
        if (ex.atoms[a].kind == nd.op && (nd.op == LH_EX_IDENTITY || (ex.atoms[a].a == nd.a &&
                                                                        (nd.op != LH_EX_POLYNOMIAL || ex.atoms[a].b == nd.b))))
          return table_of[a];
      throw Error(LH_ERR_ARG, "expression: leaf without a table");
    };
    Program prog = compile_program(expr, challenges, num_challenges, leaf_table);
    // the factored form's program: C times the pair's eq-level entry
    Program progc;
    if (shp.ok && prog.ok) {
      lh_expr sub = expr;
      sub.num_nodes = (size_t)shp.c_node + 1;  // (a node only refers to earlier nodes: C's subtree lies in this prefix)
      progc = compile_program(sub, challenges, num_challenges, leaf_table, table_of[shp.eq_atom]);
      if (progc.ok && getenv("LH_HP_DEBUG"))
        fprintf(stderr, "[expr] factored: eq * C with C of degree %d in %zu instructions, %zu linear atoms beside it\n", shp.c_degree,
                progc.code.size() / 2, shp.lin.size());
    }
    if (shp.ok && !(prog.ok && progc.ok)) {  // not factored after all: the eq table the atoms' loop left out
      const ExprAtom& at = ex.atoms[shp.eq_atom];
      Fr* eq = c.arena.alloc_n<Fr>(n);
      k_eq_xy(c, (const Fr*)(ys + (size_t)at.a * num_vars), num_vars, eq);
      tables[(size_t)table_of[shp.eq_atom]] = eq;
      shp.ok = false;
    }
    if (prog.ok && getenv("LH_HP_DEBUG")) {
      size_t muls = 0, atoms = 0;
      for (size_t i = 0; i < prog.code.size(); i += 2) {
        const uint32_t w0 = prog.code[i];
        muls += (w0 & 15u) == PROG_MUL;
        atoms += ((w0 >> 8) & 3u) == PROG_ATOM;
        atoms += ((w0 & 15u) <= PROG_MUL) && ((w0 >> 10) & 3u) == PROG_ATOM;
      }
      fprintf(stderr, "[expr] program: %zu instructions, %zu multiplications, %zu table operands, %u registers, %zu constants, degree %d, %zu tables\n",
              prog.code.size() / 2, muls, atoms, prog.num_regs, prog.consts.size(), ex.degree, T);
    }
    if (prog.ok) {
      uint32_t* d_code = c.arena.alloc_n<uint32_t>(prog.code.size());
      Fr* d_consts = c.arena.alloc_n<Fr>(std::max<size_t>(prog.consts.size(), 1));
      LH_HIP(hipMemcpyAsync(d_code, prog.code.data(), prog.code.size() * 4, hipMemcpyHostToDevice, c.stream));
      if (!prog.consts.empty())
        LH_HIP(hipMemcpyAsync(d_consts, prog.consts.data(), prog.consts.size() * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
      c.sync();
      ProgRound pr;
      memset(&pr, 0, sizeof(pr));
      pr.num_tables = (uint32_t)T;
      pr.num_instrs = (uint32_t)prog.code.size() / 2, pr.num_regs = prog.num_regs, pr.result_reg = prog.result_reg;
      pr.code = d_code, pr.consts = d_consts;
      // large sum-checks run the program as compiled straight-line code (jit.cpp), the rest interpret it.  (Factored: the
      // program of E is compiled only if a round of a large table ever needs it - a claim that is not the true sum; the small
      // rounds behind the factored ones interpret it.)
      const JitKernel* jit = nullptr;
      bool jit_asked = false;
      auto prog_round = [&](const Fr* const* in, Fr* const* out, const Fr& r, bool bind, size_t size, Fr* evals_host) {
        if (!jit_asked && jit_enabled(num_vars) && (!shp.ok || size >= ((size_t)1 << 15))) {
          jit = jit_sc_round(c, prog.code.data(), pr.num_instrs, pr.num_regs, pr.result_reg, ex.degree);
          jit_asked = true;
        }
        if (bind) {
          std::vector<const Fr*> src;
          std::vector<Fr*> dst;
          for (size_t i = 0; i < T; i++)
            if (used[i]) src.push_back(in[i]), dst.push_back(out[i]);
          k_fix_var_multi(c, src.data(), dst.data(), src.size(), 4 * size, r);
        }
        for (size_t i = 0; i < T; i++) pr.in[i] = bind ? out[i] : in[i];
        k_sc_round_prog(c, pr, ex.degree, size, evals_host, jit);
      };
      // ---- the factored form: C's program, the linear part's pair sums beside it
      if (shp.ok) {
        uint32_t* d_codec = c.arena.alloc_n<uint32_t>(progc.code.size());
        Fr* d_constsc = c.arena.alloc_n<Fr>(std::max<size_t>(progc.consts.size(), 1));
        LH_HIP(hipMemcpyAsync(d_codec, progc.code.data(), progc.code.size() * 4, hipMemcpyHostToDevice, c.stream));
        if (!progc.consts.empty())
          LH_HIP(hipMemcpyAsync(d_constsc, progc.consts.data(), progc.consts.size() * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
        c.sync();
        ProgRound prc;
        memset(&prc, 0, sizeof(prc));
        prc.num_tables = (uint32_t)T;
        prc.num_instrs = (uint32_t)progc.code.size() / 2, prc.num_regs = progc.num_regs, prc.result_reg = progc.result_reg;
        prc.code = d_codec, prc.consts = d_constsc;
        const JitKernel* jitc =
            jit_enabled(num_vars) ? jit_sc_round(c, progc.code.data(), prc.num_instrs, prc.num_regs, prc.result_reg, ex.degree - 1) : nullptr;
        const size_t t_eq = (size_t)table_of[shp.eq_atom];
        const HFr* y = ys + (size_t)ex.atoms[shp.eq_atom].a * num_vars;
        EqFactoring ef;
        {
          std::vector<HFr> d(num_vars), pre(num_vars + 1);
          pre[0] = HFr::one();
          for (size_t i = 0; i < num_vars; i++) d[i] = HFr::one() - y[i], pre[i + 1] = pre[i] * d[i];
          HFr inv = pre[num_vars].inv();  // (1 - y_j)^-1 for all rounds with one inversion
          ef.inv_1my.resize(num_vars);
          for (size_t i = num_vars; i-- > 0;) {
            ef.inv_1my[i] = inv * pre[i];
            inv = inv * d[i];
          }
        }
        ef.per_term = false, ef.trusted_claim = false;
        ef.kappa = shp.kappa;
        const HFr kappa_inv = shp.kappa.inv();
        ef.c = sum * kappa_inv;  // (round 0 takes the linear part's sum off it, below)
        EqFactoring::One one;
        one.table = t_eq, one.y = y, one.S = one.S_prev = HFr::one();
        {
          // E_0 = eq table of y[1..], then every lower level by adding pairs (sumcheck.cpp: the same tables)
          const size_t half = n >> 1;
          Fr* buf = c.arena.alloc_n<Fr>(2 * half);
          one.level.resize(num_vars);
          size_t off = 0;
          for (size_t jl = 0; jl < num_vars; jl++) one.level[jl] = buf + off, off += half >> jl;
          k_eq_xy(c, (const Fr*)(y + 1), num_vars - 1, buf);
          std::vector<Fr*> lower;
          for (size_t jl = 1; jl < num_vars; jl++) lower.push_back((Fr*)one.level[jl]);
          k_eq_levels(c, one.level[0], half, lower.data(), lower.size());
        }
        ef.eqs.push_back(one);
        // (factored down to the last round: every eq level exists, the compiled program of C serves every size, and the
        //  program of E - with its eq table - is never needed)
        ef.streams = [](bool, size_t) { return true; };
        LinSums ls;
        memset(&ls, 0, sizeof(ls));
        ls.count = (uint32_t)shp.lin.size();
        for (uint32_t i = 0; i < ls.count; i++) ls.coeff[i] = dev(shp.lin[i].second);
        ef.round = [&](const Fr* const* in, Fr* const* out, const Fr& r, bool bind, size_t size, size_t round, int points,
                       Fr* out_host) {
          ArenaScope round_scope(c.arena);
          if (bind) {
            std::vector<const Fr*> src;
            std::vector<Fr*> dst;
            for (size_t i = 0; i < T; i++)
              if (used[i] && i != t_eq) src.push_back(in[i]), dst.push_back(out[i]);
            k_fix_var_multi(c, src.data(), dst.data(), src.size(), 4 * size, r);
          }
          for (size_t i = 0; i < T; i++) prc.in[i] = bind ? out[i] : in[i];
          prc.in[t_eq] = ef.eqs[0].level[round];
          if (ls.count) {
            for (uint32_t i = 0; i < ls.count; i++) ls.t[i] = prc.in[table_of[shp.lin[i].first]];
            k_lin_sums(c, ls, size, out_host + 12);
          }
          k_sc_round_prog(c, prc, points, size, out_host, jitc);
          ef.lin0 = ls.count ? hst(out_host[12]) : HFr::zero();
          ef.lin1 = ls.count ? hst(out_host[13]) : HFr::zero();
          if (round == 0) ef.c = (sum - ef.lin0 - ef.lin1) * kappa_inv;
        };
        return sum_check_loop(c, LH_SC_EVALUATIONS, num_vars, ex.degree, tables, used, num_polys, sum, tr, sharded, prog_round, nullptr, &ef);
      }
      return sum_check_loop(c, LH_SC_EVALUATIONS, num_vars, ex.degree, tables, used, num_polys, sum, tr, sharded, prog_round);
    }
  }

  ExtRound rd;
  memset(&rd, 0, sizeof(rd));
  rd.num_tables = (uint32_t)T;
  rd.num_terms = M;
  rd.coeff = d_coeff, rd.is_one = d_is_one, rd.off = d_off, rd.fac = d_fac, rd.store = d_store;
  auto round_fn = [&](const Fr* const* in, Fr* const* out, const Fr& r, bool bind, size_t size, Fr* evals_host) {
    for (size_t i = 0; i < T; i++) {
      rd.in[i] = in[i];
      rd.out[i] = out[i];
    }
    rd.r = r;
    k_sc_round_ext(c, rd, ex.degree, bind, size, evals_host);
  };
  return sum_check_loop(c, LH_SC_EVALUATIONS, num_vars, ex.degree, tables, used, num_polys, sum, tr, sharded, round_fn);
}

}  // namespace lh
