// Expanded form of a plonkish Expression (see expr.cpp).
#pragma once
#include <vector>
#include "host.hpp"

namespace lh {

struct ExprAtom {
  uint8_t kind;  // LH_EX_IDENTITY / LH_EX_LAGRANGE / LH_EX_EQ_XY / LH_EX_POLYNOMIAL
  int32_t a, b;  // Lagrange: i ; EqXY: idx ; Polynomial: (poly, rotation)
};
struct ExprMono {
  HFr coeff;
  std::vector<uint16_t> atoms;  // ids into ExpandedExpr::atoms, sorted, with repetition
};
struct ExpandedExpr {
  std::vector<ExprAtom> atoms;
  std::vector<ExprMono> monos;
  int degree = 0;  // Expression::degree() of the original AST
};
ExpandedExpr expand_expr(const lh_expr& e, const HFr* challenges, size_t num_challenges);
// The zero-check shape  E = (linear part) + kappa * eq(y, .) * C  (preprocessor.rs:43-57: DP([h_0, .., h_{L-1}, DP(constraints,
// alpha) * eq_0], alpha)): `c_node` is the AST node of C (a factor of the one product node that has the eq leaf as its other
// factor), the linear part a sum of single atoms.  Found structurally and then VERIFIED on the expanded monomials - E's
// monomials are exactly the linear part's plus kappa * eq * (C's monomials) - so a tree of another shape is simply not
// factored.  ok = false: no such shape.
struct EqFactorShape {
  bool ok = false;
  int c_node = -1;
  uint16_t eq_atom = 0;  // id into ExpandedExpr::atoms
  HFr kappa;
  std::vector<std::pair<uint16_t, HFr>> lin;  // (atom id, coefficient)
  int c_degree = 0;
};
EqFactorShape find_eq_factor_shape(const lh_expr& e, const HFr* challenges, size_t num_challenges, const ExpandedExpr& ex);

uint32_t bh_primitive(size_t num_vars);
uint32_t bh_x_inv(size_t num_vars);
size_t bh_next(size_t b, size_t num_vars);
size_t bh_nth(size_t num_vars, size_t i);

// rotation helpers of poly/multilinear.rs:477-549 (hyperplonk.cpp)
std::vector<size_t> rotation_point_pattern(bool next, size_t num_vars, size_t distance);
std::vector<std::vector<HFr>> rotation_eval_points(const std::vector<HFr>& x, int rotation);

}  // namespace lh
