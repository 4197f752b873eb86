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

uint32_t bh_primitive(size_t num_vars);
uint32_t bh_x_inv(size_t num_vars);
size_t bh_next(size_t b, size_t num_vars);
size_t bh_nth(size_t num_vars, size_t i);

// rotation helpers of poly/multilinear.rs:477-549 (hyperplonk.cpp)
std::vector<size_t> rotation_point_pattern(bool next, size_t num_vars, size_t distance);
std::vector<std::vector<HFr>> rotation_eval_points(const std::vector<HFr>& x, int rotation);

}  // namespace lh
