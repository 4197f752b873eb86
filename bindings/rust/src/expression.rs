//! `Expression<Fr>` (util/expression.rs:67-78) -> flat lh_expr nodes: every node refers to EARLIER nodes, the root is
//! the last one.  NEVER COMPILED - see README.md.
use crate::sys::*;
use halo2_curves::bn256::Fr;
use halo2_curves::ff::Field;
use plonkish_backend::util::expression::{CommonPolynomial, Expression};

fn node(op: u32, a: i32, b: i32, scalar: Fr) -> lh_expr_node {
    lh_expr_node { op, a, b, reserved: 0, scalar }
}

fn walk(e: &Expression<Fr>, out: &mut Vec<lh_expr_node>) -> i32 {
    let z = Fr::ZERO;
    let n = match e {
        Expression::Constant(c) => node(LH_EX_CONSTANT, 0, 0, *c),
        Expression::CommonPolynomial(CommonPolynomial::Identity) => node(LH_EX_IDENTITY, 0, 0, z),
        Expression::CommonPolynomial(CommonPolynomial::Lagrange(i)) => node(LH_EX_LAGRANGE, *i, 0, z),
        Expression::CommonPolynomial(CommonPolynomial::EqXY(i)) => node(LH_EX_EQ_XY, *i as i32, 0, z),
        Expression::Polynomial(q) => node(LH_EX_POLYNOMIAL, q.poly() as i32, q.rotation().0, z),
        Expression::Challenge(i) => node(LH_EX_CHALLENGE, *i as i32, 0, z),
        Expression::Negated(a) => {
            let a = walk(a, out);
            node(LH_EX_NEGATED, a, 0, z)
        }
        Expression::Sum(a, b) => {
            let (a, b) = (walk(a, out), walk(b, out));
            node(LH_EX_SUM, a, b, z)
        }
        Expression::Product(a, b) => {
            let (a, b) = (walk(a, out), walk(b, out));
            node(LH_EX_PRODUCT, a, b, z)
        }
        Expression::Scaled(a, s) => {
            let a = walk(a, out);
            node(LH_EX_SCALED, a, 0, *s)
        }
        // e0 + base*e1 + base^2*e2 + ... exactly as Expression::evaluate lowers it (expression.rs:155-167)
        Expression::DistributePowers(es, base) => {
            let base_id = walk(base, out);
            let mut acc = walk(&es[0], out);
            let mut power = base_id;
            for (k, e) in es.iter().enumerate().skip(1) {
                if k > 1 {
                    out.push(node(LH_EX_PRODUCT, power, base_id, z));
                    power = out.len() as i32 - 1;
                }
                let e = walk(e, out);
                out.push(node(LH_EX_PRODUCT, e, power, z));
                let term = out.len() as i32 - 1;
                out.push(node(LH_EX_SUM, acc, term, z));
                acc = out.len() as i32 - 1;
            }
            return acc;
        }
    };
    out.push(n);
    out.len() as i32 - 1
}

/// keep the Vec alive as long as the lh_expr is in use
pub fn flatten(e: &Expression<Fr>) -> (Vec<lh_expr_node>, lh_expr) {
    let mut nodes = Vec::new();
    walk(e, &mut nodes);
    let ex = lh_expr { nodes: nodes.as_ptr(), num_nodes: nodes.len() };
    (nodes, ex)
}
