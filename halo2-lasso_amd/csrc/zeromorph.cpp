// Zeromorph over univariate KZG: prover half (reference pcs/multilinear/zeromorph.rs:86-213,258-296 and
// pcs/univariate/kzg.rs:23-36,175-299).  The multilinear table is committed as a coefficient vector against the
// powers of s; an opening is n quotient commitments (one batched MSM), the commitment of the shifted combination
// q_hat, and one univariate KZG opening of the degree-(2^n - 1) polynomial f at x (quotient by X - x as a suffix
// Horner scan on the device).  Restated in oracle/pyref/zeromorph.py, which the tests compare against byte for byte.
#include "host.hpp"

namespace lh {

USrs* ukzg_setup(Ctx& c, const HFr& s, size_t poly_size) {
  LH_REQUIRE(poly_size >= 1 && poly_size < ((size_t)1 << 31), LH_ERR_ARG, "univariate setup: bad poly_size");
  USrs* srs = new USrs();
  srs->size = poly_size;
  LH_HIP(hipMalloc((void**)&srs->d_powers, poly_size * sizeof(G1Affine)));
  ArenaScope scope(c.arena);
  Fr* scal = c.arena.alloc_n<Fr>(poly_size);
  k_powers(c, dev(s), poly_size, scal);
  k_fixed_base_mul_g(c, scal, poly_size, srs->d_powers);
  return srs;
}

static void check_degree(const USrs& srs, size_t poly_size, size_t num_vars, const char* what) {
  LH_REQUIRE(poly_size >= 1 && poly_size <= srs.size, LH_ERR_INVALID_PCS_PARAM, "Too large poly_size to trim to");
  if (num_vars >= 63 || ((size_t)1 << num_vars) > poly_size)  // pp.degree() + 1 < poly.evals().len() (zeromorph.rs:111,144)
    throw Error(LH_ERR_INVALID_PCS_PARAM, std::string("Too large degree of poly to ") + what +
                                              " (param supports degree up to " + std::to_string(poly_size - 1) + ")");
}

std::vector<HG1> zeromorph_batch_commit(Ctx& c, const USrs& srs, size_t poly_size, const Fr* const* d_polys,
                                        size_t num_polys, size_t num_vars) {
  check_degree(srs, poly_size, num_vars, "commit");
  std::vector<MsmJob> jobs(num_polys);
  for (size_t i = 0; i < num_polys; i++) jobs[i] = MsmJob{d_polys[i], false, srs.d_powers, (size_t)1 << num_vars};
  std::vector<HG1> out(num_polys);
  if (num_polys) msm_batch(c, jobs.data(), num_polys, (G1Affine*)out.data());
  return out;
}

std::pair<HFr, std::vector<HFr>> zeromorph_scalars(const HFr& y, const HFr& x, const HFr& z, const HFr* u, size_t n) {
  std::vector<HFr> squares(n + 1), offsets(n), denoms(n + 1), vs(n + 1), q_scalars(n);
  squares[0] = x;
  for (size_t i = 0; i < n; i++) squares[i + 1] = squares[i].sqr();
  HFr state = HFr::one();
  for (size_t i = n; i-- > 0;) {  // squares.rev().skip(1) scanned, then reversed
    state *= squares[i];
    offsets[i] = state;
  }
  const HFr one = HFr::one(), v_numer = squares[n] - one;
  for (size_t i = 0; i <= n; i++) vs[i] = v_numer * (squares[i] - one).inv();
  HFr power_of_y = one;
  for (size_t i = 0; i < n; i++) {
    q_scalars[i] = -(power_of_y * offsets[i] + z * (squares[i] * vs[i + 1] - u[i] * vs[i]));
    power_of_y *= y;
  }
  return {-(vs[0] * z), q_scalars};
}

void zeromorph_open(Ctx& c, const USrs& srs, size_t poly_size, const Fr* d_poly, size_t num_vars, const HFr* point,
                    Transcript& tr) {
  check_degree(srs, poly_size, num_vars, "open");
  LH_REQUIRE(num_vars >= 1 && num_vars < (size_t)32, LH_ERR_ARG, "zeromorph open: bad num_vars");
  const size_t n = (size_t)1 << num_vars, offset = srs.size - poly_size;
  ArenaScope scope(c.arena);
  // quotients (pcs/multilinear.rs:72-107), flat: q_k at 2^k - 1
  Fr* q = c.arena.alloc_n<Fr>(n);
  Fr* remA = c.arena.alloc_n<Fr>(std::max<size_t>(n >> 1, 1));
  Fr* remB = c.arena.alloc_n<Fr>(std::max<size_t>(n >> 2, 1));
  const Fr* rem = d_poly;
  for (size_t i = num_vars; i-- > 0;) {
    const size_t half = (size_t)1 << i;
    Fr* dst = ((num_vars - i) & 1) ? remA : remB;
    k_quotient_step(c, rem, half, dev(point[i]), q + (half - 1), dst);
    rem = dst;
  }
  {
    std::vector<MsmJob> jobs(num_vars);
    for (size_t i = 0; i < num_vars; i++) jobs[i] = MsmJob{q + (((size_t)1 << i) - 1), false, srs.d_powers, (size_t)1 << i};
    std::vector<HG1> comms(num_vars);
    msm_batch(c, jobs.data(), num_vars, (G1Affine*)comms.data());
    tr.write_commitments(comms);
  }
  const HFr y = tr.squeeze_challenge();
  std::vector<Fr> ypow(num_vars);
  {
    HFr p = HFr::one();
    for (size_t k = 0; k < num_vars; k++) {
      ypow[k] = dev(p);
      p *= y;
    }
  }
  Fr* q_hat = c.arena.alloc_n<Fr>(n);
  k_zm_qhat(c, q, num_vars, ypow.data(), q_hat);
  {
    MsmJob job{q_hat, false, srs.d_powers, n};
    HG1 comm;
    msm_batch(c, &job, 1, (G1Affine*)&comm);
    tr.write_commitment(comm);
  }
  const HFr x = tr.squeeze_challenge(), z = tr.squeeze_challenge();
  auto sc = zeromorph_scalars(y, x, z, point, num_vars);
  std::vector<Fr> qs(num_vars);
  for (size_t k = 0; k < num_vars; k++) qs[k] = dev(sc.second[k]);
  // f = z poly + q_hat + sum_k q_scalar_k q_k (+ a constant, eval_scalar * eval, which the quotient does not see)
  Fr* f = c.arena.alloc_n<Fr>(n);
  k_zm_combine(c, d_poly, q_hat, q, num_vars, dev(z), qs.data(), f);
  // UnivariateKzg::open at x with open_pp = powers[offset..]: quotient[i] = S_{i+1}
  Fr* S = c.arena.alloc_n<Fr>(n);
  k_suffix_horner(c, f, n, dev(x), S);
  MsmJob job{S + 1, false, srs.d_powers + offset, n - 1};
  HG1 pi;
  msm_batch(c, &job, 1, (G1Affine*)&pi);
  tr.write_commitment(pi);
}

void zeromorph_batch_open(Ctx& c, const USrs& srs, size_t poly_size, size_t num_vars, const Fr* const* d_polys,
                          size_t num_polys, const HFr* points, size_t num_points, const lh_evaluation* evals,
                          size_t num_evals, Transcript& tr, const SmallPoly* small) {
  check_degree(srs, poly_size, num_vars, "open");
  additive_batch_open(
      c, num_vars, d_polys, num_polys, points, num_points, evals, num_evals, tr,
      [&](const Fr* g_prime, const HFr* point) { zeromorph_open(c, srs, poly_size, g_prime, num_vars, point, tr); }, small);
}

}  // namespace lh
