// TEST INFRASTRUCTURE ONLY - multithreaded CPU restatement of the reference prover hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
// product (halo2-lasso_amd/) never links or calls it.  It follows the reference's ALGORITHMS and its
// chunk-per-thread parallelism (plonkish_backend/src/util/parallel.rs:27-46), so that timing it on
// the GPU box's host cores is a fair stand-in for the reference's rayon path, which cannot be built
// here (no Rust toolchain, un-vendored git dependencies: SURVEY.md §8c).  PARITY UNPINNED against the
// reference's bytes; pinned against oracle/pyref through tests/golden/vectors.json.
//
// Restated routines (reference file:line):
//   Keccak256Transcript            util/transcript.rs:101-238, util/hash.rs:19-21
//   eq_xy / fix_var / evaluate     poly/multilinear.rs:91-127, 179-189, 599-618
//   ClassicSumCheck::prove         piop/sum_check/classic.rs:208-240
//   EvaluationsProver::evals       piop/sum_check/classic/eval.rs:102-131 (evaluate, THEN bind: classic.rs:90-141)
//   CoefficientsProver::karatsuba  piop/sum_check/classic/coeff.rs:153-202
//   prove_fractional_sum_check     piop/gkr/fractional_sum_check.rs:62-190
//   variable_base_msm              util/arithmetic/msm.rs:84-181 (Pippenger per thread chunk, window = floor(ln n))
//   MultilinearKzg commit/open     pcs/multilinear/kzg.rs:252-302, quotients pcs/multilinear.rs:72-107
//   additive::batch_open           pcs/multilinear.rs:134-235
//   Lasso                          no reference code; spec = oracle/pyref/lasso.py
//   HyperPlonk::prove + LogUp      backend/hyperplonk.rs:164-291, hyperplonk/prover.rs:32-406 (see the section below)
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <stdexcept>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "ff.hpp"

using namespace orc;

// ------------------------------------------------------------------ threads (util/parallel.rs)
static int g_threads = 0;
static int num_threads() {
  if (g_threads <= 0) {
    g_threads = (int)std::thread::hardware_concurrency();
    if (g_threads <= 0) g_threads = 1;
  }
  return g_threads;
}
// parallelize (parallel.rs:27-46): contiguous chunks, serial when chunk_size < num_threads.  The chunks run on a
// persistent pool (rayon keeps its workers alive too: spawning std::threads per call cost more than the work of the
// small calls - thousands per proof - and put the CPU baseline at a disadvantage it does not have in the reference).
namespace {
struct Pool {
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  const std::function<void(size_t, size_t)>* fn = nullptr;
  size_t n = 0, chunk = 0, next = 0, pending = 0;
  uint64_t generation = 0;
  bool stop = false;
  static thread_local bool inside;
  explicit Pool(size_t nt) {
    for (size_t i = 0; i + 1 < nt; i++) workers.emplace_back([this] { run(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : workers) t.join();
  }
  bool take(size_t& s, size_t& e) {  // (mu held)
    if (!fn || next >= n) return false;
    s = next, e = std::min(n, next + chunk);
    next = e;
    return true;
  }
  void run() {
    inside = true;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      size_t s, e;
      if (take(s, e)) {
        const auto* f = fn;
        lk.unlock();
        (*f)(s, e);
        lk.lock();
        if (--pending == 0) cv_done.notify_all();
        continue;
      }
      if (stop) return;
      cv_work.wait(lk);
    }
  }
  void parallel(size_t n_, size_t chunk_, const std::function<void(size_t, size_t)>& f) {
    std::unique_lock<std::mutex> lk(mu);
    fn = &f, n = n_, chunk = chunk_, next = 0, pending = (n_ + chunk_ - 1) / chunk_;
    generation++;
    cv_work.notify_all();
    // the caller works too - as an insider: a chunk that parallelizes again must run that part inline (the region's
    // mutex is not recursive); a chunk that throws on this thread must not leave workers with a dangling `fn`
    struct Inside {
      bool was = inside;
      Inside() { inside = true; }
      ~Inside() { inside = was; }
    } mark;
    size_t s, e;
    try {
      while (take(s, e)) {
        lk.unlock();
        f(s, e);
        lk.lock();
        --pending;
      }
    } catch (...) {
      if (!lk.owns_lock()) lk.lock();
      --pending;                 // the chunk that threw
      pending -= (n - next + chunk - 1) / chunk;  // chunks nobody has taken: withdrawn
      next = n;
      cv_done.wait(lk, [this] { return pending == 0; });  // chunks in flight on workers finish first
      fn = nullptr;
      throw;
    }
    cv_done.wait(lk, [this] { return pending == 0; });
    fn = nullptr;
  }
};
thread_local bool Pool::inside = false;
}  // namespace
static void parallelize(size_t n, const std::function<void(size_t, size_t)>& f) {
  size_t nt = (size_t)num_threads();
  size_t chunk = (n + nt - 1) / nt;
  if (nt == 1 || chunk < nt || Pool::inside) {  // (a chunk that parallelizes again runs that part inline)
    f(0, n);
    return;
  }
  static Pool* pool = nullptr;
  static size_t pool_threads = 0;
  static std::mutex pool_mu;  // one parallel region at a time (the oracle is driven from one thread)
  std::lock_guard<std::mutex> guard(pool_mu);
  if (!pool || pool_threads != nt) {
    delete pool;
    pool = new Pool(nt);
    pool_threads = nt;
  }
  pool->parallel(n, chunk, f);
}

// ------------------------------------------------------------------ Keccak-256 (sha3 0.10.6 Keccak256)
struct Keccak {
  uint64_t st[25];
  uint8_t buf[136];
  size_t len;
  Keccak() { reset(); }
  void reset() {
    memset(st, 0, sizeof st);
    len = 0;
  }
  static uint64_t rol(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
  void permute() {
    static const uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    // rho offsets by lane position in pi order (piln walk of the Keccak team's compact implementation)
    static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; r++) {
      uint64_t bc[5];
      for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
      for (int i = 0; i < 5; i++) {
        uint64_t t = bc[(i + 4) % 5] ^ rol(bc[(i + 1) % 5], 1);
        for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
      }
      uint64_t t = st[1];
      for (int i = 0; i < 24; i++) {
        int j = PILN[i];
        uint64_t b = st[j];
        st[j] = rol(t, ROTC[i]);
        t = b;
      }
      for (int j = 0; j < 25; j += 5) {
        for (int i = 0; i < 5; i++) bc[i] = st[j + i];
        for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
      }
      st[0] ^= RC[r];
    }
  }
  void absorb() {
    for (int i = 0; i < 17; i++) {
      uint64_t w;
      memcpy(&w, buf + 8 * i, 8);
      st[i] ^= w;
    }
    permute();
    len = 0;
  }
  void update(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      buf[len++] = d[i];
      if (len == 136) absorb();
    }
  }
  void finalize_reset(uint8_t out[32]) {
    memset(buf + len, 0, 136 - len);
    buf[len] ^= 0x01;
    buf[135] ^= 0x80;
    absorb();
    memcpy(out, st, 32);
    reset();
  }
};

struct OracleError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct Transcript {
  Keccak h;
  std::vector<uint8_t> stream;
  void common_fe(const Fr& f) {
    uint64_t c[4];
    f.to_raw(c);
    h.update((const uint8_t*)c, 32);
  }
  void write_fe(const Fr& f) {
    uint64_t c[4];
    f.to_raw(c);
    h.update((const uint8_t*)c, 32);
    const uint8_t* b = (const uint8_t*)c;
    for (int i = 31; i >= 0; i--) stream.push_back(b[i]);
  }
  void write_fes(const std::vector<Fr>& v) {
    for (auto& f : v) write_fe(f);
  }
  Fr squeeze() {
    uint8_t d[32];
    h.finalize_reset(d);
    h.update(d, 32);
    uint64_t c[4];
    memcpy(c, d, 32);
    while (Fr::ge_mod(c)) Fr::sub_mod_inplace(c);  // fe_mod_from_le_bytes (arithmetic.rs:150-152)
    return Fr::from_raw(c);
  }
  std::vector<Fr> squeeze_n(size_t n) {
    std::vector<Fr> v(n);
    for (auto& f : v) f = squeeze();
    return v;
  }
  void write_comm(const Affine& p) {
    if (p.is_identity()) throw OracleError("Invalid elliptic curve point encoding");
    uint64_t cx[4], cy[4];
    p.x.to_raw(cx);
    p.y.to_raw(cy);
    h.update((const uint8_t*)cx, 32);
    h.update((const uint8_t*)cy, 32);
    for (int i = 31; i >= 0; i--) stream.push_back(((const uint8_t*)cx)[i]);
    for (int i = 31; i >= 0; i--) stream.push_back(((const uint8_t*)cy)[i]);
  }
};

// ------------------------------------------------------------------ multilinear polys
typedef std::vector<Fr> Poly;

static Poly eq_xy(const Fr* y, size_t n) {  // multilinear.rs:91-127
  Poly ev{Fr::one()};
  for (size_t i = n; i-- > 0;) {
    Poly nx(ev.size() * 2);
    const Fr yi = y[i];
    parallelize(ev.size(), [&](size_t s, size_t e) {
      for (size_t k = s; k < e; k++) {
        nx[2 * k + 1] = ev[k] * yi;
        nx[2 * k] = ev[k] - nx[2 * k + 1];
      }
    });
    ev.swap(nx);
  }
  return ev;
}
static Poly fix_var(const Poly& p, const Fr& x) {  // merge_into, multilinear.rs:599-618
  Poly out(p.size() / 2);
  parallelize(out.size(), [&](size_t s, size_t e) {
    for (size_t b = s; b < e; b++) out[b] = (p[2 * b + 1] - p[2 * b]) * x + p[2 * b];
  });
  return out;
}
static Fr evaluate(const Poly& p, const Fr* x, size_t n) {
  Poly cur = p;
  for (size_t i = 0; i < n; i++) cur = fix_var(cur, x[i]);
  return cur[0];
}
static Fr eq_xy_eval(const Fr* x, const Fr* y, size_t n) {  // sum_check.rs:112-121
  Fr acc = Fr::one();
  for (size_t i = 0; i < n; i++) acc = acc * ((x[i] * y[i]).dbl() + Fr::one() - x[i] - y[i]);
  return acc;
}

// ------------------------------------------------------------------ sum-check over [eq *] sum_m c_m prod_k T
struct Sop {  // same layout as lh_sop (include/lasso_hip.h)
  uint32_t num_terms;
  int32_t global_eq;
  Fr coeff[48];
  uint8_t num_factors[48];
  uint8_t factor[48][4];
};

static Fr interpolate(const std::vector<Fr>& ev, const Fr& x) {  // barycentric over 0..d (arithmetic.rs:108-136)
  size_t d = ev.size() - 1;
  std::vector<Fr> pts(d + 1);
  for (size_t i = 0; i <= d; i++) pts[i] = Fr::from_u64(i);
  for (size_t i = 0; i <= d; i++)
    if (x == pts[i]) return ev[i];
  Fr tot = Fr::zero();
  for (size_t j = 0; j <= d; j++) {
    Fr nu = Fr::one(), de = Fr::one();
    for (size_t i = 0; i <= d; i++)
      if (i != j) {
        nu = nu * (x - pts[i]);
        de = de * (pts[j] - pts[i]);
      }
    tot = tot + ev[j] * nu * de.inv();
  }
  return tot;
}

struct ScOut {
  std::vector<Fr> x, evals;
};

static ScOut sum_check_prove(Transcript& tr, int kind, size_t nv, const Sop& e, std::vector<Poly> polys,
                             const std::vector<std::vector<Fr>>& ys, Fr claim) {
  if (nv == 0) throw OracleError("sum-check needs num_vars > 0");
  size_t np = polys.size();
  std::vector<Poly> tabs = std::move(polys);
  for (auto& y : ys) tabs.push_back(eq_xy(y.data(), nv));
  int geq = e.global_eq >= 0 ? (int)np + e.global_eq : -1;
  int degree = 0;
  for (uint32_t m = 0; m < e.num_terms; m++) degree = std::max(degree, (int)e.num_factors[m] + (geq >= 0 ? 1 : 0));
  if (kind == 1 && degree != 2) throw OracleError("CoefficientsProver: degree 2 only");
  const Fr inv2 = Fr::from_u64(2).inv();
  ScOut out;
  for (size_t round = 0; round < nv; round++) {
    size_t size = (size_t)1 << (nv - round - 1);
    int nt = num_threads();
    size_t chunk = (size + nt - 1) / nt;
    size_t nchunks = (size + chunk - 1) / chunk;
    std::vector<std::vector<Fr>> partial(nchunks, std::vector<Fr>(degree + 1, Fr::zero()));
    auto work = [&](size_t ci) {  // EvaluationsProver::evals per-thread partials (eval.rs:105-125)
      std::vector<Fr>& acc = partial[ci];
      size_t lo = ci * chunk, hi = std::min(size, lo + chunk);
      std::vector<Fr> s(degree + 1), pm(degree + 1);
      for (size_t b = lo; b < hi; b++) {
        for (int x = 1; x <= degree; x++) s[x] = Fr::zero();
        for (uint32_t m = 0; m < e.num_terms; m++) {
          for (int k = 0; k < e.num_factors[m]; k++) {
            const Poly& t = tabs[e.factor[m][k]];
            Fr v1 = t[2 * b + 1], step = v1 - t[2 * b], val = v1;
            for (int x = 1; x <= degree; x++) {
              if (x > 1) val = val + step;
              pm[x] = k == 0 ? e.coeff[m] * val : pm[x] * val;
            }
          }
          for (int x = 1; x <= degree; x++) s[x] = s[x] + pm[x];
        }
        if (geq >= 0) {
          const Poly& t = tabs[geq];
          Fr v1 = t[2 * b + 1], step = v1 - t[2 * b], val = v1;
          for (int x = 1; x <= degree; x++) {
            if (x > 1) val = val + step;
            s[x] = s[x] * val;
          }
        }
        for (int x = 1; x <= degree; x++) acc[x] = acc[x] + s[x];
      }
    };
    if (nchunks == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (size_t ci = 0; ci < nchunks; ci++) th.emplace_back(work, ci);
      for (auto& t : th) t.join();
    }
    std::vector<Fr> ev(degree + 1, Fr::zero());
    for (auto& p : partial)
      for (int x = 1; x <= degree; x++) ev[x] = ev[x] + p[x];
    ev[0] = claim - ev[1];  // eval.rs:129
    Fr r;
    if (kind == 1) {  // coeff.rs:136-149
      std::vector<Fr> co(3);
      co[0] = ev[0];
      co[2] = (ev[2] - ev[1].dbl() + ev[0]) * inv2;
      co[1] = claim - (co[0].dbl() + co[2]);
      tr.write_fes(co);
      r = tr.squeeze();
      claim = (co[2] * r + co[1]) * r + co[0];
    } else {
      tr.write_fes(ev);
      r = tr.squeeze();
      claim = interpolate(ev, r);
    }
    out.x.push_back(r);
    for (auto& t : tabs) t = fix_var(t, r);  // ProverState::next_round (classic.rs:90-141)
  }
  for (size_t i = 0; i < np; i++) out.evals.push_back(tabs[i][0]);
  return out;
}

// ------------------------------------------------------------------ GKR
struct FracOut {
  std::vector<Fr> p_xs, q_xs, x;
};
static FracOut frac_gkr_prove(Transcript& tr, const std::vector<Poly>& ps, const std::vector<Poly>& qs) {
  size_t B = ps.size();
  size_t nv = 0;
  while (((size_t)1 << nv) < ps[0].size()) nv++;
  // levels[h] = (p, q) vectors of 2^(nv-h) entries; Layer::bottom / up (fractional_sum_check.rs:42-85)
  std::vector<std::vector<Poly>> lp(nv), lq(nv);
  lp[0] = ps, lq[0] = qs;
  for (size_t h = 1; h < nv; h++) {
    size_t half = (size_t)1 << (nv - h);
    for (size_t b = 0; b < B; b++) {
      Poly vp(half), vq(half);
      const Poly &p = lp[h - 1][b], &q = lq[h - 1][b];
      parallelize(half, [&](size_t s, size_t e) {
        for (size_t i = s; i < e; i++) {
          vp[i] = p[i] * q[half + i] + p[half + i] * q[i];
          vq[i] = q[i] * q[half + i];
        }
      });
      lp[h].push_back(vp), lq[h].push_back(vq);
    }
  }
  std::vector<Fr> cp(B), cq(B);
  for (size_t b = 0; b < B; b++) {
    const Poly &p = lp[nv - 1][b], &q = lq[nv - 1][b];
    cp[b] = p[0] * q[1] + p[1] * q[0];
    cq[b] = q[0] * q[1];
  }
  tr.write_fes(cp);
  tr.write_fes(cq);
  std::vector<Fr> y;
  for (size_t h = nv; h-- > 0;) {
    size_t m = nv - 1 - h, half = (size_t)1 << m;
    std::vector<Fr> x, evals;
    if (m == 0) {
      for (size_t b = 0; b < B; b++) {
        evals.push_back(lp[h][b][0]), evals.push_back(lp[h][b][1]);
        evals.push_back(lq[h][b][0]), evals.push_back(lq[h][b][1]);
      }
    } else {
      Fr gamma = tr.squeeze(), claim = Fr::zero(), pw = Fr::one();
      Sop e;
      memset(&e, 0, sizeof e);
      e.global_eq = 0;
      std::vector<Poly> polys;
      for (size_t b = 0; b < B; b++) {
        claim = claim + cp[b] * pw;
        Fr ge = pw;
        pw = pw * gamma;
        claim = claim + cq[b] * pw;
        Fr go = pw;
        pw = pw * gamma;
        uint32_t t = e.num_terms;
        e.coeff[t] = ge, e.num_factors[t] = 2, e.factor[t][0] = 4 * b, e.factor[t][1] = 4 * b + 3;
        e.coeff[t + 1] = ge, e.num_factors[t + 1] = 2, e.factor[t + 1][0] = 4 * b + 1, e.factor[t + 1][1] = 4 * b + 2;
        e.coeff[t + 2] = go, e.num_factors[t + 2] = 2, e.factor[t + 2][0] = 4 * b + 2, e.factor[t + 2][1] = 4 * b + 3;
        e.num_terms += 3;
        const Poly &p = lp[h][b], &q = lq[h][b];
        polys.emplace_back(p.begin(), p.begin() + half), polys.emplace_back(p.begin() + half, p.end());
        polys.emplace_back(q.begin(), q.begin() + half), polys.emplace_back(q.begin() + half, q.end());
      }
      ScOut sc = sum_check_prove(tr, 0, m, e, polys, {y}, claim);
      x = sc.x, evals = sc.evals;
    }
    tr.write_fes(evals);
    Fr mu = tr.squeeze();
    for (size_t b = 0; b < B; b++) {
      cp[b] = evals[4 * b] + mu * (evals[4 * b + 1] - evals[4 * b]);
      cq[b] = evals[4 * b + 2] + mu * (evals[4 * b + 3] - evals[4 * b + 2]);
    }
    x.push_back(mu);
    y = x;
  }
  return FracOut{cp, cq, y};
}

struct GpOut {
  std::vector<Fr> roots, claims;
  std::vector<std::vector<Fr>> points;
};
static GpOut grand_product_prove(Transcript& tr, const std::vector<Poly>& leaves) {
  size_t B = leaves.size(), maxd = 0;
  std::vector<size_t> depth(B);
  std::vector<std::vector<Poly>> level(B);  // level[b][h]: 2^(h+1) nodes
  for (size_t b = 0; b < B; b++) {
    size_t d = 0;
    while (((size_t)1 << d) < leaves[b].size()) d++;
    depth[b] = d, maxd = std::max(maxd, d);
    level[b].resize(d);
    level[b][d - 1] = leaves[b];
    for (size_t h = d - 1; h-- > 0;) {
      size_t half = (size_t)1 << (h + 1);
      Poly up(half);
      const Poly& in = level[b][h + 1];
      parallelize(half, [&](size_t s, size_t e) {
        for (size_t i = s; i < e; i++) up[i] = in[i] * in[half + i];
      });
      level[b][h] = up;
    }
  }
  GpOut out;
  out.roots.resize(B), out.claims.resize(B), out.points.resize(B);
  for (size_t b = 0; b < B; b++) out.roots[b] = level[b][0][0] * level[b][0][1];
  tr.write_fes(out.roots);
  std::vector<Fr> claims = out.roots, y;
  for (size_t h = 0; h < maxd; h++) {
    std::vector<size_t> act;
    for (size_t b = 0; b < B; b++)
      if (depth[b] > h) act.push_back(b);
    size_t half = (size_t)1 << h;
    std::vector<Fr> x, evals;
    if (h == 0) {
      for (size_t b : act) evals.push_back(level[b][0][0]), evals.push_back(level[b][0][1]);
    } else {
      Fr lam = tr.squeeze(), claim = Fr::zero(), pw = Fr::one();
      Sop e;
      memset(&e, 0, sizeof e);
      e.global_eq = 0;
      e.num_terms = (uint32_t)act.size();
      std::vector<Poly> polys;
      for (size_t k = 0; k < act.size(); k++) {
        size_t b = act[k];
        claim = claim + claims[b] * pw;
        e.coeff[k] = pw, e.num_factors[k] = 2, e.factor[k][0] = 2 * k, e.factor[k][1] = 2 * k + 1;
        pw = pw * lam;
        const Poly& v = level[b][h];
        polys.emplace_back(v.begin(), v.begin() + half), polys.emplace_back(v.begin() + half, v.end());
      }
      ScOut sc = sum_check_prove(tr, 0, h, e, polys, {y}, claim);
      x = sc.x, evals = sc.evals;
    }
    tr.write_fes(evals);
    Fr mu = tr.squeeze();
    x.push_back(mu);
    y = x;
    for (size_t k = 0; k < act.size(); k++) {
      size_t b = act[k];
      claims[b] = evals[2 * k] + mu * (evals[2 * k + 1] - evals[2 * k]);
      if (depth[b] == h + 1) out.claims[b] = claims[b], out.points[b] = y;
    }
  }
  return out;
}

// ------------------------------------------------------------------ variable_base_msm (msm.rs:84-181)
static size_t window_size(size_t n) { return n < 32 ? 3 : (size_t)floor(log((double)n)); }

static Jac msm_serial(const Fr* scalars, const Affine* bases, size_t n) {
  std::vector<uint64_t> repr(4 * n);
  for (size_t i = 0; i < n; i++) scalars[i].to_raw(&repr[4 * i]);
  size_t w = window_size(n), nb = ((size_t)1 << w) - 1, nwin = (256 + w - 1) / w;
  Jac result = Jac::identity();
  std::vector<Jac> buckets(nb);
  std::vector<uint8_t> used(nb);
  for (size_t idx = nwin; idx-- > 0;) {
    for (size_t k = 0; k < w; k++) result = jac_dbl(result);
    std::fill(used.begin(), used.end(), 0);
    size_t skip = idx * w;
    for (size_t i = 0; i < n; i++) {
      const uint64_t* r = &repr[4 * i];
      size_t limb = skip / 64, off = skip % 64;
      uint64_t d = limb < 4 ? r[limb] >> off : 0;
      if (off && limb + 1 < 4) d |= r[limb + 1] << (64 - off);
      d &= nb;
      if (d) {
        if (!used[d - 1]) {
          buckets[d - 1] = jac_from_affine(bases[i]);
          used[d - 1] = 1;
        } else {
          buckets[d - 1] = jac_add_affine(buckets[d - 1], bases[i]);
        }
      }
    }
    Jac run = Jac::identity();
    for (size_t b = nb; b-- > 0;) {
      if (used[b]) run = jac_add(run, buckets[b]);
      result = jac_add(result, run);
    }
  }
  return result;
}

static Affine msm(const Fr* scalars, const Affine* bases, size_t n) {
  size_t nt = (size_t)num_threads();
  if (n == 0) return Affine{Fq::zero(), Fq::zero()};
  if (n <= nt) return jac_to_affine(msm_serial(scalars, bases, n));
  size_t chunk = (n + nt - 1) / nt;
  size_t nch = (n + chunk - 1) / chunk;
  std::vector<Jac> res(nch, Jac::identity());
  std::vector<std::thread> th;
  for (size_t c = 0; c < nch; c++)
    th.emplace_back([&, c] {
      size_t lo = c * chunk, hi = std::min(n, lo + chunk);
      res[c] = msm_serial(scalars + lo, bases + lo, hi - lo);
    });
  for (auto& t : th) t.join();
  Jac acc = Jac::identity();
  for (auto& r : res) acc = jac_add(acc, r);
  return jac_to_affine(acc);
}

// ------------------------------------------------------------------ MultilinearKzg (flat SRS: level k at 2^k - 1)
struct Srs {
  const Affine* eqs;
  size_t nv;
  const Affine* eq(size_t k) const { return eqs + (((size_t)1 << k) - 1); }
};
static Affine commit(const Srs& s, const Poly& p) {
  size_t nv = 0;
  while (((size_t)1 << nv) < p.size()) nv++;
  if (nv > s.nv) throw OracleError("Too many variates of poly to commit");
  return msm(p.data(), s.eq(nv), p.size());
}
static Fr kzg_open(Transcript& tr, const Srs& s, const Poly& poly, const Fr* point, size_t nv) {
  Poly rem = poly;  // quotients (pcs/multilinear.rs:72-107)
  std::vector<Affine> comms(nv);
  for (size_t i = nv; i-- > 0;) {
    size_t half = (size_t)1 << i;
    Poly q(half), lo(half);
    parallelize(half, [&](size_t a, size_t b) {
      for (size_t k = a; k < b; k++) {
        q[k] = rem[half + k] - rem[k];
        lo[k] = rem[k] + q[k] * point[i];
      }
    });
    rem.swap(lo);
    comms[i] = msm(q.data(), s.eq(i), half);
  }
  for (auto& c : comms) tr.write_comm(c);
  return rem[0];
}
struct Eval {  // same layout as lh_evaluation
  uint32_t poly, point;
  Fr value;
};
// additive::batch_open (pcs/multilinear.rs:134-235), generic over the PCS through `open`
static void batch_open_with(Transcript& tr, size_t nv, const std::vector<const Poly*>& polys,
                            const std::vector<std::vector<Fr>>& points, const std::vector<Eval>& evals,
                            const std::function<void(const Poly&, const Fr*)>& open) {
  if (evals.size() < 2) throw OracleError("batch open needs >= 2 evaluations");
  size_t ell = 0;
  while (((size_t)1 << ell) < evals.size()) ell++;
  std::vector<Fr> t = tr.squeeze_n(ell);
  Poly eq_xt = eq_xy(t.data(), ell);
  size_t n = (size_t)1 << nv, np = points.size();
  std::vector<Poly> merged(np, Poly(n, Fr::zero()));  // pcs/multilinear.rs:155-170
  for (size_t i = 0; i < evals.size(); i++) {
    Poly& m = merged[evals[i].point];
    const Poly& src = *polys[evals[i].poly];
    const Fr w = eq_xt[i];
    parallelize(n, [&](size_t a, size_t b) {
      for (size_t k = a; k < b; k++) m[k] = m[k] + w * src[k];
    });
  }
  Sop e;
  memset(&e, 0, sizeof e);
  e.global_eq = -1;
  e.num_terms = (uint32_t)np;
  for (size_t j = 0; j < np; j++) {
    e.coeff[j] = Fr::one();
    e.num_factors[j] = 2;
    e.factor[j][0] = (uint8_t)(np + j);
    e.factor[j][1] = (uint8_t)j;
  }
  Fr sum = Fr::zero();
  for (size_t i = 0; i < evals.size(); i++) sum = sum + evals[i].value * eq_xt[i];
  ScOut sc = sum_check_prove(tr, 1, nv, e, merged, points, sum);
  Poly g(n, Fr::zero());
  for (size_t j = 0; j < np; j++) {
    Fr w = eq_xy_eval(sc.x.data(), points[j].data(), nv);
    parallelize(n, [&](size_t a, size_t b) {
      for (size_t k = a; k < b; k++) g[k] = g[k] + w * merged[j][k];
    });
  }
  open(g, sc.x.data());
}
static void batch_open(Transcript& tr, const Srs& s, size_t nv, const std::vector<const Poly*>& polys,
                       const std::vector<std::vector<Fr>>& points, const std::vector<Eval>& evals) {
  batch_open_with(tr, nv, polys, points, evals, [&](const Poly& g, const Fr* x) { kzg_open(tr, s, g, x, nv); });
}

// ------------------------------------------------------------------ Zeromorph over univariate KZG
// pcs/univariate/kzg.rs:23-36,175-299 and pcs/multilinear/zeromorph.rs:86-213,258-296; spec twin: oracle/pyref/zeromorph.py
struct USrs {
  const Affine* powers;  // powers_of_s_g1
  size_t size, poly_size;  // poly_size: trim size; the opening quotient is committed against powers[size - poly_size..]
};
static Affine zm_commit(const USrs& s, const Poly& p) {
  if (p.size() > s.poly_size) throw OracleError("Too large degree of poly to commit");
  return msm(p.data(), s.powers, p.size());
}
static void zm_scalars(const Fr& y, const Fr& x, const Fr& z, const Fr* u, size_t n, Fr& eval_scalar,
                       std::vector<Fr>& q_scalars) {  // zeromorph.rs:258-296
  std::vector<Fr> squares(n + 1), offsets(n), vs(n + 1);
  squares[0] = x;
  for (size_t i = 0; i < n; i++) squares[i + 1] = squares[i] * squares[i];
  Fr state = Fr::one();
  for (size_t i = n; i-- > 0;) {
    state = state * squares[i];
    offsets[i] = state;
  }
  const Fr v_numer = squares[n] - Fr::one();
  for (size_t i = 0; i <= n; i++) vs[i] = v_numer * (squares[i] - Fr::one()).inv();
  q_scalars.resize(n);
  Fr py = Fr::one();
  for (size_t i = 0; i < n; i++) {
    q_scalars[i] = Fr::zero() - (py * offsets[i] + z * (squares[i] * vs[i + 1] - u[i] * vs[i]));
    py = py * y;
  }
  eval_scalar = Fr::zero() - vs[0] * z;
}
static void zm_open(Transcript& tr, const USrs& s, const Poly& poly, const Fr* point, size_t nv) {
  const size_t n = (size_t)1 << nv;
  if (n > s.poly_size) throw OracleError("Too large degree of poly to open");
  Poly rem = poly;
  std::vector<Poly> qs(nv);
  for (size_t i = nv; i-- > 0;) {
    size_t half = (size_t)1 << i;
    Poly q(half), lo(half);
    parallelize(half, [&](size_t a, size_t b) {
      for (size_t k = a; k < b; k++) {
        q[k] = rem[half + k] - rem[k];
        lo[k] = rem[k] + q[k] * point[i];
      }
    });
    rem.swap(lo);
    qs[i] = std::move(q);
  }
  for (auto& q : qs) tr.write_comm(msm(q.data(), s.powers, q.size()));
  const Fr y = tr.squeeze();
  Poly q_hat(n, Fr::zero());
  Fr py = Fr::one();
  for (size_t k = 0; k < nv; k++) {
    const size_t off = n - ((size_t)1 << k);
    for (size_t j = 0; j < qs[k].size(); j++) q_hat[off + j] = q_hat[off + j] + py * qs[k][j];
    py = py * y;
  }
  tr.write_comm(msm(q_hat.data(), s.powers, n));
  const Fr x = tr.squeeze(), z = tr.squeeze();
  Fr eval_scalar;
  std::vector<Fr> q_scalars;
  zm_scalars(y, x, z, point, nv, eval_scalar, q_scalars);
  Poly f(n);
  parallelize(n, [&](size_t a, size_t b) {
    for (size_t j = a; j < b; j++) f[j] = z * poly[j] + q_hat[j];
  });
  for (size_t k = 0; k < nv; k++)
    for (size_t j = 0; j < qs[k].size(); j++) f[j] = f[j] + q_scalars[k] * qs[k][j];
  // quotient by X - x (the constant term, where eval_scalar * eval would go, does not enter)
  Poly quot(n - 1);
  Fr carry = Fr::zero();
  for (size_t i = n - 1; i >= 1; i--) {
    carry = f[i] + carry * x;
    quot[i - 1] = carry;
  }
  tr.write_comm(msm(quot.data(), s.powers + (s.size - s.poly_size), n - 1));
}
static void zm_batch_open(Transcript& tr, const USrs& s, size_t nv, const std::vector<const Poly*>& polys,
                          const std::vector<std::vector<Fr>>& points, const std::vector<Eval>& evals) {
  batch_open_with(tr, nv, polys, points, evals, [&](const Poly& g, const Fr* x) { zm_open(tr, s, g, x, nv); });
}

// ------------------------------------------------------------------ Lasso (spec: oracle/pyref/lasso.py)
struct LassoTable {  // same layout as lh_lasso_table
  uint32_t c, l, alpha;
  uint32_t mem_chunk[16], mem_sub[16];
  uint32_t num_terms;
  Fr g_coeff[16];
  uint8_t g_nfac[16];
  uint8_t g_fac[16][4];
};
static uint32_t subtable_entry(uint32_t kind, uint32_t m, uint32_t l) {
  uint32_t h = l / 2, x = m >> h, y = m & ((1u << h) - 1);
  return kind == 0 ? m : kind == 1 ? (x & y) : (x ^ y);
}
static Poly to_poly(const std::vector<uint32_t>& v) {
  Poly p(v.size());
  parallelize(v.size(), [&](size_t a, size_t b) {
    for (size_t k = a; k < b; k++) p[k] = Fr::from_u64(v[k]);
  });
  return p;
}
struct LassoPcsFns {
  std::function<Affine(const Poly&)> commit;
  std::function<void(size_t, const std::vector<const Poly*>&, const std::vector<std::vector<Fr>>&, const std::vector<Eval>&)>
      batch_open;
  size_t max_vars;
};
// witness of the argument: access counters in lookup order and the subtable reads (oracle/pyref/lasso.py witness)
struct LassoW {
  std::vector<std::vector<uint32_t>> rts, fcs, E;
};
static LassoW lasso_witness(const LassoTable& tb, size_t n, const uint32_t* const* dims) {
  const size_t c = tb.c, l = tb.l, alpha = tb.alpha, N = (size_t)1 << n, M = (size_t)1 << l;
  LassoW w;
  w.rts.assign(c, std::vector<uint32_t>(N));
  w.fcs.assign(c, std::vector<uint32_t>(M, 0));
  w.E.assign(alpha, std::vector<uint32_t>(N));
  for (size_t j = 0; j < c; j++)
    for (size_t k = 0; k < N; k++) w.rts[j][k] = w.fcs[j][dims[j][k]]++;
  for (size_t i = 0; i < alpha; i++)
    for (size_t k = 0; k < N; k++) w.E[i][k] = subtable_entry(tb.mem_sub[i], dims[tb.mem_chunk[i]][k], (uint32_t)l);
  return w;
}
static Poly lasso_output(const LassoTable& tb, const LassoW& w, size_t N) {
  Poly a(N);
  parallelize(N, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; k++) {
      Fr acc = Fr::zero();
      for (uint32_t m = 0; m < tb.num_terms; m++) {
        Fr v = tb.g_coeff[m];
        for (int f = 0; f < tb.g_nfac[m]; f++) v = v * Fr::from_u64(w.E[tb.g_fac[m][f]][k]);
        acc = acc + v;
      }
      a[k] = acc;
    }
  });
  return a;
}
// steps 2-7 of the argument (oracle/pyref/lasso.py argue): Surge, memory-checking grand products, evaluations.
// a, dim, rts, E: tables of >= 2^n entries; fc: >= 2^l entries
struct LassoCl {
  std::vector<Fr> r, r_z, r_N, r_M, e_rz, ev_n, ev_l;
  Fr v;
};
static LassoCl lasso_argue(Transcript& tr, const LassoTable& tb, size_t n, const Poly& a, const std::vector<const Poly*>& dimp,
                           const std::vector<const Poly*>& tsp, const std::vector<const Poly*>& ep,
                           const std::vector<const Poly*>& fcp) {
  const size_t l = tb.l, alpha = tb.alpha, N = (size_t)1 << n, M = (size_t)1 << l;
  LassoCl cl;
  cl.r = tr.squeeze_n(n);
  cl.v = evaluate(Poly(a.begin(), a.begin() + N), cl.r.data(), n);
  tr.write_fe(cl.v);
  Sop surge;
  memset(&surge, 0, sizeof surge);
  surge.global_eq = 0;
  surge.num_terms = tb.num_terms;
  for (uint32_t m = 0; m < tb.num_terms; m++) {
    surge.coeff[m] = tb.g_coeff[m];
    surge.num_factors[m] = tb.g_nfac[m];
    memcpy(surge.factor[m], tb.g_fac[m], 4);
  }
  std::vector<Poly> Ep;
  for (auto* p : ep) Ep.emplace_back(p->begin(), p->begin() + N);
  ScOut sc = sum_check_prove(tr, 0, n, surge, Ep, {cl.r}, cl.v);
  cl.r_z = sc.x;
  cl.e_rz = sc.evals;
  tr.write_fes(sc.evals);

  Fr gamma = tr.squeeze(), tau = tr.squeeze(), g2 = gamma * gamma, one = Fr::one();
  std::vector<Poly> leaves(4 * alpha);
  for (size_t i = 0; i < alpha; i++) {
    size_t j = tb.mem_chunk[i];
    Poly rs(N), ws(N), in(M), fi(M);
    const Poly &dp = *dimp[j], &epi = *ep[i], &tp = *tsp[j], &fp = *fcp[j];
    parallelize(N, [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; k++) {
        rs[k] = dp[k] * g2 + epi[k] * gamma + tp[k] - tau;
        ws[k] = rs[k] + one;
      }
    });
    parallelize(M, [&](size_t lo, size_t hi) {
      for (size_t m = lo; m < hi; m++) {
        in[m] = Fr::from_u64(m) * g2 + Fr::from_u64(subtable_entry(tb.mem_sub[i], (uint32_t)m, (uint32_t)l)) * gamma - tau;
        fi[m] = in[m] + fp[m];
      }
    });
    leaves[2 * i] = rs, leaves[2 * i + 1] = ws;
    leaves[2 * alpha + 2 * i] = in, leaves[2 * alpha + 2 * i + 1] = fi;
  }
  GpOut gp = grand_product_prove(tr, leaves);
  cl.r_N = gp.points[0], cl.r_M = gp.points[2 * alpha];
  auto head = [](const Poly& p, size_t len) { return Poly(p.begin(), p.begin() + len); };
  for (auto* p : dimp) cl.ev_n.push_back(evaluate(head(*p, N), cl.r_N.data(), n));
  for (auto* p : tsp) cl.ev_n.push_back(evaluate(head(*p, N), cl.r_N.data(), n));
  for (auto* p : ep) cl.ev_n.push_back(evaluate(head(*p, N), cl.r_N.data(), n));
  for (auto* p : fcp) cl.ev_l.push_back(evaluate(head(*p, M), cl.r_M.data(), l));
  tr.write_fes(cl.ev_n), tr.write_fes(cl.ev_l);
  return cl;
}
// commitment framing (oracle/pyref/lasso.py write_commitments): a mask of the identity commitments (identically zero
// columns) as one field element, then the other commitments in order
static void lasso_write_comms(Transcript& tr, const std::vector<Affine>& comms) {
  uint64_t mask = 0;
  for (size_t i = 0; i < comms.size(); i++)
    if (comms[i].is_identity()) mask |= (uint64_t)1 << i;
  tr.write_fe(Fr::from_u64(mask));
  for (auto& cm : comms)
    if (!cm.is_identity()) tr.write_comm(cm);
}

static void lasso_prove(Transcript& tr, const LassoPcsFns& s, const LassoTable& tb, size_t n, const uint32_t* const* dims) {
  const size_t c = tb.c, l = tb.l, alpha = tb.alpha, N = (size_t)1 << n;
  LassoW w = lasso_witness(tb, n, dims);
  Poly a = lasso_output(tb, w, N);
  std::vector<Poly> pn;  // a | dim | read_ts | E
  pn.push_back(a);
  for (size_t j = 0; j < c; j++) pn.push_back(to_poly(std::vector<uint32_t>(dims[j], dims[j] + N)));
  for (size_t j = 0; j < c; j++) pn.push_back(to_poly(w.rts[j]));
  for (size_t i = 0; i < alpha; i++) pn.push_back(to_poly(w.E[i]));
  std::vector<Poly> pl;
  for (size_t j = 0; j < c; j++) pl.push_back(to_poly(w.fcs[j]));

  tr.common_fe(Fr::from_u64(n)), tr.common_fe(Fr::from_u64(l)), tr.common_fe(Fr::from_u64(c)), tr.common_fe(Fr::from_u64(alpha));
  // every committed poly is zero-padded to nv = max(n, l) variables (one batch_open serves all)
  const size_t nv = std::max(n, l), NV = (size_t)1 << nv;
  auto padded = [&](const Poly& p) {
    Poly q = p;
    q.resize(NV, Fr::zero());
    return q;
  };
  std::vector<Poly> all;
  for (auto& p : pn) all.push_back(padded(p));
  for (auto& p : pl) all.push_back(padded(p));
  {
    std::vector<Affine> comms;
    for (auto& p : all) comms.push_back(s.commit(p));
    lasso_write_comms(tr, comms);
  }
  std::vector<const Poly*> dimp, tsp, ep, fcp;
  for (size_t j = 0; j < c; j++) dimp.push_back(&pn[1 + j]), tsp.push_back(&pn[1 + c + j]), fcp.push_back(&pl[j]);
  for (size_t i = 0; i < alpha; i++) ep.push_back(&pn[1 + 2 * c + i]);
  LassoCl cl = lasso_argue(tr, tb, n, pn[0], dimp, tsp, ep, fcp);

  std::vector<Eval> evs;
  evs.push_back(Eval{0, 0, cl.v});
  for (size_t i = 0; i < alpha; i++) evs.push_back(Eval{(uint32_t)(1 + 2 * c + i), 1, cl.e_rz[i]});
  for (size_t j = 0; j < c; j++) evs.push_back(Eval{(uint32_t)(1 + j), 2, cl.ev_n[j]});
  for (size_t j = 0; j < c; j++) evs.push_back(Eval{(uint32_t)(1 + c + j), 2, cl.ev_n[c + j]});
  for (size_t i = 0; i < alpha; i++) evs.push_back(Eval{(uint32_t)(1 + 2 * c + i), 2, cl.ev_n[2 * c + i]});
  for (size_t j = 0; j < c; j++) evs.push_back(Eval{(uint32_t)(1 + 2 * c + alpha + j), 3, cl.ev_l[j]});
  std::vector<const Poly*> pp;
  for (auto& p : all) pp.push_back(&p);
  auto pad_pt = [&](std::vector<Fr> pt) {
    pt.resize(nv, Fr::zero());
    return pt;
  };
  s.batch_open(nv, pp, {pad_pt(cl.r), pad_pt(cl.r_z), pad_pt(cl.r_N), pad_pt(cl.r_M)}, evs);
}

// ------------------------------------------------------------------ HyperPlonk with LogUp (backend/hyperplonk)
// Restated: BooleanHypercube util/arithmetic/bh.rs:5-153; Expression::evaluate util/expression.rs:107-169;
// instance polys hyperplonk.rs:365-369 + prover.rs:32-48; lookup_compressed/m/h polys prover.rs:50-260;
// permutation_z_polys prover.rs:262-345; EvaluationsProver over a general expression
// piop/sum_check/classic.rs:41-149 + classic/eval.rs:102-131 (rotated / identity / Lagrange leaves as tables,
// evaluate THEN bind); evaluations in pcs_query order prover.rs:388-406, verifier.rs:147-180;
// rotation points poly/multilinear.rs:477-526; HyperPlonk::prove hyperplonk.rs:164-291.
static const uint32_t BH_PRIM[32] = {
    1, 3, 7, 11, 19, 37, 67, 131, 285, 529, 1033, 2053, 4179, 8219, 16427, 32771, 65581, 131081, 262183, 524327,
    1048585, 2097157, 4194307, 8388641, 16777243, 33554441, 67108935, 134217767, 268435465, 536870917, 1073741907,
    2147483657u};
static const uint32_t BH_XINV[32] = {
    0, 1, 3, 5, 9, 18, 33, 65, 142, 264, 516, 1026, 2089, 4109, 8213, 16385, 32790, 65540, 131091, 262163, 524292,
    1048578, 2097153, 4194320, 8388621, 16777220, 33554467, 67108883, 134217732, 268435458, 536870953, 1073741828};
struct Hypercube {
  size_t nv;
  size_t next(size_t b) const {
    b <<= 1;
    return b ^ ((b >> nv) * BH_PRIM[nv]);
  }
  size_t prev(size_t b) const { return (b >> 1) ^ ((b & 1) * BH_XINV[nv]); }
  size_t rotate(size_t b, int rot) const {
    for (int i = 0; i > rot; i--) b = prev(b);
    for (int i = 0; i < rot; i++) b = next(b);
    return b;
  }
  std::vector<uint32_t> order() const {  // iter(): 0, 1, x, x^2, ...
    std::vector<uint32_t> o((size_t)1 << nv);
    o[0] = 0;
    size_t b = 1;
    for (size_t k = 1; k < o.size(); k++) {
      o[k] = (uint32_t)b;
      b = next(b);
    }
    return o;
  }
};

struct ExprNode {  // same layout as lh_expr_node
  uint32_t op;
  int32_t a, b;
  uint32_t reserved;
  Fr scalar;
};
struct Expr {
  const ExprNode* nodes;
  size_t n;
};
enum { EX_CONSTANT, EX_IDENTITY, EX_LAGRANGE, EX_EQ_XY, EX_POLYNOMIAL, EX_CHALLENGE, EX_NEGATED, EX_SUM, EX_PRODUCT, EX_SCALED };

// Expression::evaluate with the leaves supplied by the caller (constants/challenges are leaves too)
template <class Leaf>
static Fr eval_expr(const Expr& e, std::vector<Fr>& scratch, Leaf&& leaf) {
  scratch.resize(e.n);
  for (size_t i = 0; i < e.n; i++) {
    const ExprNode& nd = e.nodes[i];
    switch (nd.op) {
      case EX_NEGATED: scratch[i] = Fr::zero() - scratch[nd.a]; break;
      case EX_SUM: scratch[i] = scratch[nd.a] + scratch[nd.b]; break;
      case EX_PRODUCT: scratch[i] = scratch[nd.a] * scratch[nd.b]; break;
      case EX_SCALED: scratch[i] = scratch[nd.a] * nd.scalar; break;
      default: scratch[i] = leaf(nd);
    }
  }
  return scratch[e.n - 1];
}
static size_t expr_degree(const Expr& e) {
  std::vector<size_t> d(e.n);
  for (size_t i = 0; i < e.n; i++) {
    const ExprNode& nd = e.nodes[i];
    switch (nd.op) {
      case EX_CONSTANT:
      case EX_CHALLENGE: d[i] = 0; break;
      case EX_NEGATED:
      case EX_SCALED: d[i] = d[nd.a]; break;
      case EX_SUM: d[i] = std::max(d[nd.a], d[nd.b]); break;
      case EX_PRODUCT: d[i] = d[nd.a] + d[nd.b]; break;
      default: d[i] = 1;
    }
  }
  return d[e.n - 1];
}

static void batch_invert(Poly& v) {  // zero stays zero
  std::vector<Fr> pre(v.size());
  Fr acc = Fr::one();
  for (size_t i = 0; i < v.size(); i++) {
    pre[i] = acc;
    if (!v[i].is_zero()) acc = acc * v[i];
  }
  Fr inv = acc.inv();
  for (size_t i = v.size(); i-- > 0;) {
    if (v[i].is_zero()) continue;
    Fr x = inv * pre[i];
    inv = inv * v[i];
    v[i] = x;
  }
}
static void par_batch_invert(Poly& v) {
  parallelize(v.size(), [&](size_t a, size_t b) {
    Poly part(v.begin() + a, v.begin() + b);
    batch_invert(part);
    std::copy(part.begin(), part.end(), v.begin() + a);
  });
}

struct HpLookup {  // same layout as lh_hp_lookup
  const Expr* inputs;
  const Expr* tables;
  size_t width;
};
struct HpParam {  // same layout as lh_hp_param, host pointers
  size_t num_vars;
  size_t num_instance_polys;
  const size_t* num_instances;
  size_t num_preprocess_polys;
  const Fr* const* preprocess_polys;
  size_t num_witness_polys;
  size_t num_challenges;
  size_t num_lookups;
  const HpLookup* lookups;
  size_t num_permutation_polys;
  const size_t* permutation_poly_index;
  const Fr* const* permutation_polys;
  size_t num_permutation_z_polys;
  Expr expression;
  size_t num_lasso_lookups;  // lookups proven by Lasso (oracle/pyref/hyperplonk.py LassoLookup)
  const struct HpLassoLookup* lasso_lookups;
};
struct HpLassoLookup {  // same layout as lh_hp_lasso_lookup
  LassoTable table;
  size_t output_poly;
  size_t chunk_polys[8];
};

struct FrHash {
  size_t operator()(const Fr& f) const { return (size_t)(f.v[0] ^ (f.v[1] * 0x9e3779b97f4a7c15ull) ^ f.v[2] ^ f.v[3]); }
};

// poly/multilinear.rs:477-549
static std::vector<size_t> point_pattern(bool nxt, size_t nv, size_t distance) {
  size_t rem = nxt ? BH_PRIM[nv] : BH_XINV[nv];
  std::vector<size_t> pat((size_t)1 << distance, 0);
  for (size_t depth = 0; depth < distance; depth++) {
    size_t step = (size_t)1 << (distance - depth);
    for (size_t e = 0; e < pat.size(); e += step) {
      size_t rot = nxt ? pat[e] << 1 : pat[e] >> 1;
      pat[e + step / 2] = rot ^ rem;
      pat[e] = rot;
    }
  }
  return pat;
}
static std::vector<std::vector<Fr>> rotation_points(const std::vector<Fr>& x, int rot) {
  if (rot == 0) return {x};
  size_t n = x.size(), dist = (size_t)abs(rot), nx = n - dist;
  std::vector<std::vector<Fr>> out;
  for (size_t p : point_pattern(rot > 0, n, dist)) {
    std::vector<Fr> pt;
    if (rot < 0) {
      for (size_t i = 0; i < nx; i++) pt.push_back(((p >> i) & 1) ? Fr::one() - x[dist + i] : x[dist + i]);
      for (size_t i = 0; i < dist; i++) pt.push_back(((p >> (i + nx)) & 1) ? Fr::one() : Fr::zero());
    } else {
      for (size_t i = 0; i < dist; i++) pt.push_back(((p >> i) & 1) ? Fr::one() : Fr::zero());
      for (size_t i = 0; i < nx; i++) pt.push_back(((p >> (i + dist)) & 1) ? Fr::one() - x[i] : x[i]);
    }
    out.push_back(pt);
  }
  return out;
}

static void hyperplonk_prove(Transcript& tr, const Srs& srs, const HpParam& pp, const Fr* const* instances,
                             const Fr* const* witness) {
  const size_t nv = pp.num_vars, n = (size_t)1 << nv;
  if (nv == 0 || nv >= 32) throw OracleError("hyperplonk: bad num_vars");
  Hypercube bh{nv};
  const std::vector<uint32_t> order = bh.order();
  std::vector<uint32_t> nth(n);
  for (size_t k = 0; k < n; k++) nth[order[k]] = (uint32_t)k;

  std::vector<Poly> polys;
  for (size_t i = 0; i < pp.num_instance_polys; i++) {
    Poly p(n, Fr::zero());
    for (size_t k = 0; k < pp.num_instances[i]; k++) {
      tr.common_fe(instances[i][k]);
      p[k + 1 < n ? order[k + 1] : 0] = instances[i][k];
    }
    polys.push_back(std::move(p));
  }
  for (size_t i = 0; i < pp.num_preprocess_polys; i++) polys.emplace_back(pp.preprocess_polys[i], pp.preprocess_polys[i] + n);
  for (size_t i = 0; i < pp.num_witness_polys; i++) {
    polys.emplace_back(witness[i], witness[i] + n);
    tr.write_comm(commit(srs, polys.back()));
  }
  std::vector<Fr> challenges = tr.squeeze_n(pp.num_challenges);

  // lookups
  Fr beta = tr.squeeze();
  auto row_leaf = [&](size_t b) {
    return [&, b](const ExprNode& nd) -> Fr {
      switch (nd.op) {
        case EX_CONSTANT: return nd.scalar;
        case EX_IDENTITY: return Fr::from_u64(b);
        case EX_LAGRANGE: {
          long long m = (long long)nd.a % (long long)n;
          if (m < 0) m += (long long)n;
          return order[(size_t)m] == b ? Fr::one() : Fr::zero();
        }
        case EX_POLYNOMIAL: return polys[nd.a][bh.rotate(b, nd.b)];
        case EX_CHALLENGE: return challenges[nd.a];
        default: throw OracleError("lookup expression: eq_xy is not allowed");
      }
    };
  };
  std::vector<Poly> comp_in, comp_tab, m_polys, h_polys;
  for (size_t k = 0; k < pp.num_lookups; k++) {
    const HpLookup& lk = pp.lookups[k];
    Poly ci(n, Fr::zero()), ct(n, Fr::zero());
    parallelize(n, [&](size_t lo, size_t hi) {
      std::vector<Fr> scratch;
      for (size_t b = lo; b < hi; b++) {
        Fr pw = Fr::one();
        auto leaf = row_leaf(b);
        for (size_t w = 0; w < lk.width; w++) {
          ci[b] = ci[b] + pw * eval_expr(lk.inputs[w], scratch, leaf);
          ct[b] = ct[b] + pw * eval_expr(lk.tables[w], scratch, leaf);
          pw = pw * beta;
        }
      }
    });
    std::unordered_map<Fr, uint32_t, FrHash> index;
    index.reserve(2 * n);
    for (size_t b = 0; b < n; b++) index[ct[b]] = (uint32_t)b;  // the last row holding a value wins
    std::vector<uint64_t> cnt(n, 0);
    for (size_t b = 0; b < n; b++) {
      auto it = index.find(ci[b]);
      if (it == index.end()) throw OracleError("Invalid lookup input");
      cnt[it->second]++;
    }
    Poly m(n);
    for (size_t b = 0; b < n; b++) m[b] = Fr::from_u64(cnt[b]);
    comp_in.push_back(std::move(ci));
    comp_tab.push_back(std::move(ct));
    m_polys.push_back(std::move(m));
  }
  for (auto& m : m_polys) tr.write_comm(commit(srs, m));
  // Lasso lookups: witnesses from the circuit's chunk columns, commitments with the identity-mask framing
  struct LassoSt {
    const HpLassoLookup* lk;
    std::vector<std::vector<uint32_t>> dims;
    LassoW w;
    std::vector<Poly> rts, E, fcs;  // 2^nv entries each (final_cts zero padded)
  };
  std::vector<LassoSt> lasso(pp.num_lasso_lookups);
  if (pp.num_lasso_lookups) {
    std::vector<Affine> comms;
    for (size_t k = 0; k < pp.num_lasso_lookups; k++) {
      LassoSt& st = lasso[k];
      st.lk = &pp.lasso_lookups[k];
      const LassoTable& tb = st.lk->table;
      if (tb.l > nv) throw OracleError("Lasso subtable larger than the circuit");
      st.dims.assign(tb.c, std::vector<uint32_t>(n));
      std::vector<const uint32_t*> dptr;
      for (size_t j = 0; j < tb.c; j++) {
        const Poly& col = polys.at(st.lk->chunk_polys[j]);
        for (size_t b = 0; b < n; b++) {
          uint64_t canon[4];
          col[b].to_raw(canon);
          if (canon[1] || canon[2] || canon[3] || (canon[0] >> tb.l)) throw OracleError("Invalid lookup input");
          st.dims[j][b] = (uint32_t)canon[0];
        }
        dptr.push_back(st.dims[j].data());
      }
      st.w = lasso_witness(tb, nv, dptr.data());
      Poly a = lasso_output(tb, st.w, n);
      const Poly& out = polys.at(st.lk->output_poly);
      for (size_t b = 0; b < n; b++)
        if (!(a[b] == out[b])) throw OracleError("Invalid lookup input");
      for (size_t j = 0; j < tb.c; j++) st.rts.push_back(to_poly(st.w.rts[j]));
      for (size_t i = 0; i < tb.alpha; i++) st.E.push_back(to_poly(st.w.E[i]));
      for (size_t j = 0; j < tb.c; j++) {
        Poly f = to_poly(st.w.fcs[j]);
        f.resize(n, Fr::zero());
        st.fcs.push_back(std::move(f));
      }
      for (auto& p : st.rts) comms.push_back(commit(srs, p));
      for (auto& p : st.E) comms.push_back(commit(srs, p));
      for (auto& p : st.fcs) comms.push_back(commit(srs, p));
    }
    lasso_write_comms(tr, comms);
  }

  Fr gamma = tr.squeeze();
  for (size_t k = 0; k < pp.num_lookups; k++) {
    Poly hi(n), ht(n), h(n);
    for (size_t b = 0; b < n; b++) hi[b] = gamma + comp_in[k][b], ht[b] = gamma + comp_tab[k][b];
    par_batch_invert(hi);
    par_batch_invert(ht);
    for (size_t b = 0; b < n; b++) h[b] = hi[b] - ht[b] * m_polys[k][b];
    h_polys.push_back(std::move(h));
  }
  // permutation z polys
  std::vector<Poly> z_polys;
  const size_t nchunks = pp.num_permutation_z_polys, nperm = pp.num_permutation_polys;
  if (nperm && nchunks) {
    const size_t chunk = (nperm + nchunks - 1) / nchunks;
    std::vector<Poly> products;
    for (size_t c = 0; c < nchunks; c++) {
      Poly prod(n, Fr::one());
      const size_t k0 = c * chunk, k1 = std::min(nperm, k0 + chunk);
      for (size_t k = k0; k < k1; k++) {
        const Poly& val = polys[pp.permutation_poly_index[k]];
        const Fr* perm = pp.permutation_polys[k];
        parallelize(n, [&](size_t lo, size_t hi) {
          for (size_t b = lo; b < hi; b++) prod[b] = prod[b] * (beta * perm[b] + gamma + val[b]);
        });
      }
      par_batch_invert(prod);
      for (size_t k = k0; k < k1; k++) {
        const Poly& val = polys[pp.permutation_poly_index[k]];
        const uint64_t off = (uint64_t)k << nv;
        parallelize(n, [&](size_t lo, size_t hi) {
          for (size_t b = lo; b < hi; b++) prod[b] = prod[b] * (Fr::from_u64(off + b) * beta + gamma + val[b]);
        });
      }
      products.push_back(std::move(prod));
    }
    std::vector<Fr> z(nchunks * n, Fr::zero());
    Fr state = Fr::one();
    size_t pos = nchunks;
    z[pos++] = state;
    for (size_t k = 1; k < n && pos < z.size(); k++)
      for (size_t c = 0; c < nchunks && pos < z.size(); c++) {
        state = state * products[c][order[k]];
        z[pos++] = state;
      }
    for (size_t c = 0; c < nchunks; c++) {
      Poly zp(n);
      for (size_t b = 0; b < n; b++) zp[b] = z[c + nchunks * nth[b]];
      z_polys.push_back(std::move(zp));
    }
  }
  for (auto& h : h_polys) tr.write_comm(commit(srs, h));
  for (auto& z : z_polys) tr.write_comm(commit(srs, z));

  Fr alpha = tr.squeeze();
  std::vector<Fr> y = tr.squeeze_n(nv);
  for (size_t k = 0; k < nperm; k++) polys.emplace_back(pp.permutation_polys[k], pp.permutation_polys[k] + n);
  for (auto& m : m_polys) polys.push_back(m);
  for (auto& h : h_polys) polys.push_back(h);
  for (auto& z : z_polys) polys.push_back(z);
  challenges.push_back(beta);
  challenges.push_back(gamma);
  challenges.push_back(alpha);

  // zero-check: ClassicSumCheck<EvaluationsProver>, claim 0
  const Expr& E = pp.expression;
  const size_t degree = expr_degree(E);
  if (degree < 2) throw OracleError("EvaluationsProver: degree < 2");
  // leaf tables: every poly (bound each round), plus one table per distinct non-trivial leaf
  struct Leaf {
    uint32_t op;
    int32_t a, b;
  };
  std::vector<Leaf> leaves;
  std::vector<Poly> extra;  // tables of the leaves that are not a poly at rotation 0
  std::vector<int> leaf_of(E.n, -1);  // node -> index into `polys` (>= 0) or -(1 + index into `extra`)
  for (size_t i = 0; i < E.n; i++) {
    const ExprNode& nd = E.nodes[i];
    if (nd.op < EX_IDENTITY || nd.op > EX_POLYNOMIAL) continue;
    if (nd.op == EX_POLYNOMIAL && nd.b == 0) {
      leaf_of[i] = nd.a;
      continue;
    }
    int found = -1;
    for (size_t k = 0; k < leaves.size(); k++)
      if (leaves[k].op == nd.op && (nd.op == EX_IDENTITY || (leaves[k].a == nd.a && (nd.op != EX_POLYNOMIAL || leaves[k].b == nd.b))))
        found = (int)k;
    if (found < 0) {
      Poly t(n, Fr::zero());
      if (nd.op == EX_IDENTITY) {
        for (size_t b = 0; b < n; b++) t[b] = Fr::from_u64(b);
      } else if (nd.op == EX_LAGRANGE) {
        long long m = (long long)nd.a % (long long)n;
        if (m < 0) m += (long long)n;
        t[order[(size_t)m]] = Fr::one();
      } else if (nd.op == EX_EQ_XY) {
        if (nd.a != 0) throw OracleError("zero-check has a single eq_xy");
        t = eq_xy(y.data(), nv);
      } else {
        for (size_t b = 0; b < n; b++) t[b] = polys[nd.a][bh.rotate(b, nd.b)];
      }
      leaves.push_back(Leaf{nd.op, nd.a, nd.b});
      extra.push_back(std::move(t));
      found = (int)leaves.size() - 1;
    }
    leaf_of[i] = -(1 + found);
  }
  Fr claim = Fr::zero();
  std::vector<Fr> xs;
  for (size_t round = 0; round < nv; round++) {
    const size_t size = (size_t)1 << (nv - round - 1);
    const size_t nt = (size_t)num_threads();
    const size_t chunk = std::max<size_t>(1, (size + nt - 1) / nt), nchk = (size + chunk - 1) / chunk;
    std::vector<std::vector<Fr>> partial(nchk, std::vector<Fr>(degree + 1, Fr::zero()));
    parallelize(nchk, [&](size_t c0, size_t c1) {
      std::vector<Fr> scratch;
      for (size_t c = c0; c < c1; c++)
        for (size_t b = c * chunk; b < std::min(size, (c + 1) * chunk); b++)
          for (size_t X = 1; X <= degree; X++) {
            const Fr fx = Fr::from_u64(X);
            auto leaf = [&](const ExprNode& nd) -> Fr {
              if (nd.op == EX_CONSTANT) return nd.scalar;
              if (nd.op == EX_CHALLENGE) return challenges[nd.a];
              const int id = leaf_of[&nd - E.nodes];
              const Poly& t = id >= 0 ? polys[id] : extra[-id - 1];
              return t[2 * b] + (t[2 * b + 1] - t[2 * b]) * fx;
            };
            partial[c][X] = partial[c][X] + eval_expr(E, scratch, leaf);
          }
    });
    std::vector<Fr> ev(degree + 1, Fr::zero());
    for (auto& p : partial)
      for (size_t X = 1; X <= degree; X++) ev[X] = ev[X] + p[X];
    ev[0] = claim - ev[1];
    tr.write_fes(ev);
    Fr r = tr.squeeze();
    claim = interpolate(ev, r);
    xs.push_back(r);
    for (auto& p : polys) p = fix_var(p, r);
    for (auto& p : extra) p = fix_var(p, r);
  }
  // evaluations, pcs_query order: (poly, rotation) sorted, polys past the instances
  std::vector<std::pair<size_t, int>> query;
  for (size_t i = 0; i < E.n; i++)
    if (E.nodes[i].op == EX_POLYNOMIAL && (size_t)E.nodes[i].a >= pp.num_instance_polys)
      query.push_back({(size_t)E.nodes[i].a, E.nodes[i].b});
  std::sort(query.begin(), query.end());
  query.erase(std::unique(query.begin(), query.end()), query.end());
  std::vector<int> rots;
  for (auto& q : query) rots.push_back(q.second);
  std::sort(rots.begin(), rots.end());
  rots.erase(std::unique(rots.begin(), rots.end()), rots.end());
  std::vector<std::vector<Fr>> points;
  std::vector<size_t> rot_off;
  for (int r : rots) {
    rot_off.push_back(points.size());
    for (auto& pt : rotation_points(xs, r)) points.push_back(pt);
  }
  // the full tables are gone after binding: rebuild what the openings need from the inputs
  std::vector<Poly> full;
  for (size_t i = 0; i < pp.num_instance_polys; i++) {
    Poly p(n, Fr::zero());
    for (size_t k = 0; k < pp.num_instances[i]; k++) p[k + 1 < n ? order[k + 1] : 0] = instances[i][k];
    full.push_back(std::move(p));
  }
  for (size_t i = 0; i < pp.num_preprocess_polys; i++) full.emplace_back(pp.preprocess_polys[i], pp.preprocess_polys[i] + n);
  for (size_t i = 0; i < pp.num_witness_polys; i++) full.emplace_back(witness[i], witness[i] + n);
  for (size_t k = 0; k < nperm; k++) full.emplace_back(pp.permutation_polys[k], pp.permutation_polys[k] + n);
  for (auto& m : m_polys) full.push_back(m);
  for (auto& h : h_polys) full.push_back(h);
  for (auto& z : z_polys) full.push_back(z);
  std::vector<Eval> evals;
  std::vector<Fr> written;
  for (auto& q : query) {
    const size_t ri = std::lower_bound(rots.begin(), rots.end(), q.second) - rots.begin();
    if (q.second == 0) {
      evals.push_back(Eval{(uint32_t)q.first, (uint32_t)rot_off[ri], polys[q.first][0]});
    } else {
      size_t k = 0;
      for (auto& pt : rotation_points(xs, q.second))
        evals.push_back(Eval{(uint32_t)q.first, (uint32_t)(rot_off[ri] + k++), evaluate(full[q.first], pt.data(), nv)});
    }
  }
  for (auto& e : evals) written.push_back(e.value);
  tr.write_fes(written);
  // Lasso lookups: the argument, then its claims join the one batch opening
  for (LassoSt& st : lasso) {
    const LassoTable& tb = st.lk->table;
    const size_t c = tb.c, l = tb.l, alpha = tb.alpha;
    tr.common_fe(Fr::from_u64(nv)), tr.common_fe(Fr::from_u64(l)), tr.common_fe(Fr::from_u64(c)), tr.common_fe(Fr::from_u64(alpha));
    std::vector<const Poly*> dimp, tsp, ep, fcp;
    for (size_t j = 0; j < c; j++) dimp.push_back(&full.at(st.lk->chunk_polys[j])), tsp.push_back(&st.rts[j]), fcp.push_back(&st.fcs[j]);
    for (size_t i = 0; i < alpha; i++) ep.push_back(&st.E[i]);
    LassoCl cl = lasso_argue(tr, tb, nv, full.at(st.lk->output_poly), dimp, tsp, ep, fcp);
    const uint32_t base = (uint32_t)full.size(), p0 = (uint32_t)points.size();
    for (auto& p : st.rts) full.push_back(p);
    for (auto& p : st.E) full.push_back(p);
    for (auto& p : st.fcs) full.push_back(p);
    for (const std::vector<Fr>* pt : {&cl.r, &cl.r_z, &cl.r_N, &cl.r_M}) {
      std::vector<Fr> q = *pt;
      q.resize(nv, Fr::zero());
      points.push_back(q);
    }
    evals.push_back(Eval{(uint32_t)st.lk->output_poly, p0, cl.v});
    for (size_t i = 0; i < alpha; i++) evals.push_back(Eval{(uint32_t)(base + c + i), p0 + 1, cl.e_rz[i]});
    for (size_t j = 0; j < c; j++) evals.push_back(Eval{(uint32_t)st.lk->chunk_polys[j], p0 + 2, cl.ev_n[j]});
    for (size_t j = 0; j < c; j++) evals.push_back(Eval{(uint32_t)(base + j), p0 + 2, cl.ev_n[c + j]});
    for (size_t i = 0; i < alpha; i++) evals.push_back(Eval{(uint32_t)(base + c + i), p0 + 2, cl.ev_n[2 * c + i]});
    for (size_t j = 0; j < c; j++) evals.push_back(Eval{(uint32_t)(base + c + alpha + j), p0 + 3, cl.ev_l[j]});
  }
  std::vector<const Poly*> ptrs;
  for (auto& p : full) ptrs.push_back(&p);
  batch_open(tr, srs, nv, ptrs, points, evals);
}

// ------------------------------------------------------------------ C interface (ctypes)
static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH                   \
  }                                 \
  catch (const std::exception& e) { \
    g_err = e.what();               \
    return -1;                      \
  }                                 \
  return 0;

static Poly load_poly(const Fr* p, size_t n) { return Poly(p, p + n); }

extern "C" {
const char* orc_last_error() { return g_err.c_str(); }
int orc_num_threads() { return num_threads(); }
void orc_set_threads(int n) { g_threads = n; }

void* orc_tr_new() { return new Transcript(); }
void orc_tr_free(void* t) { delete (Transcript*)t; }
size_t orc_tr_proof(void* t, const uint8_t** p) {
  Transcript* tr = (Transcript*)t;
  *p = tr->stream.data();
  return tr->stream.size();
}
void orc_tr_write_fe(void* t, const Fr* f) { ((Transcript*)t)->write_fe(*f); }
void orc_tr_common_fe(void* t, const Fr* f) { ((Transcript*)t)->common_fe(*f); }
void orc_tr_squeeze(void* t, Fr* out) { *out = ((Transcript*)t)->squeeze(); }
int orc_tr_write_comm(void* t, const Affine* p) {
  ORC_TRY((Transcript*)t)->write_comm(*p);
  ORC_CATCH
}

// eqs[k][b] = eq_k(b; s) * G, flat (kzg.rs:166-228); fixed-base by 8-bit windows of the generator
int orc_setup(const Fr* ss, size_t nv, Affine* eqs_flat) {
  ORC_TRY
  size_t total = ((size_t)2 << nv) - 1;
  std::vector<Fr> scal(total);
  for (size_t k = 0; k <= nv; k++) {
    Poly e = eq_xy(ss, k);
    std::copy(e.begin(), e.end(), scal.begin() + (((size_t)1 << k) - 1));
  }
  std::vector<Affine> tab(32 * 255);
  Jac off = jac_from_affine(Affine{Fq::from_u64(1), Fq::from_u64(2)});
  for (int w = 0; w < 32; w++) {
    Jac acc = off;
    for (int d = 0; d < 255; d++) {
      tab[w * 255 + d] = jac_to_affine(acc);
      acc = jac_add(acc, off);
    }
    off = acc;
  }
  parallelize(total, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      uint64_t c[4];
      scal[i].to_raw(c);
      Jac acc = Jac::identity();
      for (int w = 0; w < 32; w++) {
        uint32_t d = (c[w / 8] >> (8 * (w % 8))) & 0xff;
        if (d) acc = jac_add_affine(acc, tab[w * 255 + d - 1]);
      }
      eqs_flat[i] = jac_to_affine(acc);
    }
  });
  ORC_CATCH
}

int orc_msm(const Fr* scalars, const Affine* bases, size_t n, Affine* out) {
  ORC_TRY* out = msm(scalars, bases, n);
  ORC_CATCH
}
int orc_commit(const Affine* eqs, size_t srs_nv, const Fr* poly, size_t nv, Affine* out) {
  ORC_TRY* out = commit(Srs{eqs, srs_nv}, load_poly(poly, (size_t)1 << nv));
  ORC_CATCH
}
int orc_open(void* t, const Affine* eqs, size_t srs_nv, const Fr* poly, size_t nv, const Fr* point, Fr* out_eval) {
  ORC_TRY* out_eval = kzg_open(*(Transcript*)t, Srs{eqs, srs_nv}, load_poly(poly, (size_t)1 << nv), point, nv);
  ORC_CATCH
}
int orc_batch_open(void* t, const Affine* eqs, size_t srs_nv, size_t nv, const Fr* const* polys, size_t num_polys,
                   const Fr* points, size_t num_points, const Eval* evals, size_t num_evals) {
  ORC_TRY
  std::vector<Poly> ps;
  for (size_t i = 0; i < num_polys; i++) ps.push_back(load_poly(polys[i], (size_t)1 << nv));
  std::vector<const Poly*> pp;
  for (auto& p : ps) pp.push_back(&p);
  std::vector<std::vector<Fr>> pts;
  for (size_t j = 0; j < num_points; j++) pts.emplace_back(points + j * nv, points + (j + 1) * nv);
  batch_open(*(Transcript*)t, Srs{eqs, srs_nv}, nv, pp, pts, std::vector<Eval>(evals, evals + num_evals));
  ORC_CATCH
}
int orc_sumcheck_prove(void* t, int kind, size_t nv, const Sop* e, const Fr* const* polys, size_t num_polys,
                       const Fr* ys, size_t num_ys, const Fr* sum, Fr* out_x, Fr* out_evals) {
  ORC_TRY
  std::vector<Poly> ps;
  for (size_t i = 0; i < num_polys; i++) ps.push_back(load_poly(polys[i], (size_t)1 << nv));
  std::vector<std::vector<Fr>> yv;
  for (size_t j = 0; j < num_ys; j++) yv.emplace_back(ys + j * nv, ys + (j + 1) * nv);
  ScOut o = sum_check_prove(*(Transcript*)t, kind, nv, *e, ps, yv, *sum);
  std::copy(o.x.begin(), o.x.end(), out_x);
  std::copy(o.evals.begin(), o.evals.end(), out_evals);
  ORC_CATCH
}
int orc_frac_gkr_prove(void* t, size_t B, size_t nv, const Fr* const* ps, const Fr* const* qs, Fr* p_xs, Fr* q_xs,
                       Fr* x) {
  ORC_TRY
  std::vector<Poly> p, q;
  for (size_t b = 0; b < B; b++) p.push_back(load_poly(ps[b], (size_t)1 << nv)), q.push_back(load_poly(qs[b], (size_t)1 << nv));
  FracOut o = frac_gkr_prove(*(Transcript*)t, p, q);
  std::copy(o.p_xs.begin(), o.p_xs.end(), p_xs);
  std::copy(o.q_xs.begin(), o.q_xs.end(), q_xs);
  std::copy(o.x.begin(), o.x.end(), x);
  ORC_CATCH
}
int orc_grand_product_prove(void* t, size_t B, const Fr* const* leaves, const size_t* nvs, Fr* roots, Fr* claims,
                            Fr* points) {
  ORC_TRY
  std::vector<Poly> lv;
  for (size_t b = 0; b < B; b++) lv.push_back(load_poly(leaves[b], (size_t)1 << nvs[b]));
  GpOut o = grand_product_prove(*(Transcript*)t, lv);
  std::copy(o.roots.begin(), o.roots.end(), roots);
  std::copy(o.claims.begin(), o.claims.end(), claims);
  for (size_t b = 0; b < B; b++) points = std::copy(o.points[b].begin(), o.points[b].end(), points);
  ORC_CATCH
}
int orc_hyperplonk_prove(void* t, const Affine* eqs, size_t srs_nv, const HpParam* pp, const Fr* const* instances,
                         const Fr* const* witness) {
  ORC_TRY hyperplonk_prove(*(Transcript*)t, Srs{eqs, srs_nv}, *pp, instances, witness);
  ORC_CATCH
}
int orc_lasso_prove(void* t, const Affine* eqs, size_t srs_nv, const LassoTable* tb, size_t n,
                    const uint32_t* const* dims) {
  ORC_TRY
  Transcript& tr = *(Transcript*)t;
  const Srs s{eqs, srs_nv};
  LassoPcsFns f;
  f.commit = [&](const Poly& p) { return commit(s, p); };
  f.batch_open = [&](size_t nv, const std::vector<const Poly*>& polys, const std::vector<std::vector<Fr>>& pts,
                     const std::vector<Eval>& evs) { batch_open(tr, s, nv, polys, pts, evs); };
  f.max_vars = srs_nv;
  lasso_prove(tr, f, *tb, n, dims);
  ORC_CATCH
}
// powers_of_s_g1[i] = s^i G (univariate/kzg.rs:175-218)
int orc_usetup(const Fr* s, size_t poly_size, Affine* powers) {
  ORC_TRY
  std::vector<Fr> scal(poly_size);
  Fr p = Fr::one();
  for (size_t i = 0; i < poly_size; i++) {
    scal[i] = p;
    p = p * *s;
  }
  std::vector<Affine> tab(32 * 255);
  Jac off = jac_from_affine(Affine{Fq::from_u64(1), Fq::from_u64(2)});
  for (int w = 0; w < 32; w++) {
    Jac acc = off;
    for (int d = 0; d < 255; d++) {
      tab[w * 255 + d] = jac_to_affine(acc);
      acc = jac_add(acc, off);
    }
    off = acc;
  }
  parallelize(poly_size, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      uint64_t c[4];
      scal[i].to_raw(c);
      Jac acc = Jac::identity();
      for (int w = 0; w < 32; w++) {
        uint32_t d = (c[w / 8] >> (8 * (w % 8))) & 0xff;
        if (d) acc = jac_add_affine(acc, tab[w * 255 + d - 1]);
      }
      powers[i] = jac_to_affine(acc);
    }
  });
  ORC_CATCH
}
int orc_zm_commit(const Affine* powers, size_t size, size_t poly_size, const Fr* poly, size_t nv, Affine* out) {
  ORC_TRY* out = zm_commit(USrs{powers, size, poly_size}, load_poly(poly, (size_t)1 << nv));
  ORC_CATCH
}
int orc_zm_open(void* t, const Affine* powers, size_t size, size_t poly_size, const Fr* poly, size_t nv, const Fr* point) {
  ORC_TRY zm_open(*(Transcript*)t, USrs{powers, size, poly_size}, load_poly(poly, (size_t)1 << nv), point, nv);
  ORC_CATCH
}
int orc_zm_batch_open(void* t, const Affine* powers, size_t size, size_t poly_size, size_t nv, const Fr* const* polys,
                      size_t num_polys, const Fr* points, size_t num_points, const Eval* evals, size_t num_evals) {
  ORC_TRY
  std::vector<Poly> ps;
  for (size_t i = 0; i < num_polys; i++) ps.push_back(load_poly(polys[i], (size_t)1 << nv));
  std::vector<const Poly*> pp;
  for (auto& p : ps) pp.push_back(&p);
  std::vector<std::vector<Fr>> pts;
  for (size_t j = 0; j < num_points; j++) pts.emplace_back(points + j * nv, points + (j + 1) * nv);
  zm_batch_open(*(Transcript*)t, USrs{powers, size, poly_size}, nv, pp, pts, std::vector<Eval>(evals, evals + num_evals));
  ORC_CATCH
}
int orc_lasso_prove_zm(void* t, const Affine* powers, size_t size, size_t poly_size, const LassoTable* tb, size_t n,
                       const uint32_t* const* dims) {
  ORC_TRY
  Transcript& tr = *(Transcript*)t;
  const USrs s{powers, size, poly_size};
  LassoPcsFns f;
  f.commit = [&](const Poly& p) { return zm_commit(s, p); };
  f.batch_open = [&](size_t nv, const std::vector<const Poly*>& polys, const std::vector<std::vector<Fr>>& pts,
                     const std::vector<Eval>& evs) { zm_batch_open(tr, s, nv, polys, pts, evs); };
  f.max_vars = 0;
  while (((size_t)2 << f.max_vars) <= poly_size) f.max_vars++;
  lasso_prove(tr, f, *tb, n, dims);
  ORC_CATCH
}
}
