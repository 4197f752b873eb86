// extern "C" boundary: thin wrappers translating C++ exceptions into lh_status codes.
#include "host.hpp"
#include <memory>

using namespace lh;

struct lh_ctx {
  Ctx c;
};
struct lh_srs {
  Srs s;
};
struct lh_mkzg_vp {
  VerifierParams* p;
};
struct lh_usrs {
  USrs s;
};
struct lh_zm_vp {
  ZmVerifierParams* p;
};

#define LH_TRY try {
#define LH_CATCH                                  \
  }                                               \
  catch (const lh::Error& e) {                    \
    lh::set_last_error(e.what());                 \
    return e.code;                                \
  }                                               \
  catch (const std::exception& e) {               \
    lh::set_last_error(e.what());                 \
    return LH_ERR_DEVICE;                         \
  }                                               \
  return LH_OK;

#define NEED(p) LH_REQUIRE((p) != nullptr, LH_ERR_ARG, "null argument: " #p)
// (a vector argument may be null only when it is empty)
#define NEED_N(p, n) LH_REQUIRE((p) != nullptr || (n) == 0, LH_ERR_ARG, "null argument: " #p)

// The current HIP device is a per-host-thread setting: every entry point that takes a ctx makes the ctx's device
// current for the duration of the call (workspace growth, SRS shards and runtime-compiled modules must land on the
// device the ctx's stream belongs to, whichever thread calls) and restores the caller's device afterwards.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) {
      LH_HIP(hipSetDevice(dev));
      switched = true;
    }
  }
  ~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
};
#define NEED_CTX(ctx) \
  NEED(ctx);          \
  DeviceGuard device_guard_((ctx)->c.device)

extern "C" {

const char* lh_last_error(void) { return lh::get_last_error(); }
const char* lh_version(void) { return "lasso-hip 0.1 (gfx950)"; }

lh_status lh_ctx_create(int device_id, lh_ctx** out) {
  LH_TRY
  NEED(out);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    throw lh::Error(LH_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  LH_REQUIRE(device_id >= 0 && device_id < count, LH_ERR_DEVICE, "device id out of range");
  LH_HIP(hipSetDevice(device_id));
  lh_ctx* ctx = new lh_ctx();
  ctx->c.device = device_id;
  LH_HIP(hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  LH_HIP(hipGetDeviceProperties(&prop, device_id));
  ctx->c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) == hipSuccess && khz > 0)
    ctx->c.wall_clock_khz = khz;
  ctx->c.pin(65536);
  ctx->c.host_trace_on = getenv("LH_HOST_TRACE") != nullptr && atoi(getenv("LH_HOST_TRACE")) != 0;
  // line 0: device -> host sequence flag; lines 1-2: host -> device mailbox of the resident sum-check tail
  LH_HIP(hipHostMalloc((void**)&ctx->c.flag, 256, hipHostMallocCoherent | hipHostMallocMapped));
  memset(ctx->c.flag, 0, 256);
  LH_HIP(hipMalloc((void**)&ctx->c.ticket, 256));  // word 0: ticket, word 8: device flag, words 32..47: the tail's relay chunks
  LH_HIP(hipMemset(ctx->c.ticket, 0, 256));
  LH_HIP(hipMalloc((void**)&ctx->c.fin_lanes, (size_t)FIN_LANE_SUMS * 64));
  LH_HIP(hipMemset(ctx->c.fin_lanes, 0, (size_t)FIN_LANE_SUMS * 64));
  *out = ctx;
  LH_CATCH
}

}  // extern "C"
namespace lh {
Ctx& ctx_helper(Ctx& c) {
  if (!c.helper) {
    lh_ctx* h = nullptr;
    if (lh_ctx_create(c.device, &h) != LH_OK || !h) throw Error(LH_ERR_DEVICE, std::string("helper ctx: ") + get_last_error());
    c.helper_handle = h;
    c.helper = &h->c;
    h->c.is_helper = true;
  }
  return *c.helper;
}
}  // namespace lh
extern "C" {

void lh_ctx_destroy(lh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->c.device);
  try {
    open_precommit_cancel(ctx->c);
  } catch (...) {
  }
  if (ctx->c.helper_handle) {
    lh_ctx_destroy((lh_ctx*)ctx->c.helper_handle);
    ctx->c.helper_handle = nullptr, ctx->c.helper = nullptr;
  }
  delete ctx->c.worker;  // (joins; after the precommit it may still be running was cancelled above)
  ctx->c.worker = nullptr;
  if (ctx->c.handoff_ev) (void)hipEventDestroy(ctx->c.handoff_ev);
  if (ctx->c.aux_stream) (void)hipStreamSynchronize(ctx->c.aux_stream), (void)hipStreamDestroy(ctx->c.aux_stream);
  if (ctx->c.aux_ev) (void)hipEventDestroy(ctx->c.aux_ev);
  for (hipEvent_t& ev : ctx->c.phase_ev)
    if (ev) (void)hipEventDestroy(ev), ev = nullptr;
  (void)hipStreamSynchronize(ctx->c.stream);
  try {
    comm_detach(ctx->c);
  } catch (...) {
  }
  if (ctx->c.gkr_mbox) (void)hipHostFree(ctx->c.gkr_mbox);
  if (ctx->c.gkr_relay) (void)hipFree(ctx->c.gkr_relay);
  if (ctx->c.comm_stage) (void)hipFree(ctx->c.comm_stage);
  if (ctx->c.pinned) (void)hipHostFree(ctx->c.pinned);
  if (ctx->c.stage) (void)hipHostFree(ctx->c.stage);
  for (int k = 0; k < 2; k++) {
    if (ctx->c.sort_stage[k]) (void)hipHostFree(ctx->c.sort_stage[k]);
    if (ctx->c.sort_ev[k]) (void)hipEventDestroy(ctx->c.sort_ev[k]);
  }
  if (ctx->c.flag) (void)hipHostFree(ctx->c.flag);
  if (ctx->c.lanes_host) (void)hipHostFree(ctx->c.lanes_host);
  if (ctx->c.ticket) (void)hipFree(ctx->c.ticket);
  if (ctx->c.fin_lanes) (void)hipFree(ctx->c.fin_lanes);
  (void)hipStreamDestroy(ctx->c.stream);
  delete ctx;
}

lh_status lh_ctx_sync(lh_ctx* ctx) {
  LH_TRY
  NEED_CTX(ctx);
  ctx->c.sync();
  LH_CATCH
}
void* lh_ctx_stream(lh_ctx* ctx) { return ctx ? (void*)ctx->c.stream : nullptr; }

lh_status lh_alloc(lh_ctx* ctx, size_t bytes, void** d_out) {
  LH_TRY
  NEED_CTX(ctx);
  NEED(d_out);
  LH_HIP(hipMalloc(d_out, bytes ? bytes : 1));
  LH_CATCH
}
lh_status lh_free(lh_ctx* ctx, void* d_ptr) {
  LH_TRY
  NEED_CTX(ctx);
  ctx->c.sync();
  if (d_ptr) LH_HIP(hipFree(d_ptr));
  LH_CATCH
}
lh_status lh_upload(lh_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  LH_TRY
  NEED_CTX(ctx);
  NEED_N(d_dst, bytes);
  NEED_N(src, bytes);
  if (bytes) {
    LH_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->c.stream));
    ctx->c.sync();
  }
  LH_CATCH
}
lh_status lh_download(lh_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  LH_TRY
  NEED_CTX(ctx);
  NEED_N(dst, bytes);
  NEED_N(d_src, bytes);
  if (bytes) {
    LH_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->c.stream));
    ctx->c.sync();
  }
  LH_CATCH
}

// ---------------------------------------------------------------- transcript
lh_status lh_keccak_transcript_new(lh_transcript** out) {
  LH_TRY
  NEED(out);
  KeccakTranscript* t = new KeccakTranscript();
  *out = &t->vt;
  LH_CATCH
}
void lh_keccak_transcript_free(lh_transcript* t) {
  if (t) delete (KeccakTranscript*)t->user;
}
lh_status lh_keccak_transcript_proof(lh_transcript* t, const uint8_t** bytes, size_t* len) {
  LH_TRY
  NEED(t);
  NEED(bytes);
  NEED(len);
  KeccakTranscript* k = (KeccakTranscript*)t->user;
  *bytes = k->stream.data();
  *len = k->stream.size();
  LH_CATCH
}
lh_status lh_keccak_transcript_from_proof(const uint8_t* proof, size_t len, lh_transcript** out) {
  LH_TRY
  NEED(out);
  LH_REQUIRE(proof || !len, LH_ERR_ARG, "null argument: proof");
  KeccakTranscript* t = new KeccakTranscript();
  t->stream.assign(proof, proof + len);
  *out = &t->vt;
  LH_CATCH
}
lh_status lh_keccak_transcript_remaining(lh_transcript* t, size_t* out) {
  LH_TRY
  NEED(t);
  NEED(out);
  KeccakTranscript* k = (KeccakTranscript*)t->user;
  *out = k->stream.size() - k->pos;
  LH_CATCH
}

// ---------------------------------------------------------------- Fr vectors
lh_status lh_fr_from_u64(lh_ctx* ctx, const uint64_t* d_in, size_t n, lh_fr* d_out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(d_in, n);
  NEED_N(d_out, n);
  k_fr_from_u64(ctx->c, d_in, n, (Fr*)d_out);
  LH_CATCH
}
lh_status lh_fr_from_u32(lh_ctx* ctx, const uint32_t* d_in, size_t n, lh_fr* d_out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(d_in, n);
  NEED_N(d_out, n);
  k_fr_from_u32(ctx->c, d_in, n, (Fr*)d_out);
  LH_CATCH
}
lh_status lh_fr_to_repr(lh_ctx* ctx, const lh_fr* d_in, size_t n, uint8_t* d_out32) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(d_in, n);
  NEED_N(d_out32, n);
  k_fr_to_repr(ctx->c, (const Fr*)d_in, n, (Fr*)d_out32);
  LH_CATCH
}
lh_status lh_fr_from_repr(lh_ctx* ctx, const uint8_t* d_in32, size_t n, lh_fr* d_out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(d_in32, n);
  NEED_N(d_out, n);
  k_fr_from_repr(ctx->c, (const Fr*)d_in32, n, (Fr*)d_out);
  LH_CATCH
}
lh_status lh_fr_add(lh_ctx* ctx, const lh_fr* a, const lh_fr* b, size_t n, lh_fr* out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(a, n);
  NEED_N(b, n);
  NEED_N(out, n);
  k_fr_binop(ctx->c, 0, (const Fr*)a, (const Fr*)b, n, (Fr*)out);
  LH_CATCH
}
lh_status lh_fr_sub(lh_ctx* ctx, const lh_fr* a, const lh_fr* b, size_t n, lh_fr* out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(a, n);
  NEED_N(b, n);
  NEED_N(out, n);
  k_fr_binop(ctx->c, 1, (const Fr*)a, (const Fr*)b, n, (Fr*)out);
  LH_CATCH
}
lh_status lh_fr_mul(lh_ctx* ctx, const lh_fr* a, const lh_fr* b, size_t n, lh_fr* out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(a, n);
  NEED_N(b, n);
  NEED_N(out, n);
  k_fr_binop(ctx->c, 2, (const Fr*)a, (const Fr*)b, n, (Fr*)out);
  LH_CATCH
}
lh_status lh_fr_mul_chain(lh_ctx* ctx, const lh_fr* a, const lh_fr* b, size_t n, int iters, lh_fr* out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(a, n);
  NEED_N(b, n);
  NEED_N(out, n);
  k_fr_mul_chain(ctx->c, (const Fr*)a, (const Fr*)b, n, iters, (Fr*)out);
  LH_CATCH
}
lh_status lh_fr_batch_invert(lh_ctx* ctx, const lh_fr* d_in, size_t n, lh_fr* d_out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(d_in, n);
  NEED_N(d_out, n);
  k_fr_batch_invert(ctx->c, (const Fr*)d_in, n, (Fr*)d_out);
  LH_CATCH
}

// ---------------------------------------------------------------- MultilinearPolynomial
static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }

lh_status lh_fix_var(lh_ctx* ctx, const lh_fr* d_in, size_t n_in, const lh_fr* x, lh_fr* d_out) {
  LH_TRY NEED_CTX(ctx);
  NEED(x);
  NEED(d_in);
  NEED(d_out);
  LH_REQUIRE(is_pow2(n_in) && n_in >= 2, LH_ERR_ARG, "fix_var: table must have 2^m >= 2 entries");
  Fr xr;
  memcpy(&xr, x, 32);
  k_fix_var(ctx->c, (const Fr*)d_in, n_in, xr, (Fr*)d_out);
  LH_CATCH
}
lh_status lh_eq_xy(lh_ctx* ctx, const lh_fr* y, size_t num_vars, lh_fr* d_out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(y, num_vars);
  NEED(d_out);
  LH_REQUIRE(num_vars < 32, LH_ERR_ARG, "eq_xy: num_vars too large");
  k_eq_xy(ctx->c, (const Fr*)y, num_vars, (Fr*)d_out);
  LH_CATCH
}
lh_status lh_evaluate(lh_ctx* ctx, const lh_fr* const* d_polys, size_t num_polys, size_t num_vars,
                      const lh_fr* point, lh_fr* out_evals) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(d_polys, num_polys);
  NEED_N(out_evals, num_polys);
  NEED_N(point, num_vars);
  std::vector<HFr> ev = evaluate_polys(ctx->c, (const Fr* const*)d_polys, num_polys, num_vars, (const HFr*)point);
  memcpy(out_evals, ev.data(), num_polys * 32);
  LH_CATCH
}
lh_status lh_lincomb(lh_ctx* ctx, const lh_fr* const* d_polys, const lh_fr* w, size_t num_polys, size_t n,
                     lh_fr* d_out) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(d_polys, num_polys);
  NEED_N(w, num_polys);
  NEED_N(d_out, n);
  k_lincomb(ctx->c, (const Fr* const*)d_polys, (const Fr*)w, num_polys, n, (Fr*)d_out);
  LH_CATCH
}

// ---------------------------------------------------------------- sum-check / GKR
lh_status lh_sumcheck_prove(lh_ctx* ctx, int prover_kind, size_t num_vars, const lh_sop* expr,
                            const lh_fr* const* d_polys, size_t num_polys, const lh_fr* ys, size_t num_ys,
                            const lh_fr* sum, lh_transcript* t, lh_fr* out_challenges, lh_fr* out_evals) {
  LH_TRY NEED_CTX(ctx);
  NEED(expr);
  NEED(sum);
  NEED_N(d_polys, num_polys);
  NEED_N(ys, num_ys);
  LH_REQUIRE(prover_kind == LH_SC_EVALUATIONS || prover_kind == LH_SC_COEFFICIENTS, LH_ERR_ARG, "bad prover kind");
  Transcript tr(t);
  HFr s;
  memcpy(&s, sum, 32);
  SumCheckResult r = sum_check_prove(ctx->c, prover_kind, num_vars, *expr, (const Fr* const*)d_polys, num_polys,
                                     (const HFr*)ys, num_ys, s, tr);
  if (out_challenges) memcpy(out_challenges, r.challenges.data(), r.challenges.size() * 32);
  if (out_evals) memcpy(out_evals, r.evals.data(), r.evals.size() * 32);
  LH_CATCH
}

lh_status lh_sumcheck_prove_expr(lh_ctx* ctx, size_t num_vars, const lh_expr* expr, const lh_fr* const* d_polys,
                                 size_t num_polys, const lh_fr* challenges, size_t num_challenges, const lh_fr* ys,
                                 size_t num_ys, const lh_fr* sum, lh_transcript* t, lh_fr* out_challenges,
                                 lh_fr* out_evals) {
  LH_TRY NEED_CTX(ctx);
  NEED(expr);
  NEED(sum);
  NEED_N(d_polys, num_polys);
  NEED_N(challenges, num_challenges);
  NEED_N(ys, num_ys);
  Transcript tr(t);
  HFr s;
  memcpy(&s, sum, 32);
  SumCheckResult r = sum_check_prove_expr(ctx->c, num_vars, *expr, (const Fr* const*)d_polys, num_polys,
                                          (const HFr*)challenges, num_challenges, (const HFr*)ys, num_ys, s, tr);
  if (out_challenges) memcpy(out_challenges, r.challenges.data(), r.challenges.size() * 32);
  if (out_evals) memcpy(out_evals, r.evals.data(), r.evals.size() * 32);
  LH_CATCH
}

lh_status lh_gkr_fractional_prove(lh_ctx* ctx, size_t num_batching, size_t num_vars,
                                  const lh_fr* const* claimed_p_0s, const lh_fr* const* claimed_q_0s,
                                  const lh_fr* const* d_ps, const lh_fr* const* d_qs, lh_transcript* t,
                                  lh_fr* out_p_xs, lh_fr* out_q_xs, lh_fr* out_x) {
  LH_TRY NEED_CTX(ctx);
  NEED_N(claimed_p_0s, num_batching);
  NEED_N(claimed_q_0s, num_batching);
  NEED_N(d_ps, num_batching);
  NEED_N(d_qs, num_batching);
  Transcript tr(t);
  FracSumCheckResult r =
      prove_fractional_sum_check(ctx->c, num_batching, num_vars, (const HFr* const*)claimed_p_0s,
                                 (const HFr* const*)claimed_q_0s, (const Fr* const*)d_ps, (const Fr* const*)d_qs, tr);
  if (out_p_xs) memcpy(out_p_xs, r.p_xs.data(), r.p_xs.size() * 32);
  if (out_q_xs) memcpy(out_q_xs, r.q_xs.data(), r.q_xs.size() * 32);
  if (out_x) memcpy(out_x, r.x.data(), r.x.size() * 32);
  LH_CATCH
}

lh_status lh_grand_product_prove(lh_ctx* ctx, size_t num_trees, const lh_fr* const* d_leaves, const size_t* num_vars,
                                 lh_transcript* t, lh_fr* out_roots, lh_fr* out_claims, lh_fr* out_points) {
  LH_TRY NEED_CTX(ctx);
  NEED(num_vars);
  NEED_N(d_leaves, num_trees);
  Transcript tr(t);
  GrandProductResult r = prove_grand_product(ctx->c, num_trees, (const Fr* const*)d_leaves, num_vars, tr);
  if (out_roots) memcpy(out_roots, r.roots.data(), num_trees * 32);
  if (out_claims) memcpy(out_claims, r.claims.data(), num_trees * 32);
  if (out_points) {
    lh_fr* p = out_points;
    for (size_t b = 0; b < num_trees; b++) {
      memcpy(p, r.points[b].data(), r.points[b].size() * 32);
      p += r.points[b].size();
    }
  }
  LH_CATCH
}

// ---------------------------------------------------------------- MSM
lh_status lh_msm(lh_ctx* ctx, const lh_fr* d_scalars, const lh_g1* d_bases, size_t n, lh_g1* out) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  NEED_N(d_scalars, n);
  NEED_N(d_bases, n);
  MsmJob job{d_scalars, false, (const G1Affine*)d_bases, n};
  msm_batch(ctx->c, &job, 1, (G1Affine*)out);
  LH_CATCH
}
lh_status lh_msm_u32(lh_ctx* ctx, const uint32_t* d_scalars, const lh_g1* d_bases, size_t n, lh_g1* out) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  NEED_N(d_scalars, n);
  NEED_N(d_bases, n);
  MsmJob job{d_scalars, true, (const G1Affine*)d_bases, n};
  msm_batch(ctx->c, &job, 1, (G1Affine*)out);
  LH_CATCH
}

// ---------------------------------------------------------------- multilinear KZG
lh_status lh_mkzg_setup(lh_ctx* ctx, const lh_fr* ss, size_t num_vars, lh_srs** out) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  NEED_N(ss, num_vars);
  Srs* s = mkzg_setup(ctx->c, (const HFr*)ss, num_vars);
  lh_srs* w = new lh_srs();
  w->s = *s;
  delete s;
  *out = w;
  LH_CATCH
}
lh_status lh_srs_upload(lh_ctx* ctx, const lh_g1* eqs_flat, size_t num_vars, lh_srs** out) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  NEED(eqs_flat);
  LH_REQUIRE(num_vars < 31, LH_ERR_ARG, "srs: num_vars too large");
  std::unique_ptr<lh_srs> w(new lh_srs());  // (a failed allocation or copy below must not leak the wrapper)
  w->s.num_vars = num_vars;
  size_t total = ((size_t)2 << num_vars) - 1;
  LH_HIP(hipMalloc((void**)&w->s.d_eqs, total * sizeof(G1Affine)));
  if (hipMemcpyAsync(w->s.d_eqs, eqs_flat, total * sizeof(G1Affine), hipMemcpyHostToDevice, ctx->c.stream) != hipSuccess ||
      hipStreamSynchronize(ctx->c.stream) != hipSuccess) {
    (void)hipFree(w->s.d_eqs);
    throw lh::Error(LH_ERR_DEVICE, "srs upload failed");
  }
  *out = w.release();
  LH_CATCH
}
lh_status lh_srs_download(lh_ctx* ctx, const lh_srs* srs, lh_g1* eqs_flat) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(eqs_flat);
  size_t total = ((size_t)2 << srs->s.num_vars) - 1;
  LH_HIP(hipMemcpyAsync(eqs_flat, srs->s.d_eqs, total * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->c.stream));
  ctx->c.sync();
  LH_CATCH
}
size_t lh_srs_num_vars(const lh_srs* srs) { return srs ? srs->s.num_vars : 0; }
void lh_srs_free(lh_ctx* ctx, lh_srs* srs) {
  if (!srs) return;
  if (ctx) (void)hipStreamSynchronize(ctx->c.stream);
  if (srs->s.d_eqs) {
    (void)hipFree(srs->s.d_eqs);
  }
  for (G1Affine* p : srs->s.shard_levels)
    if (p) {
      (void)hipFree(p);
    }
  for (auto& kv : srs->s.win_tables)
    if (kv.second.d) (void)hipFree(kv.second.d);
  delete srs;
}

lh_status lh_mkzg_commit(lh_ctx* ctx, const lh_srs* srs, const lh_fr* d_poly, size_t num_vars, lh_g1* out) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(d_poly);
  NEED(out);
  const Fr* p = (const Fr*)d_poly;
  std::vector<HG1> c = mkzg_batch_commit(ctx->c, srs->s, &p, 1, num_vars);
  memcpy(out, c.data(), 64);
  LH_CATCH
}
lh_status lh_mkzg_batch_commit(lh_ctx* ctx, const lh_srs* srs, const lh_fr* const* d_polys, size_t num_polys,
                               size_t num_vars, lh_g1* out_comms) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED_N(d_polys, num_polys);
  NEED_N(out_comms, num_polys);
  std::vector<HG1> c = mkzg_batch_commit(ctx->c, srs->s, (const Fr* const*)d_polys, num_polys, num_vars);
  if (num_polys) memcpy(out_comms, c.data(), num_polys * 64);
  LH_CATCH
}
lh_status lh_mkzg_open(lh_ctx* ctx, const lh_srs* srs, const lh_fr* d_poly, size_t num_vars, const lh_fr* point,
                       lh_transcript* t, lh_fr* out_eval) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(d_poly);
  NEED_N(point, num_vars);
  Transcript tr(t);
  HFr e = mkzg_open(ctx->c, srs->s, (const Fr*)d_poly, num_vars, (const HFr*)point, tr);
  if (out_eval) memcpy(out_eval, &e, 32);
  LH_CATCH
}
lh_status lh_mkzg_batch_open(lh_ctx* ctx, const lh_srs* srs, size_t num_vars, const lh_fr* const* d_polys,
                             size_t num_polys, const lh_fr* points, size_t num_points, const lh_evaluation* evals,
                             size_t num_evals, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED_N(d_polys, num_polys);
  NEED_N(points, num_points);
  NEED_N(evals, num_evals);
  Transcript tr(t);
  mkzg_batch_open(ctx->c, srs->s, num_vars, (const Fr* const*)d_polys, num_polys, (const HFr*)points, num_points,
                  evals, num_evals, tr);
  LH_CATCH
}

// ---------------------------------------------------------------- Lasso
lh_status lh_lasso_prove(lh_ctx* ctx, const lh_srs* srs, const lh_lasso_table* table, size_t num_vars,
                         const uint32_t* const* d_dims, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(table);
  NEED(d_dims);
  Transcript tr(t);
  lasso_prove(ctx->c, lasso_mkzg_pcs(ctx->c, srs->s), *table, num_vars, d_dims, tr);
  LH_CATCH
}
lh_status lh_lasso_last_timing(lh_ctx* ctx, double* out_ms) {
  LH_TRY NEED_CTX(ctx);
  NEED(out_ms);
  LH_HIP(hipSetDevice(ctx->c.device));
  ctx->c.phase_times_resolve();  // (the phase boundaries are events on the stream: lasso.cpp lap)
  memcpy(out_ms, ctx->c.lasso_ms, sizeof(ctx->c.lasso_ms));
  LH_CATCH
}

lh_status lh_ctx_set_option(lh_ctx* ctx, const char* name, int64_t value) {
  LH_TRY NEED_CTX(ctx);
  int64_t* slot = ctx->c.opt.find(name);
  LH_REQUIRE(slot != nullptr, LH_ERR_ARG, std::string("unknown option: ") + (name ? name : "(null)"));
  LH_REQUIRE(Options::in_range(name, value), LH_ERR_ARG, std::string("option value out of range: ") + name);
  *slot = value;
  if (slot == &ctx->c.opt.open_small_min_vars) ctx->c.opt.open_small_min_vars_forced = true;
  LH_CATCH
}
lh_status lh_ctx_get_option(lh_ctx* ctx, const char* name, int64_t* out) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  int64_t* slot = ctx->c.opt.find(name);
  LH_REQUIRE(slot != nullptr, LH_ERR_ARG, std::string("unknown option: ") + (name ? name : "(null)"));
  *out = *slot;
  LH_CATCH
}
lh_status lh_lasso_last_route(lh_ctx* ctx, lh_lasso_route* out) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  static_assert(sizeof(lh_lasso_route) == sizeof(uint32_t) * LH_LASSO_ROUTE_WORDS, "lh_lasso_route layout");
  memcpy(out, ctx->c.route.v, sizeof(*out));
  LH_CATCH
}

lh_status lh_ctx_set_comm(lh_ctx* ctx, const lh_comm* comm, size_t shard_bit) {
  LH_TRY NEED_CTX(ctx);
  ctx->c.sync();
  comm_detach(ctx->c);
  if (comm) {
    LH_REQUIRE(comm->size >= 1 && (comm->size & (comm->size - 1)) == 0 && comm->rank >= 0 && comm->rank < comm->size,
               LH_ERR_ARG, "communicator: size must be a power of two and 0 <= rank < size");
    LH_REQUIRE(comm->all_gather || comm->all_gather_device, LH_ERR_ARG, "communicator: no all_gather callback");
    ctx->c.comm = *comm;
    ctx->c.has_comm = true;
    ctx->c.shard_bit = shard_bit;
    ctx->c.comm_stats[0] = ctx->c.comm_stats[1] = 0;
  }
  LH_CATCH
}
lh_status lh_rccl_unique_id(uint8_t out[LH_RCCL_UNIQUE_ID_BYTES]) {
  LH_TRY
  NEED(out);
  rccl_unique_id(out);
  LH_CATCH
}
lh_status lh_ctx_set_comm_rccl(lh_ctx* ctx, int rank, int size, const uint8_t unique_id[LH_RCCL_UNIQUE_ID_BYTES],
                               size_t shard_bit) {
  LH_TRY NEED_CTX(ctx);
  NEED(unique_id);
  LH_REQUIRE(size >= 1 && (size & (size - 1)) == 0 && rank >= 0 && rank < size, LH_ERR_ARG,
             "communicator: size must be a power of two and 0 <= rank < size");
  ctx->c.sync();
  comm_attach_rccl(ctx->c, rank, size, unique_id, shard_bit);
  ctx->c.comm_stats[0] = ctx->c.comm_stats[1] = 0;
  LH_CATCH
}
lh_status lh_ctx_set_comm_loopback(lh_ctx* ctx, int rank, int size, size_t shard_bit) {
  LH_TRY NEED_CTX(ctx);
  LH_REQUIRE(size >= 1 && (size & (size - 1)) == 0 && rank >= 0 && rank < size, LH_ERR_ARG,
             "communicator: size must be a power of two and 0 <= rank < size");
  ctx->c.sync();
  comm_attach_loopback(ctx->c, rank, size, shard_bit);
  ctx->c.comm_stats[0] = ctx->c.comm_stats[1] = 0;
  LH_CATCH
}
lh_status lh_ctx_comm_stats(lh_ctx* ctx, uint64_t out[2]) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  out[0] = ctx->c.comm_stats[0];
  out[1] = ctx->c.comm_stats[1];
  LH_CATCH
}
lh_status lh_ctx_comm_phase_stats(lh_ctx* ctx, uint64_t out[16], int reset) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  for (int p = 0; p < 8; p++) {
    out[2 * p] = ctx->c.comm_phase_stats[p][0];
    out[2 * p + 1] = ctx->c.comm_phase_stats[p][1];
    if (reset) ctx->c.comm_phase_stats[p][0] = ctx->c.comm_phase_stats[p][1] = 0;
  }
  LH_CATCH
}
lh_status lh_ctx_memory_stats(lh_ctx* ctx, uint64_t out[4]) {
  LH_TRY NEED_CTX(ctx);
  NEED(out);
  out[0] = ctx->c.arena.high_water() + (ctx->c.helper ? ctx->c.helper->arena.high_water() : 0);
  out[1] = ctx->c.arena.reserved() + (ctx->c.helper ? ctx->c.helper->arena.reserved() : 0);
  size_t free_b = 0, total_b = 0;
  LH_HIP(hipMemGetInfo(&free_b, &total_b));
  out[2] = free_b, out[3] = total_b;
  LH_CATCH
}
lh_status lh_ctx_host_cpus(lh_ctx* ctx, char* bus_id, size_t bus_id_cap, char* cpulist, size_t cpulist_cap) {
  LH_TRY NEED_CTX(ctx);
  NEED(bus_id);
  NEED(cpulist);
  LH_REQUIRE(bus_id_cap >= 16 && cpulist_cap >= 2, LH_ERR_ARG, "host cpus: buffers too small");
  char id[64] = {0};
  LH_HIP(hipDeviceGetPCIBusId(id, (int)sizeof(id), ctx->c.device));
  for (char* p = id; *p; p++) *p = (char)tolower((unsigned char)*p);  // (sysfs spells the address in lower case)
  snprintf(bus_id, bus_id_cap, "%s", id);
  cpulist[0] = 0;
  const std::string path = std::string("/sys/bus/pci/devices/") + id + "/local_cpulist";
  if (FILE* f = fopen(path.c_str(), "r")) {
    if (fgets(cpulist, (int)cpulist_cap, f)) {
      size_t n = strlen(cpulist);
      while (n && (cpulist[n - 1] == '\n' || cpulist[n - 1] == ' ')) cpulist[--n] = 0;
    } else {
      cpulist[0] = 0;
    }
    fclose(f);
  }
  LH_CATCH
}
lh_status lh_lasso_prove_sharded(lh_ctx* ctx, const lh_srs* srs, const lh_lasso_table* table, size_t num_vars,
                                 const uint32_t* const* d_dims, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(table);
  NEED(d_dims);
  Transcript tr(t);
  lasso_prove_sharded(ctx->c, srs->s, *table, num_vars, d_dims, tr);
  LH_CATCH
}

// the phase loop's arguments as the prover wants them (hyperplonk.rs:185-205): shared by both PCS entry points
static HpPhases hp_phases_of(const lh_hp_param* pp, size_t num_phases, const size_t* num_witness_polys,
                             const size_t* num_challenges, const lh_hp_circuit* circuit) {
  LH_REQUIRE(circuit->synthesize, LH_ERR_ARG, "circuit: synthesize callback missing");
  LH_REQUIRE(num_phases == 0 || (num_witness_polys && num_challenges), LH_ERR_ARG, "null argument: phases");
  HpPhases ph;
  ph.num_witness_polys.assign(num_witness_polys, num_witness_polys + num_phases);
  ph.num_challenges.assign(num_challenges, num_challenges + num_phases);
  size_t tw = 0, tc = 0;
  for (size_t r = 0; r < num_phases; r++) tw += num_witness_polys[r], tc += num_challenges[r];
  LH_REQUIRE(tw == pp->num_witness_polys && tc == pp->num_challenges, LH_ERR_ARG,
             "hyperplonk: phases do not add up to num_witness_polys / num_challenges");
  const std::vector<size_t> per_phase = ph.num_witness_polys;
  ph.synthesize = [circuit, per_phase](size_t round, const std::vector<HFr>& challenges) {
    std::vector<const void*> out(per_phase[round], nullptr);
    int rc = circuit->synthesize(circuit->user, round, (const lh_fr*)challenges.data(), challenges.size(), out.data(),
                                 out.size());
    if (rc != LH_OK) throw lh::Error(rc < 0 ? rc : LH_ERR_INVALID_SNARK, "circuit synthesize callback failed");
    std::vector<const Fr*> w;
    for (const void* p : out) {
      LH_REQUIRE(p != nullptr, LH_ERR_ARG, "circuit synthesize left a witness poly unset");
      w.push_back((const Fr*)p);
    }
    return w;
  };
  return ph;
}

lh_status lh_hyperplonk_prove_phases(lh_ctx* ctx, const lh_srs* srs, const lh_hp_param* pp, size_t num_phases,
                                     const size_t* num_witness_polys, const size_t* num_challenges,
                                     const lh_fr* const* instances, const lh_hp_circuit* circuit, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(pp);
  NEED(circuit);
  Transcript tr(t);
  const HpPhases ph = hp_phases_of(pp, num_phases, num_witness_polys, num_challenges, circuit);
  hyperplonk_prove_phases(ctx->c, mkzg_pcs(ctx->c, srs->s), *pp, ph, (const HFr* const*)instances, tr);
  LH_CATCH
}

lh_status lh_hyperplonk_prove(lh_ctx* ctx, const lh_srs* srs, const lh_hp_param* pp, const lh_fr* const* instances,
                              const lh_fr* const* d_witness_polys, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(pp);
  NEED_N(d_witness_polys, pp->num_witness_polys);
  Transcript tr(t);
  hyperplonk_prove(ctx->c, mkzg_pcs(ctx->c, srs->s), *pp, (const HFr* const*)instances, (const Fr* const*)d_witness_polys, tr);
  LH_CATCH
}

lh_status lh_shard_extract(lh_ctx* ctx, const void* d_global, size_t n_local, size_t shard_bit, size_t rho, size_t rank,
                           size_t elem_bytes, void* d_local) {
  LH_TRY NEED_CTX(ctx);
  NEED(d_global);
  NEED(d_local);
  LH_REQUIRE(rho < 16 && rank < ((size_t)1 << rho) && shard_bit < 40, LH_ERR_ARG, "shard_extract: bad geometry");
  k_shard_extract(ctx->c, d_global, n_local, shard_bit, rho, rank, elem_bytes, d_local);
  ctx->c.sync();
  LH_CATCH
}
lh_status lh_hyperplonk_prove_sharded(lh_ctx* ctx, const lh_srs* srs, const lh_hp_param* pp, const lh_fr* const* instances,
                                      const lh_fr* const* d_witness_polys, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(pp);
  NEED_N(d_witness_polys, pp->num_witness_polys);
  Ctx& c = ctx->c;
  LH_REQUIRE(c.has_comm, LH_ERR_ARG, "lh_hyperplonk_prove_sharded: no communicator attached");
  const size_t R = (size_t)c.comm.size;
  LH_REQUIRE(R >= 1 && (R & (R - 1)) == 0, LH_ERR_ARG, "sharded prove: the number of ranks must be a power of two");
  struct Active {
    Ctx& c;
    explicit Active(Ctx& c_) : c(c_) { c.shard_active = true; }
    ~Active() { c.shard_active = false; }
  } active(c);
  Transcript tr(t);
  hyperplonk_prove(c, mkzg_pcs(c, srs->s), *pp, (const HFr* const*)instances, (const Fr* const*)d_witness_polys, tr);
  LH_CATCH
}

// ---------------------------------------------------------------- verifiers (host only)
lh_status lh_mkzg_vp_setup(const lh_fr* ss, size_t num_vars, lh_mkzg_vp** out) {
  LH_TRY
  NEED(out);
  LH_REQUIRE(ss || !num_vars, LH_ERR_ARG, "null argument: ss");
  *out = new lh_mkzg_vp{mkzg_vp_setup((const HFr*)ss, num_vars)};
  LH_CATCH
}
lh_status lh_mkzg_vp_new(const lh_g1* g1, const lh_g2* g2, const lh_g2* ss, size_t num_vars, lh_mkzg_vp** out) {
  LH_TRY
  NEED(out);
  NEED(g1);
  NEED(g2);
  LH_REQUIRE(ss || !num_vars, LH_ERR_ARG, "null argument: ss");
  *out = new lh_mkzg_vp{mkzg_vp_new(*g1, *g2, ss, num_vars)};
  LH_CATCH
}
lh_status lh_mkzg_vp_export(const lh_mkzg_vp* vp, lh_g1* g1, lh_g2* g2, lh_g2* ss) {
  LH_TRY
  NEED(vp);
  NEED(g1);
  NEED(g2);
  NEED_N(ss, mkzg_vp_num_vars(*vp->p));
  mkzg_vp_export(*vp->p, g1, g2, ss);
  LH_CATCH
}
size_t lh_mkzg_vp_num_vars(const lh_mkzg_vp* vp) { return vp ? mkzg_vp_num_vars(*vp->p) : 0; }
void lh_mkzg_vp_free(lh_mkzg_vp* vp) {
  if (!vp) return;
  mkzg_vp_free(vp->p);
  delete vp;
}
lh_status lh_pairing_check(const lh_g1* ps, const lh_g2* qs, size_t n, int* out_is_identity) {
  LH_TRY
  NEED(out_is_identity);
  LH_REQUIRE((ps && qs) || !n, LH_ERR_ARG, "null argument: points");
  *out_is_identity = pairing_check(ps, qs, n) ? 1 : 0;
  LH_CATCH
}
lh_status lh_mkzg_verify(const lh_mkzg_vp* vp, const lh_g1* comm, const lh_fr* point, size_t num_vars,
                         const lh_fr* eval, lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(comm);
  NEED(eval);
  LH_REQUIRE(point || !num_vars, LH_ERR_ARG, "null argument: point");
  Transcript tr(t);
  HG1 c;
  memcpy(&c, comm, sizeof(c));
  HFr e;
  memcpy(&e, eval, 32);
  mkzg_verify(*vp->p, c, (const HFr*)point, num_vars, e, tr);
  LH_CATCH
}
lh_status lh_mkzg_batch_verify(const lh_mkzg_vp* vp, size_t num_vars, const lh_g1* comms, size_t num_comms,
                               const lh_fr* points, size_t num_points, const lh_evaluation* evals, size_t num_evals,
                               lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED_N(comms, num_comms);
  NEED_N(points, num_points);
  NEED_N(evals, num_evals);
  Transcript tr(t);
  mkzg_batch_verify(*vp->p, num_vars, (const HG1*)comms, num_comms, (const HFr*)points, num_points, evals, num_evals,
                    tr);
  LH_CATCH
}
lh_status lh_sumcheck_verify(int prover_kind, size_t num_vars, size_t degree, const lh_fr* sum, lh_transcript* t,
                             lh_fr* out_eval, lh_fr* out_x) {
  LH_TRY
  NEED(sum);
  Transcript tr(t);
  HFr s;
  memcpy(&s, sum, 32);
  auto res = sum_check_verify(prover_kind, num_vars, degree, s, tr);
  if (out_eval) memcpy(out_eval, &res.first, 32);
  if (out_x) memcpy(out_x, res.second.data(), 32 * res.second.size());
  LH_CATCH
}
lh_status lh_lasso_verify(const lh_mkzg_vp* vp, const lh_lasso_table* table, size_t num_vars, lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(table);
  Transcript tr(t);
  const VerifierParams& pcs = *vp->p;
  lasso_verify([&pcs](size_t nv, const HG1* comms, size_t nc, const HFr* points, size_t np, const lh_evaluation* evals,
                      size_t ne, Transcript& t2) { mkzg_batch_verify(pcs, nv, comms, nc, points, np, evals, ne, t2); },
               *table, num_vars, tr);
  LH_CATCH
}
lh_status lh_hyperplonk_verify(const lh_mkzg_vp* vp, const lh_hp_vparam* hvp, const lh_fr* const* instances,
                               lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(hvp);
  Transcript tr(t);
  const VerifierParams& pcs = *vp->p;
  hyperplonk_verify([&pcs](size_t nv, const HG1* comms, size_t nc, const HFr* points, size_t np, const lh_evaluation* evals,
                           size_t ne, Transcript& t2) { mkzg_batch_verify(pcs, nv, comms, nc, points, np, evals, ne, t2); },
                    *hvp, (const HFr* const*)instances, tr);
  LH_CATCH
}
lh_status lh_hyperplonk_verify_phases(const lh_mkzg_vp* vp, const lh_hp_vparam* hvp, size_t num_phases,
                                      const size_t* num_witness_polys, const size_t* num_challenges,
                                      const lh_fr* const* instances, lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(hvp);
  LH_REQUIRE(num_phases == 0 || (num_witness_polys && num_challenges), LH_ERR_ARG, "null argument: phases");
  Transcript tr(t);
  const VerifierParams& pcs = *vp->p;
  hyperplonk_verify_phases(
      [&pcs](size_t nv, const HG1* comms, size_t nc, const HFr* points, size_t np, const lh_evaluation* evals, size_t ne,
             Transcript& t2) { mkzg_batch_verify(pcs, nv, comms, nc, points, np, evals, ne, t2); },
      *hvp, std::vector<size_t>(num_witness_polys, num_witness_polys + num_phases),
      std::vector<size_t>(num_challenges, num_challenges + num_phases), (const HFr* const*)instances, tr);
  LH_CATCH
}

// ---------------------------------------------------------------- Zeromorph over univariate KZG
lh_status lh_ukzg_setup(lh_ctx* ctx, const lh_fr* s, size_t poly_size, lh_usrs** out) {
  LH_TRY NEED_CTX(ctx);
  NEED(s);
  NEED(out);
  HFr sv;
  memcpy(&sv, s, 32);
  USrs* u = ukzg_setup(ctx->c, sv, poly_size);
  *out = new lh_usrs{*u};
  delete u;
  LH_CATCH
}
lh_status lh_usrs_upload(lh_ctx* ctx, const lh_g1* powers, size_t poly_size, lh_usrs** out) {
  LH_TRY NEED_CTX(ctx);
  NEED(powers);
  NEED(out);
  LH_REQUIRE(poly_size >= 1 && poly_size < ((size_t)1 << 31), LH_ERR_ARG, "univariate srs: bad poly_size");
  std::unique_ptr<lh_usrs> w(new lh_usrs());
  w->s.size = poly_size;
  LH_HIP(hipMalloc((void**)&w->s.d_powers, poly_size * sizeof(G1Affine)));
  if (hipMemcpyAsync(w->s.d_powers, powers, poly_size * sizeof(G1Affine), hipMemcpyHostToDevice, ctx->c.stream) != hipSuccess ||
      hipStreamSynchronize(ctx->c.stream) != hipSuccess) {
    (void)hipFree(w->s.d_powers);
    throw lh::Error(LH_ERR_DEVICE, "univariate srs upload failed");
  }
  *out = w.release();
  LH_CATCH
}
lh_status lh_usrs_download(lh_ctx* ctx, const lh_usrs* srs, lh_g1* powers) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(powers);
  LH_HIP(hipMemcpyAsync(powers, srs->s.d_powers, srs->s.size * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->c.stream));
  ctx->c.sync();
  LH_CATCH
}
size_t lh_usrs_size(const lh_usrs* srs) { return srs ? srs->s.size : 0; }
void lh_usrs_free(lh_ctx* ctx, lh_usrs* srs) {
  if (!srs) return;
  if (ctx) (void)hipStreamSynchronize(ctx->c.stream);
  if (srs->s.d_powers) (void)hipFree(srs->s.d_powers);
  delete srs;
}
lh_status lh_zeromorph_batch_commit(lh_ctx* ctx, const lh_usrs* srs, size_t poly_size, const lh_fr* const* d_polys,
                                    size_t num_polys, size_t num_vars, lh_g1* out_comms) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  LH_REQUIRE((d_polys && out_comms) || !num_polys, LH_ERR_ARG, "null argument: polys");
  std::vector<HG1> c = zeromorph_batch_commit(ctx->c, srs->s, poly_size, (const Fr* const*)d_polys, num_polys, num_vars);
  if (num_polys) memcpy(out_comms, c.data(), 64 * num_polys);
  LH_CATCH
}
lh_status lh_zeromorph_open(lh_ctx* ctx, const lh_usrs* srs, size_t poly_size, const lh_fr* d_poly, size_t num_vars,
                            const lh_fr* point, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(d_poly);
  NEED_N(point, num_vars);
  Transcript tr(t);
  zeromorph_open(ctx->c, srs->s, poly_size, (const Fr*)d_poly, num_vars, (const HFr*)point, tr);
  LH_CATCH
}
lh_status lh_zeromorph_batch_open(lh_ctx* ctx, const lh_usrs* srs, size_t poly_size, size_t num_vars,
                                  const lh_fr* const* d_polys, size_t num_polys, const lh_fr* points,
                                  size_t num_points, const lh_evaluation* evals, size_t num_evals, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED_N(d_polys, num_polys);
  NEED_N(points, num_points);
  NEED_N(evals, num_evals);
  Transcript tr(t);
  zeromorph_batch_open(ctx->c, srs->s, poly_size, num_vars, (const Fr* const*)d_polys, num_polys, (const HFr*)points,
                       num_points, evals, num_evals, tr);
  LH_CATCH
}
lh_status lh_zeromorph_vp_setup(const lh_fr* s, size_t param_size, size_t poly_size, lh_zm_vp** out) {
  LH_TRY
  NEED(s);
  NEED(out);
  HFr sv;
  memcpy(&sv, s, 32);
  *out = new lh_zm_vp{zeromorph_vp_setup(sv, param_size, poly_size)};
  LH_CATCH
}
lh_status lh_zeromorph_vp_new(const lh_g1* g1, const lh_g2* g2, const lh_g2* s_g2, const lh_g2* s_offset_g2,
                              lh_zm_vp** out) {
  LH_TRY
  NEED(g1);
  NEED(g2);
  NEED(s_g2);
  NEED(s_offset_g2);
  NEED(out);
  *out = new lh_zm_vp{zeromorph_vp_new(*g1, *g2, *s_g2, *s_offset_g2)};
  LH_CATCH
}
lh_status lh_zeromorph_vp_export(const lh_zm_vp* vp, lh_g1* g1, lh_g2* g2, lh_g2* s_g2, lh_g2* s_offset_g2) {
  LH_TRY
  NEED(vp);
  NEED(g1);
  NEED(g2);
  NEED(s_g2);
  NEED(s_offset_g2);
  zeromorph_vp_export(*vp->p, g1, g2, s_g2, s_offset_g2);
  LH_CATCH
}
void lh_zeromorph_vp_free(lh_zm_vp* vp) {
  if (!vp) return;
  zeromorph_vp_free(vp->p);
  delete vp;
}
lh_status lh_zeromorph_verify(const lh_zm_vp* vp, const lh_g1* comm, const lh_fr* point, size_t num_vars,
                              const lh_fr* eval, lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(comm);
  NEED(eval);
  LH_REQUIRE(point || !num_vars, LH_ERR_ARG, "null argument: point");
  Transcript tr(t);
  HG1 c;
  memcpy(&c, comm, sizeof(c));
  HFr e;
  memcpy(&e, eval, 32);
  zeromorph_verify(*vp->p, c, (const HFr*)point, num_vars, e, tr);
  LH_CATCH
}
lh_status lh_zeromorph_batch_verify(const lh_zm_vp* vp, size_t num_vars, const lh_g1* comms, size_t num_comms,
                                    const lh_fr* points, size_t num_points, const lh_evaluation* evals,
                                    size_t num_evals, lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED_N(comms, num_comms);
  NEED_N(points, num_points);
  NEED_N(evals, num_evals);
  Transcript tr(t);
  zeromorph_batch_verify(*vp->p, num_vars, (const HG1*)comms, num_comms, (const HFr*)points, num_points, evals,
                         num_evals, tr);
  LH_CATCH
}

lh_status lh_lasso_prove_zeromorph(lh_ctx* ctx, const lh_usrs* srs, size_t poly_size, const lh_lasso_table* table,
                                   size_t num_vars, const uint32_t* const* d_dims, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(table);
  NEED(d_dims);
  Transcript tr(t);
  lasso_prove(ctx->c, lasso_zeromorph_pcs(ctx->c, srs->s, poly_size), *table, num_vars, d_dims, tr);
  LH_CATCH
}
lh_status lh_lasso_verify_zeromorph(const lh_zm_vp* vp, const lh_lasso_table* table, size_t num_vars, lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(table);
  Transcript tr(t);
  const ZmVerifierParams& pcs = *vp->p;
  lasso_verify([&pcs](size_t nv, const HG1* comms, size_t nc, const HFr* points, size_t np, const lh_evaluation* evals,
                      size_t ne, Transcript& t2) { zeromorph_batch_verify(pcs, nv, comms, nc, points, np, evals, ne, t2); },
               *table, num_vars, tr);
  LH_CATCH
}
lh_status lh_hyperplonk_prove_zeromorph(lh_ctx* ctx, const lh_usrs* srs, size_t poly_size, const lh_hp_param* pp,
                                        const lh_fr* const* instances, const lh_fr* const* d_witness_polys,
                                        lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(pp);
  NEED_N(d_witness_polys, pp->num_witness_polys);
  Transcript tr(t);
  hyperplonk_prove(ctx->c, zeromorph_pcs(ctx->c, srs->s, poly_size), *pp, (const HFr* const*)instances,
                   (const Fr* const*)d_witness_polys, tr);
  LH_CATCH
}
lh_status lh_hyperplonk_verify_zeromorph(const lh_zm_vp* vp, const lh_hp_vparam* hvp, const lh_fr* const* instances,
                                         lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(hvp);
  Transcript tr(t);
  const ZmVerifierParams& pcs = *vp->p;
  hyperplonk_verify([&pcs](size_t nv, const HG1* comms, size_t nc, const HFr* points, size_t np, const lh_evaluation* evals,
                           size_t ne, Transcript& t2) { zeromorph_batch_verify(pcs, nv, comms, nc, points, np, evals, ne, t2); },
                    *hvp, (const HFr* const*)instances, tr);
  LH_CATCH
}

lh_status lh_hyperplonk_prove_phases_zeromorph(lh_ctx* ctx, const lh_usrs* srs, size_t poly_size, const lh_hp_param* pp,
                                               size_t num_phases, const size_t* num_witness_polys,
                                               const size_t* num_challenges, const lh_fr* const* instances,
                                               const lh_hp_circuit* circuit, lh_transcript* t) {
  LH_TRY NEED_CTX(ctx);
  NEED(srs);
  NEED(pp);
  NEED(circuit);
  Transcript tr(t);
  const HpPhases ph = hp_phases_of(pp, num_phases, num_witness_polys, num_challenges, circuit);
  hyperplonk_prove_phases(ctx->c, zeromorph_pcs(ctx->c, srs->s, poly_size), *pp, ph, (const HFr* const*)instances, tr);
  LH_CATCH
}
lh_status lh_hyperplonk_verify_phases_zeromorph(const lh_zm_vp* vp, const lh_hp_vparam* hvp, size_t num_phases,
                                                const size_t* num_witness_polys, const size_t* num_challenges,
                                                const lh_fr* const* instances, lh_transcript* t) {
  LH_TRY
  NEED(vp);
  NEED(hvp);
  LH_REQUIRE(num_phases == 0 || (num_witness_polys && num_challenges), LH_ERR_ARG, "null argument: phases");
  Transcript tr(t);
  const ZmVerifierParams& pcs = *vp->p;
  hyperplonk_verify_phases(
      [&pcs](size_t nv, const HG1* comms, size_t nc, const HFr* points, size_t np, const lh_evaluation* evals, size_t ne,
             Transcript& t2) { zeromorph_batch_verify(pcs, nv, comms, nc, points, np, evals, ne, t2); },
      *hvp, std::vector<size_t>(num_witness_polys, num_witness_polys + num_phases),
      std::vector<size_t>(num_challenges, num_challenges + num_phases), (const HFr* const*)instances, tr);
  LH_CATCH
}

lh_status lh_debug_jit_source(const uint32_t* code, size_t num_instrs, uint32_t num_regs, uint32_t result_reg, int degree,
                              char* out, size_t cap, size_t* len) {
  LH_TRY
  NEED(code);
  NEED(len);
  LH_REQUIRE(num_regs >= 1 && num_regs <= 16 && result_reg < num_regs && degree >= 1, LH_ERR_ARG, "jit source: bad program header");
  const std::string src = lh::jit_debug_source(code, num_instrs, num_regs, result_reg, degree);
  *len = src.size();
  if (out && cap) {
    const size_t n = std::min(cap - 1, src.size());
    memcpy(out, src.data(), n);
    out[n] = 0;
  }
  LH_CATCH
}

lh_status lh_profile_enable(lh_ctx* ctx, int on) {
  LH_TRY NEED_CTX(ctx);
  ctx->c.sync();
  ctx->c.prof = on == 1;
  ctx->c.prof_recs.clear();
  // 2: live records of the bucket-accumulation launches only (dev.hpp Ctx::live); the helper ctx's launches count too
  std::vector<ProfRec> drop;
  ctx->c.live_resolve(drop);
  ctx->c.live = on == 2;
  if (ctx->c.helper) {
    ctx->c.helper->live_resolve(drop);
    ctx->c.helper->live = on == 2;
  }
  LH_CATCH
}
lh_status lh_profile_read(lh_ctx* ctx, lh_prof_rec* out, size_t cap, size_t* count) {
  LH_TRY NEED_CTX(ctx);
  NEED(count);
  static_assert(sizeof(lh_prof_rec) == sizeof(lh::ProfRec), "profile record layout");
  if (ctx->c.live) {  // (resolved at the first read after the timed region: waits for the streams)
    ctx->c.live_resolve(ctx->c.prof_recs);
    if (ctx->c.helper) {
      LH_HIP(hipStreamSynchronize(ctx->c.helper->stream));
      ctx->c.helper->live_resolve(ctx->c.prof_recs);
    }
  }
  size_t n = ctx->c.prof_recs.size();
  *count = n;
  if (out && cap) memcpy(out, ctx->c.prof_recs.data(), (n < cap ? n : cap) * sizeof(lh_prof_rec));
  LH_CATCH
}

}  // extern "C"
