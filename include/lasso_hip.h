/* lasso_hip.h -- C-ABI of the MI355X-native prover hot path (liblasso_hip.so).
 *
 * The reference (DoHoonKim8/halo2-lasso, Rust) has no FFI: its seam is a set of generic traits
 * (SURVEY.md §8b).  Every entry point below names the reference item it replaces; a Rust shim
 * implementing those traits binds exactly these symbols.
 *
 * Conventions
 *  - Field elements cross the ABI in halo2curves' in-memory form: 4 x u64 little-endian limbs,
 *    Montgomery (R = 2^256).  `lh_fr` is therefore bit-identical to a Rust `bn256::Fr`, and
 *    `lh_g1` ({x, y} in Fq, identity = (0,0)) to `bn256::G1Affine`.
 *  - `d_*` arguments are DEVICE pointers (from lh_alloc, or any HIP allocation such as a torch
 *    tensor's data_ptr on the ctx's device); everything else is host memory owned by the caller
 *    for the duration of the call.
 *  - Every function returns an `lh_status`; 0 is success, negatives mirror
 *    plonkish_backend::Error (plonkish_backend/src/lib.rs:12-20).  lh_last_error() gives the
 *    message of the last failure on the calling thread.
 *  - One ctx per process per GPU; calls on one ctx are not re-entrant (the reference calls
 *    `prove` from one thread and fans out on rayon, util/parallel.rs:9-46).  A ctx may start host threads of its own and a
 *    helper ctx on the same device (second stream, own arena and pinned blocks: lh_lasso_prove commits the challenge-free
 *    part of its opening there, option open_precommit); both end with the call that started them / with lh_ctx_destroy.
 *  - There is NO CPU fallback: without a usable HIP device lh_ctx_create fails with
 *    LH_ERR_DEVICE and nothing else can be called.
 *  - Every entry point that takes an lh_ctx makes the ctx's HIP device current for the duration of the
 *    call (and restores the caller's): a ctx may be used from any host thread, one call at a time.
 *  - The last rounds of a sum-check run inside ONE resident kernel that exchanges message and challenge
 *    with the calling thread through pinned memory while the lh_* call is in progress.  The kernel waits a
 *    bounded time for each challenge (environment LH_SC_TAIL_TIMEOUT_MS, default 2000); when the calling
 *    thread stalls longer (debugger, SIGSTOP, slow transcript callback) the prover resumes with launched
 *    rounds - same proof bytes, no error.  LH_SC_TAIL=0 disables the resident rounds.  (The kernel is at most 64
 *    workgroups that wait for the host only, never for one another or for another kernel; messages and challenges
 *    cross as 16-byte chunks that carry their own sequence number, written with single 16-byte stores: the host
 *    side is x86-64 with SSE2.)
 */
#ifndef LASSO_HIP_H
#define LASSO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } lh_fr;      /* bn256::Fr, Montgomery */
typedef struct { uint64_t x[4], y[4]; } lh_g1; /* bn256::G1Affine, Montgomery Fq coords */

typedef int lh_status;
#define LH_OK 0
#define LH_ERR_INVALID_SUMCHECK (-1)  /* Error::InvalidSumcheck  lib.rs:14 */
#define LH_ERR_INVALID_PCS_PARAM (-2) /* Error::InvalidPcsParam  lib.rs:15 */
#define LH_ERR_INVALID_PCS_OPEN (-3)  /* Error::InvalidPcsOpen   lib.rs:16 */
#define LH_ERR_INVALID_SNARK (-4)     /* Error::InvalidSnark     lib.rs:17 */
#define LH_ERR_SERIALIZATION (-5)     /* Error::Serialization    lib.rs:18 */
#define LH_ERR_TRANSCRIPT (-6)        /* Error::Transcript       lib.rs:19 */
#define LH_ERR_DEVICE (-7)            /* no HIP device / HIP runtime failure */
#define LH_ERR_ARG (-8)               /* assert!/panic in the reference: bad argument */

typedef struct lh_ctx lh_ctx;

const char* lh_last_error(void);
const char* lh_version(void);

/* ---------------------------------------------------------------- context & memory */
lh_status lh_ctx_create(int device_id, lh_ctx** out);
void lh_ctx_destroy(lh_ctx* ctx);
lh_status lh_ctx_sync(lh_ctx* ctx);
/* opaque hipStream_t the ctx launches on (for bench.py's HIP-event timing) */
void* lh_ctx_stream(lh_ctx* ctx);
lh_status lh_alloc(lh_ctx* ctx, size_t bytes, void** d_out);
lh_status lh_free(lh_ctx* ctx, void* d_ptr);
lh_status lh_upload(lh_ctx* ctx, void* d_dst, const void* src, size_t bytes);
lh_status lh_download(lh_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ---------------------------------------------------------------- transcript
 * Callback table standing for `&mut impl TranscriptWrite<G1Affine, Fr>`
 * (util/transcript.rs:15-97).  Each callback returns 0 or a negative lh_status. */
typedef struct lh_transcript {
  void* user;
  int (*write_field_element)(void* user, const lh_fr* fe);  /* transcript.rs:157-165 */
  int (*common_field_element)(void* user, const lh_fr* fe); /* transcript.rs:133-136 */
  int (*squeeze_challenge)(void* user, lh_fr* out);          /* transcript.rs:126-131 */
  int (*write_commitment)(void* user, const lh_g1* pt);     /* transcript.rs:213-226 */
  int (*common_commitment)(void* user, const lh_g1* pt);    /* transcript.rs:170-183 */
  /* TranscriptRead half (verifiers only; may be NULL on a write-only transcript) */
  int (*read_field_element)(void* user, lh_fr* out);         /* transcript.rs:138-154 */
  int (*read_commitment)(void* user, lh_g1* out);            /* transcript.rs:185-210 */
} lh_transcript;

/* Built-in Keccak256Transcript<Cursor<Vec<u8>>> (transcript.rs:99-121) */
lh_status lh_keccak_transcript_new(lh_transcript** out);
void lh_keccak_transcript_free(lh_transcript* t);
/* InMemoryTranscript::into_proof (transcript.rs:110-112): pointer valid until the next write */
lh_status lh_keccak_transcript_proof(lh_transcript* t, const uint8_t** bytes, size_t* len);
/* InMemoryTranscript::from_proof (transcript.rs:114-123): a reading transcript over a copy of `proof` */
lh_status lh_keccak_transcript_from_proof(const uint8_t* proof, size_t len, lh_transcript** out);
/* bytes of the proof not yet read */
lh_status lh_keccak_transcript_remaining(lh_transcript* t, size_t* out);

/* ---------------------------------------------------------------- a1: Fr arithmetic
 * halo2_curves bn256::Fr ops (via util/arithmetic.rs:15-22) over device vectors. */
lh_status lh_fr_from_u64(lh_ctx*, const uint64_t* d_in, size_t n, lh_fr* d_out); /* Fr::from(u64) */
lh_status lh_fr_from_u32(lh_ctx*, const uint32_t* d_in, size_t n, lh_fr* d_out);
lh_status lh_fr_to_repr(lh_ctx*, const lh_fr* d_in, size_t n, uint8_t* d_out32); /* to_repr(): canonical LE */
lh_status lh_fr_from_repr(lh_ctx*, const uint8_t* d_in32, size_t n, lh_fr* d_out); /* must be < r */
lh_status lh_fr_add(lh_ctx*, const lh_fr* d_a, const lh_fr* d_b, size_t n, lh_fr* d_out);
lh_status lh_fr_sub(lh_ctx*, const lh_fr* d_a, const lh_fr* d_b, size_t n, lh_fr* d_out);
lh_status lh_fr_mul(lh_ctx*, const lh_fr* d_a, const lh_fr* d_b, size_t n, lh_fr* d_out);
/* `iters` dependent multiplications per element: peak Fr-mul/s micro-benchmark */
lh_status lh_fr_mul_chain(lh_ctx*, const lh_fr* d_a, const lh_fr* d_b, size_t n, int iters, lh_fr* d_out);
/* ff::BatchInvert (call sites prover.rs:226-234, arithmetic.rs:121): zeros stay zero */
lh_status lh_fr_batch_invert(lh_ctx*, const lh_fr* d_in, size_t n, lh_fr* d_out);

/* ---------------------------------------------------------------- a3/a4: MultilinearPolynomial
 * poly/multilinear.rs.  Tables are 2^num_vars device lh_fr, index bit i <-> variable i. */
/* fix_var (multilinear.rs:179-183,599-618): d_out[b] = e[2b] + (e[2b+1]-e[2b])*x, n_in = 2^m */
lh_status lh_fix_var(lh_ctx*, const lh_fr* d_in, size_t n_in, const lh_fr* x, lh_fr* d_out);
/* eq_xy (multilinear.rs:91-127): d_out has 2^num_vars entries */
lh_status lh_eq_xy(lh_ctx*, const lh_fr* y, size_t num_vars, lh_fr* d_out);
/* evaluate (multilinear.rs:137-156) of `num_polys` tables at one point */
lh_status lh_evaluate(lh_ctx*, const lh_fr* const* d_polys, size_t num_polys, size_t num_vars,
                      const lh_fr* point, lh_fr* out_evals);
/* AddAssign<(&F,&Self)> folded (multilinear.rs:294-320): d_out = sum_i w[i] * d_polys[i] */
lh_status lh_lincomb(lh_ctx*, const lh_fr* const* d_polys, const lh_fr* w, size_t num_polys,
                     size_t n, lh_fr* d_out);

/* ---------------------------------------------------------------- a5-a8: piop::sum_check
 * ClassicSumCheck::prove (piop/sum_check/classic.rs:208-240) for expressions of the form
 *     [eq_xy(ys[eq_y]) *]  sum_m coeff[m] * prod_k table[factor[m][k]]
 * which covers fractional_sum_check.rs:272-281, pcs/multilinear.rs:182-190 and the Surge /
 * grand-product expressions.  Table ids < num_polys name d_polys, ids >= num_polys name
 * eq_xy(ys[id - num_polys]).  `degree` is Expression::degree() (expression.rs:171-182). */
#define LH_SC_MAX_TERMS 48
#define LH_SC_MAX_FACTORS 4
typedef struct lh_sop {
  uint32_t num_terms;
  int32_t global_eq; /* index into ys multiplied onto the whole sum, or -1 */
  lh_fr coeff[LH_SC_MAX_TERMS];
  uint8_t num_factors[LH_SC_MAX_TERMS];
  uint8_t factor[LH_SC_MAX_TERMS][LH_SC_MAX_FACTORS];
} lh_sop;

#define LH_SC_EVALUATIONS 0  /* EvaluationsProver  classic/eval.rs:68-131: d+1 evals per round */
#define LH_SC_COEFFICIENTS 1 /* CoefficientsProver classic/coeff.rs:62-150: 3 coeffs, degree 2 */
/* Returns challenges x (num_vars) and evals = each poly at x (classic.rs:143-149). */
lh_status lh_sumcheck_prove(lh_ctx*, int prover_kind, size_t num_vars, const lh_sop* expr,
                            const lh_fr* const* d_polys, size_t num_polys,
                            const lh_fr* ys, size_t num_ys, /* num_ys points of num_vars each */
                            const lh_fr* sum, lh_transcript* t,
                            lh_fr* out_challenges, lh_fr* out_evals);

/* General Expression (util/expression.rs:67-78), flattened: every node refers to EARLIER nodes, the
 * root is the last one.  DistributePowers(exprs, base) is lowered by the caller exactly as
 * Expression::evaluate does (expression.rs:155-167): e0 + base*e1 + base^2*e2 + ...            */
#define LH_EX_CONSTANT 0   /* scalar */
#define LH_EX_IDENTITY 1   /* CommonPolynomial::Identity */
#define LH_EX_LAGRANGE 2   /* CommonPolynomial::Lagrange(a) */
#define LH_EX_EQ_XY 3      /* CommonPolynomial::EqXY(a) */
#define LH_EX_POLYNOMIAL 4 /* Query { poly: a, rotation: b } */
#define LH_EX_CHALLENGE 5  /* Challenge(a) */
#define LH_EX_NEGATED 6    /* -node[a] */
#define LH_EX_SUM 7        /* node[a] + node[b] */
#define LH_EX_PRODUCT 8    /* node[a] * node[b] */
#define LH_EX_SCALED 9     /* node[a] * scalar */
typedef struct lh_expr_node { uint32_t op; int32_t a, b; uint32_t reserved; lh_fr scalar; } lh_expr_node;
typedef struct lh_expr { const lh_expr_node* nodes; size_t num_nodes; } lh_expr;
/* ClassicSumCheck::<EvaluationsProver>::prove over a VirtualPolynomial{expression, polys, challenges, ys}
 * (piop/sum_check.rs:16-37, classic.rs:208-240) including rotations (BooleanHypercube, util/arithmetic/bh.rs),
 * Identity and Lagrange.  out_evals: every poly at x (classic.rs:143-149). */
lh_status lh_sumcheck_prove_expr(lh_ctx*, size_t num_vars, const lh_expr* expr, const lh_fr* const* d_polys,
                                 size_t num_polys, const lh_fr* challenges, size_t num_challenges,
                                 const lh_fr* ys, size_t num_ys, const lh_fr* sum, lh_transcript* t,
                                 lh_fr* out_challenges, lh_fr* out_evals);

/* ---------------------------------------------------------------- a9: piop::gkr
 * prove_fractional_sum_check (piop/gkr/fractional_sum_check.rs:89-190).  claimed_*[b] may be
 * NULL (None => root written) or point to a claim (Some => root only hashed).  Outputs
 * p_xs[B], q_xs[B], x[num_vars]. */
lh_status lh_gkr_fractional_prove(lh_ctx*, size_t num_batching, size_t num_vars,
                                  const lh_fr* const* claimed_p_0s, const lh_fr* const* claimed_q_0s,
                                  const lh_fr* const* d_ps, const lh_fr* const* d_qs,
                                  lh_transcript* t, lh_fr* out_p_xs, lh_fr* out_q_xs, lh_fr* out_x);
/* Product-only layered circuit used by the Lasso memory check (no reference code; layering
 * and schedule follow fractional_sum_check.rs:62-190 with p dropped).  Trees may have different
 * depth (num_vars[b] >= 1).  out_points holds, per tree, num_vars[b] elements back to back. */
lh_status lh_grand_product_prove(lh_ctx*, size_t num_trees, const lh_fr* const* d_leaves,
                                 const size_t* num_vars, lh_transcript* t, lh_fr* out_roots,
                                 lh_fr* out_claims, lh_fr* out_points);

/* ---------------------------------------------------------------- a10: util::arithmetic::msm
 * variable_base_msm (util/arithmetic/msm.rs:84-181); result normalised to affine. */
lh_status lh_msm(lh_ctx*, const lh_fr* d_scalars, const lh_g1* d_bases, size_t n, lh_g1* out);
/* same with u32 scalars (Lasso's dim / read_ts / final_cts / E polys are small-valued) */
lh_status lh_msm_u32(lh_ctx*, const uint32_t* d_scalars, const lh_g1* d_bases, size_t n, lh_g1* out);

/* ---------------------------------------------------------------- a11/a12: pcs::multilinear::kzg */
typedef struct lh_srs lh_srs; /* MultilinearKzgProverParams (kzg.rs:56-77): eqs[0..=num_vars] on device */
/* setup (kzg.rs:166-228) with the trapdoor given explicitly: eqs[k][b] = eq_k(b; s) * G */
lh_status lh_mkzg_setup(lh_ctx*, const lh_fr* ss, size_t num_vars, lh_srs** out);
/* upload of an existing param: eqs flattened, level k at offset 2^k - 1 (trim, kzg.rs:230-250) */
lh_status lh_srs_upload(lh_ctx*, const lh_g1* eqs_flat, size_t num_vars, lh_srs** out);
lh_status lh_srs_download(lh_ctx*, const lh_srs*, lh_g1* eqs_flat);
size_t lh_srs_num_vars(const lh_srs*);
void lh_srs_free(lh_ctx*, lh_srs*);
/* commit / batch_commit (kzg.rs:252-274) */
lh_status lh_mkzg_commit(lh_ctx*, const lh_srs*, const lh_fr* d_poly, size_t num_vars, lh_g1* out);
lh_status lh_mkzg_batch_commit(lh_ctx*, const lh_srs*, const lh_fr* const* d_polys, size_t num_polys,
                               size_t num_vars, lh_g1* out_comms);
/* open (kzg.rs:276-302): writes n quotient commitments to the transcript; *out_eval = remainder */
lh_status lh_mkzg_open(lh_ctx*, const lh_srs*, const lh_fr* d_poly, size_t num_vars,
                       const lh_fr* point, lh_transcript* t, lh_fr* out_eval);
/* batch_open (pcs/multilinear.rs:134-235); Evaluation = {poly, point, value} (pcs.rs:132-155) */
typedef struct lh_evaluation { uint32_t poly, point; lh_fr value; } lh_evaluation;
lh_status lh_mkzg_batch_open(lh_ctx*, const lh_srs*, size_t num_vars,
                             const lh_fr* const* d_polys, size_t num_polys,
                             const lh_fr* points, size_t num_points,
                             const lh_evaluation* evals, size_t num_evals, lh_transcript* t);

/* ---------------------------------------------------------------- a': Lasso lookup argument
 * No reference code (README.md:1-9 only); protocol specified in oracle/pyref/lasso.py. */
#define LH_SUBTABLE_IDENTITY 0
#define LH_SUBTABLE_AND 1
#define LH_SUBTABLE_XOR 2
#define LH_LASSO_MAX_CHUNKS 8
#define LH_LASSO_MAX_MEMORIES 16
#define LH_LASSO_MAX_TERMS 16
typedef struct lh_lasso_table {
  uint32_t num_chunks;   /* c */
  uint32_t chunk_bits;   /* l : subtable size 2^l */
  uint32_t num_memories; /* alpha */
  uint32_t memory_chunk[LH_LASSO_MAX_MEMORIES];    /* j(i) */
  uint32_t memory_subtable[LH_LASSO_MAX_MEMORIES]; /* LH_SUBTABLE_* */
  uint32_t num_terms; /* g = sum_m coeff[m] * prod_{k < num_factors[m]} E_{factor[m][k]} */
  lh_fr g_coeff[LH_LASSO_MAX_TERMS];
  uint8_t g_num_factors[LH_LASSO_MAX_TERMS];
  uint8_t g_factor[LH_LASSO_MAX_TERMS][LH_SC_MAX_FACTORS];
} lh_lasso_table;
/* d_dims[j]: device u32[2^num_vars] chunk indices (< 2^chunk_bits) of every lookup. */
lh_status lh_lasso_prove(lh_ctx*, const lh_srs*, const lh_lasso_table*, size_t num_vars,
                         const uint32_t* const* d_dims, lh_transcript* t);

/* per-phase wall-clock of the last lh_lasso_prove on this ctx, milliseconds:
 * [witness, commit, surge, leaves+trees, gkr, evals, open_n, open_l, total] */
#define LH_LASSO_NUM_PHASES 9
lh_status lh_lasso_last_timing(lh_ctx*, double* out_ms);

/* ---------------------------------------------------------------- route options
 * The switches that decide WHICH code proves (never which bytes: every route yields the reference's transcript).  They
 * are per ctx; at lh_ctx_create each takes the value of the environment variable LH_<NAME IN UPPER CASE> when that is
 * set (LH_OPEN_SMALL_MIN_VARS=64 ...), else its default.  lh_ctx_set_option returns LH_ERR_ARG for an unknown name.
 *   open_small_min_vars  21   smallest batch opening (variables) whose largest quotient(s) are committed column by column
 *                             from 32-bit differences of the small-valued Lasso columns instead of from full-size
 *                             scalars (mkzg_open's column route; 64: never; below it an opening of at most four full
 *                             columns of >= 17 variables still takes it unless the option was set explicitly)
 *   open_small_depth     0    1 / 2: exactly that many column-wise quotient levels (0: chosen by cost)
 *   sc_eq_factoring      1    0: every sum-check round streams and binds its eq tables (the reference's shape) instead
 *                             of the factored form eq(y, x) = S_j eq(y_j, X) E_j[b]
 *   lasso_pack_ts        1    0: one MSM pass per read_ts column instead of packed pairs
 *   sc_tail              1    0: one kernel launch per sum-check round all the way down (no resident tail kernel)
 *   sc_tail_max_len      8192 longest table (entries) that enters the resident tail
 *   shard_exchange_log   19   sharded proofs: a sum-check goes on replicated once its residual tables hold <= 2^this
 *                             entries together (one all-gather), at the latest when the shard bits reach bit 0
 *   open_precommit       1    Lasso proofs of >= 2^this lookups (0: never, 1: always) run the column-wise quotient
 *                             commitments of their opening - MSMs over differences of witness columns, challenge-free - on
 *                             a helper ctx (own stream and host thread) beside the memory-checking sum-checks; the opening
 *                             then only combines their results
 *   msm_window_tables    0    SRS levels of <= 2^this points get a window table on first use (2^(c w) multiples of every
 *                             base, W-fold the level's memory): the W windows of a full-width column then fill ONE
 *                             bucket set - one bucket reduction, no doublings.  Measured neutral at 2^24 lookups (shorter
 *                             reduction, longer bucket runs): off by default
 *   msm_half_batches     1    an MSM batch of >= 2^24 (point, window) entries runs as two halves: the latency-bound tails of
 *                             the first half (continuation levels, bucket reduction, window sums) run on a second stream of
 *                             the ctx beside the second half's bucket accumulation; the jobs with the most entries per
 *                             bucket go last (csrc/msm.hip msm_pick_split).  0: one batch on one stream
 *   gkr_resident         1    the layers near the roots of a grand-product argument (tables of <= 2^14 entries, <= 16 trees)
 *                             run in ONE resident launch - layer loop, eq tables, rounds and final evaluations inside the
 *                             kernel (kernels_gkr.hip); 0: one sum-check per layer (eq kernels, launched rounds, a resident
 *                             tail each).  Needs sc_tail and sc_eq_factoring
 *   sc_pp_fold           1    sum-checks of the shape eq * sum_m c_m l_m r_m (the generic layers of the grand products):
 *                             the streaming rounds share one Montgomery reduction among four products and the first
 *                             binding round folds c_m into l_m (sc_round_pp_kernel); the leaf layers of the lookup-sized
 *                             trees, sum_p cs_p (l_p + k_p)(r_p + k_p), store cs_p (l_p + k_p) and r_p + k_p at their first
 *                             bind and go on as such rounds; 0: sc_round_e2_kernel / sc_round_rw_kernel in every round
 *                             2: the product-pair kernel in every streaming round of the generic layers WITHOUT the fold
 *                             (the coefficients are applied to the lane's value, the tables stay as they are; a test shape)
 *   comm_round           0    how the partial sums of a sharded sum-check round are combined over the ranks:
 *                             0 = ncclAllGather of every rank's sums + a one-thread sum-and-publish kernel;
 *                             1 = ONE collective: the round kernel leaves its sums as u64 lanes (32-bit limb | tag << 40),
 *                                 ncclAllReduce(ncclSum, ncclUint64) adds them straight into the pinned memory the host
 *                                 polls (a lane whose upper bits read R * tag is finished), the host reduces mod r;
 *                             2 = the same all-reduce into device memory, then a copy to the host.
 *                             Same proof bytes; which is faster has never been measured on more than one GPU
 * Values outside an option's range are refused (LH_ERR_ARG).
 * lh_lasso_last_route reports which of these routes the last Lasso prove on the ctx actually took, so that a byte
 * mismatch in the field can be bisected from the outside. */
lh_status lh_ctx_set_option(lh_ctx*, const char* name, int64_t value);
lh_status lh_ctx_get_option(lh_ctx*, const char* name, int64_t* out);
#define LH_LASSO_ROUTE_WORDS 24
typedef struct lh_lasso_route {
  uint32_t open_small_depth;    /* column-wise quotient levels of the opening (0: plain route) */
  uint32_t open_small_passes;   /* column passes (MSM jobs) of those levels */
  uint32_t eq_factored_rounds;  /* launched sum-check rounds with factored eq tables */
  uint32_t standard_rounds;     /* launched rounds on the standard path (eq tables streamed and bound) */
  uint32_t rw_leaf_rounds;      /* of the factored rounds: the read/write leaf-layer kernel (sc_round_rw) */
  uint32_t resident_tails;      /* resident tail launches */
  uint32_t resident_rounds;     /* rounds that ran inside them */
  uint32_t packed_ts_pairs;     /* read_ts columns committed two to a pass */
  uint32_t derived_commitments; /* E columns committed from their dim column's buckets */
  uint32_t sorted_dim_reuse;    /* dim columns whose MSM entry stream came from the access counters' sort */
  uint32_t sharded_rounds;      /* rounds that carried a collective (sharded proofs) */
  uint32_t shard_exchanges;     /* residual-table / tree-level / remainder exchanges (sharded proofs) */
  uint32_t window_table_jobs;   /* MSM jobs that ran over a window table (msm_window_tables) */
  uint32_t open_precommit;      /* 1: the opening's column-wise commitments were taken from the helper ctx (open_precommit) */
  uint32_t resident_layers;     /* grand-product layers that ran inside the resident multi-layer kernel (gkr_resident) */
  uint32_t pp_folds;            /* sum-checks whose batching coefficients were folded into the left factors (sc_pp_fold) */
  uint32_t msm_half_batches;    /* MSM batches that ran as two pipelined halves (msm_half_batches; the slot was msm29_batches,
                                   always 0 since round 5) */
  uint32_t reserved[7];
} lh_lasso_route;
lh_status lh_lasso_last_route(lh_ctx*, lh_lasso_route* out);

/* ---------------------------------------------------------------- e: one proof sharded over 2^rho GPUs
 * (the reference is single-process, there is nothing to cite.)  Every table of 2^m
 * entries is split on `rho` index bits [shard_bit, shard_bit + rho): the rank whose id equals those
 * bits holds the 2^(m-rho) entries (hi || lo).  Sum-check pairs (bit 0) and GKR / quotient halves (top
 * bit) stay local; per round each rank contributes D partial sums, per MSM one partial point, and when
 * the shard bits reach bit 0 the residual tables (2^(m-shard_bit) entries) are exchanged once.
 *
 * Transport.  On a multi-GPU node the communicator is RCCL over xGMI (lh_ctx_set_comm_rccl): one process per
 * GPU, one ncclComm per ctx, every device-side exchange is an ncclAllGather enqueued on the ctx's stream (a
 * sharded sum-check round is [round kernel -> all-gather of the D partial sums -> sum-and-publish kernel],
 * no host round trip in between).  The 128-byte unique id comes from lh_rccl_unique_id on one rank and reaches
 * the others over the caller's control plane.  lh_ctx_set_comm takes caller-supplied collectives instead
 * (tests: gloo through torch.distributed, several ranks on one GPU); device gathers are then staged through
 * the host unless all_gather_device is given. */
typedef struct lh_comm {
  int rank, size; /* size = 2^rho */
  void* user;
  /* host buffers: recv holds size * bytes_per_rank bytes, rank-major; returns 0 or a negative lh_status */
  int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank);
  /* optional (may be NULL): the same over DEVICE buffers, enqueued on hip_stream (the ctx's stream) */
  int (*all_gather_device)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);
} lh_comm;
/* comm == NULL detaches.  shard_bit + rho >= chunk_bits is required by lh_lasso_prove_sharded. */
lh_status lh_ctx_set_comm(lh_ctx*, const lh_comm* comm, size_t shard_bit);
#define LH_RCCL_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: call on ONE rank, distribute the bytes to every rank */
lh_status lh_rccl_unique_id(uint8_t out[LH_RCCL_UNIQUE_ID_BYTES]);
/* ncclCommInitRank on the ctx's device (collective: every rank of the job calls it with the same id) */
lh_status lh_ctx_set_comm_rccl(lh_ctx*, int rank, int size, const uint8_t unique_id[LH_RCCL_UNIQUE_ID_BYTES],
                               size_t shard_bit);
/* MEASUREMENT AID, not a transport: every peer is a copy of this rank (device gathers are device copies, nothing leaves
 * the GPU).  Rank `rank` of a `size`-rank sharded proof then runs alone with exactly the kernels, sizes and exchange
 * volumes it has in the real job - the compute half of a scaling curve on one GPU (tools/sharded_rank_profile.py).  The
 * transcript lh_lasso_prove_sharded produces over it is NOT a valid proof (the sums are R times this rank's share). */
lh_status lh_ctx_set_comm_loopback(lh_ctx*, int rank, int size, size_t shard_bit);
/* collectives issued on this ctx since the communicator was attached: out[0] device-side (RCCL or
 * all_gather_device), out[1] host callback.  A job on RCCL shows out[1] == 0. */
lh_status lh_ctx_comm_stats(lh_ctx*, uint64_t out[2]);
/* the same by phase of the Lasso prove in progress when the collective was issued: out[2 p] collectives, out[2 p + 1]
 * bytes this rank contributed, p = 0..6 for witness (the access counters' exchange), commit, surge, leaves, gkr, evals,
 * open; p = 7: outside a Lasso prove.  reset != 0 clears the counters after reading.  (bench.py prints them next to
 * the N > 1 line so that a measured scaling curve can be read against the per-rank compute profile.) */
lh_status lh_ctx_comm_phase_stats(lh_ctx*, uint64_t out[16], int reset);
/* device memory of a ctx (diagnostics; bench.py prints it per rank on the N > 1 line): out[0] = high-water mark of the
 * ctx's workspace arena in bytes (the prover's temporaries; its helper ctx's arena included), out[1] = bytes the arena
 * holds from the device now, out[2] / out[3] = free / total bytes of the device as the runtime reports them (all processes) */
lh_status lh_ctx_memory_stats(lh_ctx*, uint64_t out[4]);
/* the CPUs next to the ctx's device: its PCI address ("0000:72:00.0") into bus_id and the kernel's list of the CPUs on its
 * NUMA node ("0-63,128-191", sysfs local_cpulist) into cpulist - empty where the system does not say.  A deployment binds
 * the thread that proves to them (one process per GPU, bound to the GPU's node): the small proofs are a few hundred PCIe
 * round trips between that thread and the device, each one longer across the socket interconnect (tools/numa_ab.sh on a
 * quiet host: 2^20 lookups 8.09 against 8.26 ms).  The library never changes a caller's affinity itself. */
lh_status lh_ctx_host_cpus(lh_ctx*, char* bus_id, size_t bus_id_cap, char* cpulist, size_t cpulist_cap);
/* Same proof bytes as lh_lasso_prove on one GPU - it IS lh_lasso_prove with every table a shard: the same kernels (eq-
 * factored rounds, leaf-layer kernel, derived / packed commitments, column-wise top quotient) run on the shards, with a
 * collective where a round's partial sums or an MSM's partial commitments are added.  d_dims_local[j]: THIS RANK'S shard
 * of chunk column j, device u32[2^(num_vars - rho)] in the shard layout (local index hi || lo <-> lookup (hi, rank, lo)).
 * The access counters (a rank in the global lookup order per address) repartition the lookups by address owner with
 * one personalised exchange per column and direction; no rank ever holds a whole column.  Multilinear KZG only. */
lh_status lh_lasso_prove_sharded(lh_ctx*, const lh_srs*, const lh_lasso_table*, size_t num_vars,
                                 const uint32_t* const* d_dims_local, lh_transcript* t);

/* ---------------------------------------------------------------- f1: HyperPlonk with LogUp lookups
 * HyperPlonk::prove (backend/hyperplonk.rs:164-291), single-phase form (lh_hyperplonk_prove_phases below runs the
 * phase loop with a synthesize callback): instance hashing and
 * instance polys (hyperplonk.rs:170-177,365-369), witness commit, lookup_compressed_polys /
 * lookup_m_polys / lookup_h_polys / permutation_z_polys (backend/hyperplonk/prover.rs:50-313), the
 * zero-check sum-check over the composed expression (prover.rs:330-386, preprocessor.rs:25-60), the
 * evaluations in pcs_query order (verifier.rs:147-180) and MultilinearKzg::batch_open.
 * Poly numbering inside expressions is the reference's: instance | preprocess | witness |
 * permutation | lookup m | lookup h | permutation z.  Challenge numbering: circuit challenges, then
 * beta, gamma, alpha. */
typedef struct lh_hp_lookup {
  const lh_expr* inputs; /* [width] */
  const lh_expr* tables; /* [width] */
  size_t width;
} lh_hp_lookup;
/* A lookup proven by the Lasso argument INSIDE HyperPlonk::prove, in place of LogUp's m / h polys and constraint
 * (hyperplonk.rs:211-252, preprocessor.rs:79-109) - north_star's "hyperplonk::prover Lasso/Surge memory-check".  The
 * reference snapshot has no Lasso code; the schedule is specified in oracle/pyref/hyperplonk.py (LassoLookup).
 * On every row k the circuit poly `output_poly` holds a[k] = g(T_1[dim_1[k]], ..) and `chunk_polys[j]` the chunk
 * index dim_j[k] < 2^chunk_bits (ordinary circuit polys, normally witness columns tied to the circuit by its own gates;
 * not instance polys).  The argument adds committed polys read_ts_j | E_i | final_cts_j, numbered after the
 * permutation z polys; they are committed in round n after the LogUp m commitments (identity-mask framing), the
 * argument runs after the zero-check, and ONE batch_open serves the zero-check's queries and every Lasso claim.
 * Requires chunk_bits <= num_vars; output_poly and chunk_polys must be preprocess or witness polys (index >=
 * num_instance_polys and < num_instance_polys + num_preprocess_polys + witness polys - prover and verifier check the same
 * range); all Lasso commitments of one proof share one identity mask of 63 bits, so the lookups of a circuit may commit
 * to at most LH_HP_LASSO_MAX_COMMITMENTS polys together (sum over the lookups of 2 * num_chunks + num_memories). */
#define LH_HP_LASSO_MAX_COMMITMENTS 63
typedef struct lh_hp_lasso_lookup {
  lh_lasso_table table;
  size_t output_poly;
  size_t chunk_polys[LH_LASSO_MAX_CHUNKS];
} lh_hp_lasso_lookup;
typedef struct lh_hp_param { /* HyperPlonkProverParam (hyperplonk.rs:38-55), device-resident */
  size_t num_vars;
  size_t num_instance_polys;
  const size_t* num_instances;             /* [num_instance_polys] */
  size_t num_preprocess_polys;
  const void* const* d_preprocess_polys;   /* device lh_fr[2^num_vars] each */
  size_t num_witness_polys;                /* of the single phase */
  size_t num_challenges;                   /* squeezed after the witness commitments */
  size_t num_lookups;
  const lh_hp_lookup* lookups;
  size_t num_permutation_polys;
  const size_t* permutation_poly_index;    /* sorted poly indices under the permutation argument */
  const void* const* d_permutation_polys;  /* device lh_fr[2^num_vars] each (preprocessor.rs:172-203) */
  size_t num_permutation_z_polys;
  lh_expr expression;                      /* compose() output (preprocessor.rs:25-60) */
  size_t num_lasso_lookups;                /* lookups proven by Lasso instead of LogUp (0: none) */
  const lh_hp_lasso_lookup* lasso_lookups;
} lh_hp_param;
/* instances[i]: host, num_instances[i] elements.  d_witness_polys: device tables of the phase.
 * LH_ERR_INVALID_SNARK "Invalid lookup input" if an input row is not in its table (prover.rs:176), or if a Lasso
 * lookup's chunk column holds a value >= 2^chunk_bits or its output column is not the table's value. */
lh_status lh_hyperplonk_prove(lh_ctx*, const lh_srs*, const lh_hp_param*, const lh_fr* const* instances,
                              const lh_fr* const* d_witness_polys, lh_transcript* t);

/* ONE HyperPlonk proof over the 2^rho ranks of the ctx's communicator (SURVEY.md 8e / BASELINE.json configs[4]: the
 * Keccak-f circuit on 8 GPUs): HyperPlonk::prove (backend/hyperplonk.rs:164-291) with every table a shard.  Same proof
 * bytes on every rank as lh_hyperplonk_prove on one GPU.  `pp`: num_vars is the circuit's; d_preprocess_polys,
 * d_permutation_polys and d_witness_polys are THIS RANK'S shards - device lh_fr[2^(num_vars - rho)] in the shard layout
 * of lh_lasso_prove_sharded (local index hi || lo <-> row (hi, rank, lo)); instances are the full lists on every rank.
 * What crosses ranks: partial commitments (one exchange per commit round), the zero-check's partial sums per round and
 * its residual tables once (piop/sum_check/classic.rs:90-141 over shards), the rows of polys queried at a rotation
 * (gathered once in round 0, classic.rs:104-126), the per-row products of the permutation argument (gathered once; the
 * prefix product in hypercube order, backend/hyperplonk/prover.rs:308-323, runs on every rank), the Lasso lookups'
 * exchanges and the shared batch opening's as in lh_lasso_prove_sharded.  Circuits with LogUp lookups (a global
 * sort-merge join, prover.rs:139-200) are refused with LH_ERR_ARG: they run as replicas.  Needs shard_bit >= 1,
 * shard_bit + rho >= the Lasso lookups' chunk_bits and num_vars > shard_bit + rho; multilinear KZG. */
/* this rank's shard of a full device table: local[hi || lo] = global[(hi, rank, lo)], n_local = n / 2^rho entries of
 * elem_bytes (4, 32 or 64) each - how a caller that holds whole polys makes the inputs of the sharded proves */
lh_status lh_shard_extract(lh_ctx*, const void* d_global, size_t n_local, size_t shard_bit, size_t rho, size_t rank,
                           size_t elem_bytes, void* d_local);
lh_status lh_hyperplonk_prove_sharded(lh_ctx*, const lh_srs*, const lh_hp_param*, const lh_fr* const* instances,
                                      const lh_fr* const* d_witness_polys_local, lh_transcript* t);

/* Multi-phase circuits: the phase loop of HyperPlonk::prove (backend/hyperplonk.rs:185-205).  Phase r calls
 * PlonkishCircuit::synthesize(r, challenges so far) (backend.rs:139) for num_witness_polys[r] polys, commits them and
 * squeezes num_challenges[r] challenges; Challenge(i) in expressions indexes the concatenation, then beta, gamma, alpha.
 * The param's num_witness_polys / num_challenges hold the totals over the phases. */
typedef struct lh_hp_circuit { /* the witness half of `&impl PlonkishCircuit<F>` */
  void* user;
  /* fills d_out_polys[0 .. num_out) with DEVICE pointers to 2^num_vars lh_fr tables that stay valid until the prove
   * returns; `challenges` are those of the earlier phases; returns 0 or a negative lh_status */
  int (*synthesize)(void* user, size_t round, const lh_fr* challenges, size_t num_challenges,
                    const void** d_out_polys, size_t num_out);
} lh_hp_circuit;
lh_status lh_hyperplonk_prove_phases(lh_ctx*, const lh_srs*, const lh_hp_param*, size_t num_phases,
                                     const size_t* num_witness_polys, const size_t* num_challenges,
                                     const lh_fr* const* instances, const lh_hp_circuit* circuit, lh_transcript* t);

/* ---------------------------------------------------------------- f1: verifiers (host only, no GPU, no lh_ctx)
 * The verify half of the trait surface: PolynomialCommitmentScheme::{verify, batch_verify}
 * (pcs/multilinear/kzg.rs:330-375, pcs/multilinear.rs:237-276), SumCheck::verify (classic.rs:242-272),
 * PlonkishBackend::verify (backend/hyperplonk.rs:293-362, hyperplonk/verifier.rs:39-182), and the verifier
 * of the Lasso argument specified in oracle/pyref/lasso.py.  Pairings: BN254 optimal ate
 * (MultiMillerLoop::pairings_product_is_identity, util/arithmetic.rs:24-33). */
typedef struct lh_g2 { uint64_t x_c0[4], x_c1[4], y_c0[4], y_c1[4]; } lh_g2; /* bn256::G2Affine, Montgomery; identity = 0 */
typedef struct lh_mkzg_vp lh_mkzg_vp; /* MultilinearKzgVerifierParams (kzg.rs:79-101): g1, g2, ss[i] = s_i * g2 */
/* the verifier half of lh_mkzg_setup: same trapdoor, generators (1,2) and the bn256 G2 generator */
lh_status lh_mkzg_vp_setup(const lh_fr* ss, size_t num_vars, lh_mkzg_vp** out);
lh_status lh_mkzg_vp_new(const lh_g1* g1, const lh_g2* g2, const lh_g2* ss, size_t num_vars, lh_mkzg_vp** out);
lh_status lh_mkzg_vp_export(const lh_mkzg_vp*, lh_g1* g1, lh_g2* g2, lh_g2* ss);
size_t lh_mkzg_vp_num_vars(const lh_mkzg_vp*);
void lh_mkzg_vp_free(lh_mkzg_vp*);
/* e(p_i, q_i) product == 1 */
lh_status lh_pairing_check(const lh_g1* ps, const lh_g2* qs, size_t n, int* out_is_identity);
/* LH_ERR_INVALID_PCS_OPEN "Invalid multilinear KZG open" on a failed check */
lh_status lh_mkzg_verify(const lh_mkzg_vp*, const lh_g1* comm, const lh_fr* point, size_t num_vars,
                         const lh_fr* eval, lh_transcript* t);
lh_status lh_mkzg_batch_verify(const lh_mkzg_vp*, size_t num_vars, const lh_g1* comms, size_t num_comms,
                               const lh_fr* points, size_t num_points, const lh_evaluation* evals,
                               size_t num_evals, lh_transcript* t);
/* prover_kind selects the round-message type (Evaluations / Coefficients).  Outputs the final claim and
 * the challenges x[num_vars]; LH_ERR_INVALID_SUMCHECK on an inconsistent round. */
lh_status lh_sumcheck_verify(int prover_kind, size_t num_vars, size_t degree, const lh_fr* sum,
                             lh_transcript* t, lh_fr* out_eval, lh_fr* out_x);
lh_status lh_lasso_verify(const lh_mkzg_vp*, const lh_lasso_table*, size_t num_vars, lh_transcript* t);
typedef struct lh_hp_vparam { /* HyperPlonkVerifierParam (hyperplonk.rs:57-74) */
  size_t num_vars;
  size_t num_instance_polys;
  const size_t* num_instances;
  size_t num_witness_polys;
  size_t num_challenges;
  size_t num_lookups;
  size_t num_permutation_z_polys;
  lh_expr expression;
  size_t num_preprocess_polys;
  const lh_g1* preprocess_comms;
  size_t num_permutation_polys;
  const lh_g1* permutation_comms;
  size_t num_lasso_lookups;
  const lh_hp_lasso_lookup* lasso_lookups;
} lh_hp_vparam;
lh_status lh_hyperplonk_verify(const lh_mkzg_vp*, const lh_hp_vparam*, const lh_fr* const* instances,
                               lh_transcript* t);
/* multi-phase (hyperplonk.rs:309-316): per phase num_witness_polys[r] commitments are read, num_challenges[r] squeezed */
lh_status lh_hyperplonk_verify_phases(const lh_mkzg_vp*, const lh_hp_vparam*, size_t num_phases,
                                      const size_t* num_witness_polys, const size_t* num_challenges,
                                      const lh_fr* const* instances, lh_transcript* t);

/* ---------------------------------------------------------------- f3: Zeromorph over univariate KZG
 * PolynomialCommitmentScheme for Zeromorph<UnivariateKzg<Bn256>> (pcs/multilinear/zeromorph.rs:67-256 on top of
 * pcs/univariate/kzg.rs:23-36,161-378): a multilinear table of 2^n evaluations is committed as the coefficient
 * vector of a univariate polynomial against powers_of_s_g1 (2^n points instead of multilinear KZG's 2^(n+1)-1).
 * `poly_size` is the trim size (zeromorph.rs:90-108): commitments use powers[..poly_size], the final quotient of
 * an opening powers[size - poly_size..]. */
typedef struct lh_usrs lh_usrs; /* UnivariateKzgParam: powers_of_s_g1 on the device */
lh_status lh_ukzg_setup(lh_ctx*, const lh_fr* s, size_t poly_size, lh_usrs** out); /* kzg.rs:175-218, trapdoor explicit */
lh_status lh_usrs_upload(lh_ctx*, const lh_g1* powers_of_s_g1, size_t poly_size, lh_usrs** out);
lh_status lh_usrs_download(lh_ctx*, const lh_usrs*, lh_g1* powers_of_s_g1);
size_t lh_usrs_size(const lh_usrs*);
void lh_usrs_free(lh_ctx*, lh_usrs*);
lh_status lh_zeromorph_batch_commit(lh_ctx*, const lh_usrs*, size_t poly_size, const lh_fr* const* d_polys,
                                    size_t num_polys, size_t num_vars, lh_g1* out_comms);
/* zeromorph.rs:134-199: writes n quotient commitments, q_hat's commitment and the KZG proof pi */
lh_status lh_zeromorph_open(lh_ctx*, const lh_usrs*, size_t poly_size, const lh_fr* d_poly, size_t num_vars,
                            const lh_fr* point, lh_transcript* t);
lh_status lh_zeromorph_batch_open(lh_ctx*, const lh_usrs*, size_t poly_size, size_t num_vars,
                                  const lh_fr* const* d_polys, size_t num_polys, const lh_fr* points,
                                  size_t num_points, const lh_evaluation* evals, size_t num_evals, lh_transcript* t);
/* verifier half (host only): ZeromorphKzgVerifierParam = (g1, g2, [s]_2, [s^offset]_2) */
typedef struct lh_zm_vp lh_zm_vp;
lh_status lh_zeromorph_vp_setup(const lh_fr* s, size_t param_size, size_t poly_size, lh_zm_vp** out);
lh_status lh_zeromorph_vp_new(const lh_g1* g1, const lh_g2* g2, const lh_g2* s_g2, const lh_g2* s_offset_g2,
                              lh_zm_vp** out);
lh_status lh_zeromorph_vp_export(const lh_zm_vp*, lh_g1* g1, lh_g2* g2, lh_g2* s_g2, lh_g2* s_offset_g2);
void lh_zeromorph_vp_free(lh_zm_vp*);
/* LH_ERR_INVALID_PCS_OPEN "Invalid Zeromorph KZG open" on a failed pairing check (zeromorph.rs:215-247) */
lh_status lh_zeromorph_verify(const lh_zm_vp*, const lh_g1* comm, const lh_fr* point, size_t num_vars,
                              const lh_fr* eval, lh_transcript* t);
lh_status lh_zeromorph_batch_verify(const lh_zm_vp*, size_t num_vars, const lh_g1* comms, size_t num_comms,
                                    const lh_fr* points, size_t num_points, const lh_evaluation* evals,
                                    size_t num_evals, lh_transcript* t);

/* HyperPlonk<Zeromorph<UnivariateKzg<Bn256>>> (the reference's second tested backend configuration,
 * backend/hyperplonk.rs:426): same schedule as lh_hyperplonk_prove / lh_hyperplonk_verify with the other PCS */
/* the Lasso argument over Zeromorph (same protocol, oracle/pyref/lasso.py with pcs = zeromorph) */
lh_status lh_lasso_prove_zeromorph(lh_ctx*, const lh_usrs*, size_t poly_size, const lh_lasso_table*, size_t num_vars,
                                   const uint32_t* const* d_dims, lh_transcript* t);
lh_status lh_lasso_verify_zeromorph(const lh_zm_vp*, const lh_lasso_table*, size_t num_vars, lh_transcript* t);
lh_status lh_hyperplonk_prove_zeromorph(lh_ctx*, const lh_usrs*, size_t poly_size, const lh_hp_param*,
                                        const lh_fr* const* instances, const lh_fr* const* d_witness_polys,
                                        lh_transcript* t);
lh_status lh_hyperplonk_verify_zeromorph(const lh_zm_vp*, const lh_hp_vparam*, const lh_fr* const* instances,
                                         lh_transcript* t);
/* multi-phase circuits over Zeromorph: the phase loop of backend/hyperplonk.rs:185-205 / 309-316 is generic over the PCS
 * (the reference instantiates HyperPlonk<Zeromorph<..>> at hyperplonk.rs:426); arguments as lh_hyperplonk_prove_phases /
 * lh_hyperplonk_verify_phases */
lh_status lh_hyperplonk_prove_phases_zeromorph(lh_ctx*, const lh_usrs*, size_t poly_size, const lh_hp_param*,
                                               size_t num_phases, const size_t* num_witness_polys,
                                               const size_t* num_challenges, const lh_fr* const* instances,
                                               const lh_hp_circuit* circuit, lh_transcript* t);
lh_status lh_hyperplonk_verify_phases_zeromorph(const lh_zm_vp*, const lh_hp_vparam*, size_t num_phases,
                                                const size_t* num_witness_polys, const size_t* num_challenges,
                                                const lh_fr* const* instances, lh_transcript* t);

/* development / tests: the HIP source the runtime compiler (csrc/jit.cpp) is given for a register program - words
 * {op | dst << 4 | a.kind << 8 | b.kind << 10, a.idx | b.idx << 16} with op ADD 0, SUB 1, MUL 2, NEG 3, MOV 4 and operand
 * kinds register 0, table 1, constant 2 (csrc/dev.hpp PROG_*).  Host code only (no GPU needed): tests/test_jit_source.py
 * evaluates the emitted statements against the program.  *len = length of the text; `out` may be null. */
lh_status lh_debug_jit_source(const uint32_t* code, size_t num_instrs, uint32_t num_regs, uint32_t result_reg, int degree,
                              char* out, size_t cap, size_t* len);

/* ---------------------------------------------------------------- measurement (bench.py)
 * Per-kernel HIP-event timing on the ctx stream.  While enabled every instrumented launch is
 * synchronised, so whole-prove wall time is NOT representative; use a separate pass. */
typedef struct lh_prof_rec {
  char name[40];
  double ms;    /* HIP-event duration of the launch */
  double bytes; /* algorithmic bytes of the launch */
  double muls;  /* field multiplications of the launch */
  double items; /* work items of the launch */
} lh_prof_rec;
/* on = 1: every instrumented launch bracketed by HIP events and SYNCHRONISED (exact per-kernel durations, a slower prove: not
 * for a timed region).  on = 2: LIVE records - an event pair around every bucket-accumulation launch (the dominant kernel), on
 * the stream it is launched on, nothing waited for: usable inside a timed region.  The two halves of a pipelined MSM batch
 * run at the same time, so a launch's `ms` is its span and a record named "msm_accumulate0/batch" carries the span of all
 * launches of one batch (first start to last end) with their bytes / muls / items added up.  Also clears the record list. */
lh_status lh_profile_enable(lh_ctx*, int on);
/* copies up to `cap` records, returns the total number recorded in *count */
lh_status lh_profile_read(lh_ctx*, lh_prof_rec* out, size_t cap, size_t* count);

#ifdef __cplusplus
}
#endif
#endif /* LASSO_HIP_H */
