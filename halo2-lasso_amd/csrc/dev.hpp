// Internal: device context, workspace arena, kernel launcher declarations.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <functional>
#include <string>
#include <vector>
#include <utility>
#include <stdexcept>

#include "../../include/lasso_hip.h"
#include "ec.cuh"

namespace lh {

// ------------------------------------------------------------------ errors
struct Error : std::exception {
  int code;
  std::string msg;
  Error(int c, std::string m) : code(c), msg(std::move(m)) {}
  const char* what() const noexcept override { return msg.c_str(); }
};

void set_last_error(const char* msg);

#define LH_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      throw ::lh::Error(LH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

#define LH_REQUIRE(cond, code, text) \
  do {                               \
    if (!(cond)) throw ::lh::Error((code), (text)); \
  } while (0)

// ------------------------------------------------------------------ workspace arena
// Stack-disciplined device allocator: the prover's temporaries nest (per sum-check, per layer),
// so a bump pointer with mark/release avoids hipMalloc/hipFree (both synchronise) on the hot path.
class Arena {
 public:
  struct Mark {
    size_t block, used;
  };
  ~Arena();
  void* alloc(size_t bytes);
  void* alloc_raw(size_t bytes);
  template <class T>
  T* alloc_n(size_t n) {
    return (T*)alloc(n * sizeof(T));
  }
  Mark mark() const { return {cur_, blocks_.empty() ? 0 : blocks_[cur_].used}; }
  void release(Mark m);
  size_t high_water() const { return high_; }
  size_t reserved() const {  // bytes held from the device
    size_t t = 0;
    for (const Block& b : blocks_) t += b.size;
    return t;
  }

 private:
  struct Block {
    char* p;
    size_t size, used;
  };
  std::vector<Block> blocks_;
  size_t cur_ = 0, high_ = 0;
};

struct ArenaScope {
  Arena& a;
  Arena::Mark m;
  explicit ArenaScope(Arena& a_) : a(a_), m(a_.mark()) {}
  ~ArenaScope() { a.release(m); }
};

// ------------------------------------------------------------------ per-kernel profiling (off by default)
// When enabled every instrumented launch is bracketed by HIP events on the ctx stream and synchronised,
// so the per-kernel durations are exact but the prove as a whole is slower: bench.py runs its timed
// region with profiling off and a separate profiled pass for the roofline numbers.
struct ProfRec {
  char name[40];
  double ms;     // HIP-event duration
  double bytes;  // algorithmic bytes of the launch (SURVEY.md §8d formulas)
  double muls;   // field multiplications of the launch (integer-ALU roofline)
  double items;  // work items (pairs, points, ...)
};

struct ScFinishArgs {
  uint32_t* ticket;      // device counter
  uint32_t last_ticket;  // value the last workgroup of this launch draws
  Fr* out_host;          // pinned
  uint32_t* flag;        // pinned
  uint32_t seq;
  // sharded rounds, all-reduce variant (Options::comm_round, comm.cpp comm_sum_publish): the sums ALSO leave as 8 u64
  // lanes each, lane = 32-bit limb | tag << SC_LANE_TAG_SHIFT - what ncclAllReduce(ncclSum, ncclUint64) adds over the ranks
  // without losing a carry; null otherwise
  uint64_t* wide;
  uint32_t tag;
  // the workgroups' partial sums on their way to the workgroup that draws the launch's last ticket, as 8-byte lanes
  // (limb | seq << 32) in a buffer that holds nothing else (Ctx::fin_lanes, resident.cuh fin_put / fin_get): each lane
  // validates itself, so the hand-off needs no fence - an agent-scope release per workgroup makes the L2 of its XCD walk its
  // dirty lines, ~27 ns per workgroup of a launch that has just stored a table (tools/ubench/u32_bind.hip: a 2^24-entry
  // bind 0.147 -> 0.108 ms).  Null: the partials travel through `partials` under release / acquire fences.
  uint64_t* lanes;
};
constexpr uint32_t FIN_LANE_SUMS = 65536;  // partial sums the lane buffer holds (a launch's workgroups x its sums per workgroup)
// a lane of the all-reduce variant: bits [0, 40) the sum of <= 2^8 32-bit limbs, bits [40, 64) the sum of the ranks' tags -
// every rank stamps the same tag < 2^24 / R, so a lane whose upper bits read R * tag is the finished sum of THIS round
// (lanes are 8-byte stores: each validates itself, no flag and no ordering between them is needed)
constexpr unsigned SC_LANE_TAG_SHIFT = 40;

// ------------------------------------------------------------------ route options and the route a proof took
// The switches that decide WHICH code proves (not how fast a kernel runs).  Per ctx: set through lh_ctx_set_option
// (include/lasso_hip.h lists them), initialised at ctx creation from the environment variable of the same name in upper
// case with an LH_ prefix (LH_OPEN_SMALL_MIN_VARS, ...).  Proof bytes never depend on them.
struct Options {
  int64_t open_small_min_vars = 21;    // smallest opening whose largest quotient(s) are committed column by column (64: never)
  bool open_small_min_vars_forced = false;  // set explicitly: no automatic "few columns" exception below the threshold
  int64_t open_small_depth = 0;        // 1 / 2: that many column-wise quotient levels whatever the shape (0: by cost)
  int64_t sc_eq_factoring = 1;         // 0: every sum-check round streams and binds its eq tables
  int64_t lasso_pack_ts = 1;           // 0: one MSM pass per read_ts column
  int64_t sc_tail = 1;                 // 0: one launch per sum-check round all the way down (no resident tail)
  int64_t sc_tail_max_len = 8192;      // longest table that enters the resident tail
  int64_t shard_exchange_log = 19;     // sharded sum-check: the residual tables travel once they hold <= 2^this entries
                                       // (17 until round 5; 19 removes two sharded rounds - two collectives - per sum-check at
                                       // the same per-rank compute time, 20 two more for +2 % of it: tools/r05_xlog_sweep.sh)
  int64_t open_precommit = 1;          // proofs of >= 2^this lookups run the challenge-free half of the opening's column route
                                       // (the MSMs over differences of witness columns) on a helper ctx beside the sum-checks of
                                       // a Lasso prove (0: never; 1: always)
  int64_t msm_window_tables = 0;       // SRS levels of <= 2^this points get a window table (MsmJob::win_table) on first use:
                                       // full-width columns over them reduce ONE bucket set (0: no tables)
  int64_t sc_pp_fold = 1;              // 0: the generic layers of the grand products keep their coefficients as products of
                                       // every round (sc_round_e2) instead of folding them into the left factors and sharing
                                       // Montgomery reductions (sc_round_pp); the leaf layers (ScRwRound) keep cs and k
                                       // outside their tables and run the leaf kernel in every streaming round
                                       // 2: sc_round_pp in every streaming round, nothing folded (coefficients applied on the way)
  int64_t msm_half_batches = 1;        // an MSM batch of >= 2^24 entries runs as two halves: the first half's latency-bound tails
                                       // (continuation levels, bucket reduction, window sums) on the aux stream beside the
                                       // second half's accumulation (msm.hip msm_pick_split; 0: one batch, one stream)
  int64_t gkr_resident = 1;            // the layers of a grand-product argument whose tables fit the resident kernel run in ONE
                                       // launch (layer loop, eq tables and rounds inside; 0: one sum-check per layer)
  int64_t comm_round = 0;              // how the partial sums of a sharded sum-check round are combined: 0 = all-gather +
                                       // a sum-and-publish kernel; 1 = ONE collective, ncclAllReduce(ncclSum) over u64 lanes
                                       // of 32-bit limbs straight into the pinned memory the host polls (lazy reduction mod
                                       // r on the host, no kernel behind the collective); 2 = the same all-reduce into
                                       // device memory followed by a copy to the host (should a fabric refuse host memory
                                       // as a receive buffer).  Same proof bytes; unmeasured on more than one GPU.
  Options();                           // environment defaults (dev.cpp)
  int64_t* find(const char* name);
  static bool in_range(const char* name, int64_t value);  // the range lh_ctx_set_option accepts
};
struct RouteStats {  // lh_lasso_route (include/lasso_hip.h): counters of the last Lasso prove on the ctx
  uint32_t v[LH_LASSO_ROUTE_WORDS] = {0};
  enum { OPEN_DEPTH, OPEN_PASSES, EF_ROUNDS, STD_ROUNDS, RW_ROUNDS, TAILS, TAIL_ROUNDS, PACKED_TS, DERIVED, SORTED_REUSE,
         SHARDED_ROUNDS, SHARD_EXCHANGES, WIN_TABLE_JOBS, OPEN_PRECOMMIT, RESIDENT_LAYERS, PP_FOLDS, MSM_HALF_BATCHES };
};

// ------------------------------------------------------------------ a long-lived host thread (dev.cpp)
// Runs submitted tasks one after the other: the driver of a helper ctx (open_columns.cpp open_precommit_*) - starting a
// std::thread per proof costs tens of microseconds on the prover's critical path, a condition variable costs two.
class HostWorker {
 public:
  HostWorker();
  ~HostWorker();  // finishes what was submitted, then joins
  void submit(std::function<void()> fn);
  void wait();    // until every submitted task has finished
 private:
  struct Impl;
  Impl* impl_;
};

// ------------------------------------------------------------------ context
struct Ctx {
  Options opt;
  RouteStats route;
  int device = 0;
  hipStream_t stream = nullptr;
  Arena arena;
  // pinned host staging for small D2H results (round messages, window sums)
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  int num_cus = 256;
  int wall_clock_khz = 100000;  // rate of wall_clock64() on the device
  double lasso_ms[LH_LASSO_NUM_PHASES] = {0};
  // one proof over several GPUs: host-side communicator + position of the shard bits
  lh_comm comm = {0, 1, nullptr, nullptr, nullptr};
  bool has_comm = false;
  bool comm_loopback = false;  // measurement aid: every peer is a copy of this rank (comm.cpp comm_attach_loopback)
  size_t shard_bit = 0;
  // a sharded proof is running on this ctx (set for the duration of lh_lasso_prove_sharded): every routine of the prover
  // then takes its tables as this rank's shards (struct Shard below); otherwise an attached communicator is ignored
  bool shard_active = false;
  void* rccl_comm = nullptr;      // ncclComm_t of the built-in RCCL backend (comm.cpp)
  void* comm_stage = nullptr;     // device staging of host-side gathers over a device-only communicator
  size_t comm_stage_bytes = 0;
  uint64_t comm_stats[2] = {0, 0};  // collectives issued: device-side, host callback
  // the same by phase of the Lasso prove that issued them (index = the phase in progress: witness, commit, surge, leaves,
  // gkr, evals, open; 7 = outside a Lasso prove): [phase][0] collectives, [phase][1] bytes this rank contributed
  uint64_t comm_phase_stats[8][2] = {};
  int comm_phase = 7;
  // hint for the NEXT sum_check_prove (consumed and cleared at its entry): its single table d_polys[0] has NOT been written -
  // its values are this 32-bit column.  The sum-check either runs its first three rounds from the column (sumcheck.cpp: the
  // sums of k_inner_products_small_quads are rounds 0 and 1, k_sc_round_u32_bind2 is round 2) or fills the table itself.
  struct ScU32 {
    const uint32_t* col = nullptr;
    bool have_sums = false;
    Fr odd, s2, s3;  // (when have_sums: out_host[1..3] of k_inner_products_small_quads against the sum-check's E_0)
  } sc_u32;
  // hint for the NEXT sum-check of the batch-opening shape sum_m eq(y_m, .) poly_m (consumed and cleared at its entry):
  // poly b = sum_k w[k] col[k] over 32-bit columns (entries beyond len[k] are zero) and d_polys[b] has NOT been written.  The
  // sum-check runs its first three rounds from the columns (sumcheck.cpp: k_inner_products_quads, k_lincomb_bind2) or
  // fills the tables itself (k_lincomb_mixed) and says so in `built`.
  struct ScU32Terms {
    struct Poly {
      std::vector<const uint32_t*> col;
      std::vector<size_t> len;
      std::vector<Fr> w;
    };
    std::vector<Poly> polys;  // empty: no hint
    bool built = false;       // out: the tables of d_polys hold the polys in full
  } sc_u32_terms;
  // sharded sum-check rounds: the round kernel leaves its D sums in this DEVICE buffer (and "publishes" to a device
  // word) instead of pinned host memory; the all-gather and the sum-and-publish kernel follow on the stream
  Fr* sc_redirect = nullptr;
  uint64_t* sc_wide = nullptr;  // all-reduce variant: the round kernel also leaves its sums as tagged u64 lanes here (device)
  uint32_t sc_tag = 0;          // the tag of the round in progress (ScFinishArgs::tag)
  uint32_t sc_tag_seq = 0;      // tags handed out so far
  uint64_t* lanes_host = nullptr;  // pinned: where the all-reduce leaves the lanes' sums (created on first use)
  bool comm_host_recv_ok = true;   // the attach-time probe (comm.cpp comm_probe_host_recv): RCCL delivers into pinned host memory here
  Fr* round_out(Fr* out_host) const { return sc_redirect ? sc_redirect : out_host; }
  void wait_round(uint32_t seq) {  // the host's wait for a round kernel's sums (nothing to wait for when they stay on the device)
    if (!sc_redirect) wait_flag(seq);
  }
  bool last_round_folded = false;  // k_sc_round: the launch it chose folded the coefficients into the left factors (ScRound::pp)
  uint64_t* tail_trace = nullptr;  // development: device stamps of the last resident tail (LH_SC_TAIL_TRACE)
  // development (LH_HOST_TRACE=1): host wall-clock stamps at named points of a prove, printed (deltas in us) when the prove
  // ends - where the host's share of a gap between two kernels goes
  std::vector<std::pair<const char*, double>> host_stamps;
  bool host_trace_on = false;
  void host_stamp(const char* tag);
  void host_stamps_print();
  // eq tables of point tails y[1..n) built during one proof (sumcheck.cpp eq_half_*): an evaluation, a sum-check and the batch
  // opening at the same point share one table.  Arena memory of the proof's scope: the proof clears the list (EqHalfScope).
  struct EqHalfEntry {
    std::vector<uint8_t> key;  // the bytes of y[1..n)
    const Fr* table;
    bool sharded = false;  // this rank's shard of the table (sharded proofs)
  };
  std::vector<EqHalfEntry> eq_half_cache;
  bool prof = false;
  std::vector<ProfRec> prof_recs;
  // LIVE records (lh_profile_enable(ctx, 2)): a HIP-event pair around every bucket-accumulation launch, on the stream it is
  // launched on, nothing synchronised and nothing serialised - the timed region as it runs.  The two halves of a pipelined
  // MSM batch run at the same time on two streams, so a launch's duration is its SPAN (what a kernel trace shows too) and
  // the chip's rate follows from the batch's span (first start to last end): lh_profile_read resolves both.
  bool live = false;
  struct LiveRec {
    ProfRec rec;
    hipEvent_t e0, e1;
    uint32_t batch;
  };
  std::vector<LiveRec> live_recs;
  std::vector<hipEvent_t> live_pool;  // events not in use
  uint32_t live_batch = 0;
  hipEvent_t live_event();
  void live_resolve(std::vector<ProfRec>& out);  // waits for the streams, appends per-launch and per-batch records, recycles the events
  // helper ctx (same device, own stream / arena / pinned blocks; capi.cpp ctx_helper) and what it is committing ahead of
  // the opening (open_columns.cpp open_precommit_*): both owned by this ctx
  bool is_helper = false;  // this ctx is somebody's helper: its throughput kernels leave wave slots to the owner's stream
  Ctx* helper = nullptr;
  void* helper_handle = nullptr;
  void* precommit = nullptr;
  HostWorker* worker = nullptr;   // the host thread that drives THIS ctx when it is somebody's helper (created on first use)
  hipEvent_t handoff_ev = nullptr;  // recorded on this ctx's stream where a helper's stream may start reading its columns
  // phase boundaries of the last Lasso prove as events on the stream (no host sync at a boundary: lasso.cpp lap)
  hipEvent_t phase_ev[LH_LASSO_NUM_PHASES] = {};
  bool phase_ev_pending = false;
  void phase_times_resolve();  // events -> lasso_ms (waits for the last one)
  // called once (and cleared) when a grand-product argument has built its trees and starts its layer sum-checks: the
  // latency-bound stretch of a Lasso prove, where lasso_prove starts the opening's precommit
  std::function<void()> gkr_hook;
  hipEvent_t prof_ev[2] = {nullptr, nullptr};
  void* pin(size_t bytes);  // grows the pinned buffer if needed
  // small device -> host download through a second pinned staging buffer, synchronising: an async copy into
  // pageable memory makes the runtime pin pages on the fly (hundreds of microseconds for a few KB)
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // pinned staging of the radix sort's slab descriptors (sort.hip) and the event after their last upload
  void* sort_stage[2] = {nullptr, nullptr};
  size_t sort_stage_bytes[2] = {0, 0};
  hipEvent_t sort_ev[2] = {nullptr, nullptr};
  hipEvent_t sort_stage_done(int side) {
    if (!sort_ev[side]) LH_HIP(hipEventCreateWithFlags(&sort_ev[side], hipEventDisableTiming));
    return sort_ev[side];
  }
  void d2h(void* dst, const void* d_src, size_t bytes);
  void sync() { LH_HIP(hipStreamSynchronize(stream)); }
  // Round-trip fast path: a kernel publishes its (small) result into pinned memory and then stores a
  // sequence number with system-scope release; the host spins on it instead of going through
  // hipStreamSynchronize (tens of microseconds per call, paid once per sum-check round).
  uint32_t* flag = nullptr;  // pinned, coherent
  uint32_t flag_seq = 0;
  uint32_t* ticket = nullptr;  // device counter for in-launch final reductions; only ever grows (word 8: device flag;
                               // word 10: the resident grand-product kernel's start verdict; words 32..47: the resident
                               // tail's relay chunks)
  uint32_t ticket_base = 0;    // its value before the next launch
  uint64_t* fin_lanes = nullptr;  // ScFinishArgs::lanes: 8 lanes per partial sum, FIN_LANE_SUMS of them; zeroed once, then only tagged lanes
  // a second stream of the ctx for work that runs BESIDE the ctx's stream inside one call (msm.hip: the tails of a batch's
  // first half); its kernels draw tickets from word 16 and publish to flag word 4, so the two streams never share a counter
  hipStream_t aux_stream = nullptr;
  hipEvent_t aux_ev = nullptr;
  uint32_t aux_ticket_base = 0;
  void aux_streams();  // creates them on first use
  struct ScFinishArgs finish_for_aux(uint32_t grid, uint32_t seq);
  void wait_flag_aux(uint32_t seq);
  // (`stored_bytes`: what the launch's workgroups store besides their sums - LH_FIN_LANES_MIN_BYTES, development)
  struct ScFinishArgs finish_for(uint32_t grid, Fr* out_host, uint32_t seq, double stored_bytes = 0);
  uint32_t next_seq() { return ++flag_seq; }
  // host -> resident kernel mailbox (second cache line of the flag allocation)
  struct TailMbox* mbox() { return (struct TailMbox*)((char*)flag + 64); }
  void mbox_send(const Fr& r, uint32_t seq);
  void mbox_abort();
  // the resident grand-product kernel's boxes (kernels_gkr.hip): host -> kernel layer messages (pinned) and the device
  // relay of whatever one workgroup fetched from the host; created on first use
  struct TailChunk* gkr_mbox = nullptr;
  struct TailChunk* gkr_relay = nullptr;
  void gkr_boxes();
  void gkr_send_layer(const Fr* vals, size_t count, uint32_t seq);
  void gkr_abort();   // mbox_abort + the layer box
  void gkr_resync();  // after an aborted launch: markers cleared, ticket counter re-read
  void wait_flag(uint32_t seq);
  // more than 64 KB of dynamic LDS for `fn` on this ctx's device (the attribute is per device: once per ctx and kernel)
  std::vector<const void*> lds_opted;
  void opt_in_lds(const void* fn, int bytes);
  // message of a resident tail round: `count` chunks that all carry `seq` -> `count` / 3 field elements
  void wait_chunks(const struct TailChunk* chunks, size_t count, uint32_t seq, Fr* out);
  // the same; false (nothing copied) when the first chunk carries `alt` instead
  bool wait_chunks_or(const struct TailChunk* chunks, size_t count, uint32_t seq, uint32_t alt, Fr* out);
};

// Geometry of a proof sharded over R = 2^rho ranks.  A table of 2^m entries is split on the index bits
// [j, j + rho): rank s holds the 2^(m - rho) entries (hi || lo) <-> global index (hi, s, lo).  Every kernel is
// index-agnostic over such a local table because
//   * sum-check pairs are (2b, 2b+1): bit 0, local while it is not a shard bit (rounds 0..j-1);
//   * product-tree / quotient halves are split on the top bit, local while the table has more than j + rho variables.
// `on` false (no sharded proof running): rho = 0 and nothing is sharded - the single-GPU prover is the world of one.
struct Shard {
  bool on = false;
  bool loopback = false;  // the measurement communicator (every peer is a copy of this rank)
  size_t rho = 0, j = 0, rank = 0, R = 1;
  explicit Shard(const Ctx& c) {
    if (!c.shard_active) return;
    on = true, j = c.shard_bit, rank = (size_t)c.comm.rank, R = (size_t)c.comm.size, loopback = c.comm_loopback;
    while (((size_t)1 << rho) < R) rho++;
  }
  // is a table of `num_vars` variables held in shards?  Smaller ones are replicated and worked on redundantly.  (A
  // communicator of ONE rank goes through the same code paths - its "exchange" round still needs one round before it.)
  bool sharded(size_t num_vars) const { return on && num_vars >= j + (rho ? rho : 1) + 1; }
  size_t local_vars(size_t num_vars) const { return sharded(num_vars) ? num_vars - rho : num_vars; }
  size_t local_len(size_t num_vars) const { return (size_t)1 << local_vars(num_vars); }
};

// A REPLICATED array of `len` entries (final_cts, a quotient level below the replication point) whose MSM every rank would
// repeat: an MSM is additive over point ranges (the chunk-then-sum of util/arithmetic/msm.rs:101-114), so rank s takes the
// entries [first, first + count) = [s len / R, (s + 1) len / R) and the partial commitments are added over the ranks.
// Arrays shorter than the world go to rank 0 whole; outside a sharded proof the range is the whole array.
struct ReplicatedRange {
  size_t first = 0, count = 0;
  ReplicatedRange(const Shard& sh, size_t len) {
    if (!sh.on || sh.R <= 1) {
      count = len;
    } else if (len < sh.R) {
      // (over the loopback communicator "the sum over the ranks" is R copies of this rank's part: a part that is empty on
      // every rank but 0 would make the sum the identity there, which no transcript can carry - a handful of points)
      count = sh.rank == 0 || sh.loopback ? len : 0;
    } else {
      const size_t per = len / sh.R;  // (len and R are powers of two in every use; a remainder would go to the last rank)
      first = sh.rank * per;
      count = sh.rank + 1 == sh.R ? len - first : per;
    }
  }
};

// Persistent host worker threads for the short host-side tails (window combines of an MSM batch): spawning
// std::threads per call costs ~50 us each, more than the work itself.
void host_parallel_for(size_t n, const std::function<void(size_t)>& fn);
// a host_parallel_for over about n items follows within `us` microseconds: wake that many workers now and let them poll
// for it (the poster otherwise pays the sleepers' wake-up, ~50 us, on the critical path between two kernels)
void host_parallel_prewake(size_t n, unsigned us);

struct ProfScope {
  Ctx& c;
  ProfRec rec;
  bool on;
  ProfScope(Ctx& c_, const char* name, double bytes, double muls, double items) : c(c_), on(c_.prof) {
    if (!on) return;
    snprintf(rec.name, sizeof rec.name, "%s", name);
    rec.bytes = bytes, rec.muls = muls, rec.items = items, rec.ms = 0;
    if (!c.prof_ev[0]) {
      LH_HIP(hipEventCreate(&c.prof_ev[0]));
      LH_HIP(hipEventCreate(&c.prof_ev[1]));
    }
    LH_HIP(hipEventRecord(c.prof_ev[0], c.stream));
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(c.prof_ev[1], c.stream);
    (void)hipEventSynchronize(c.prof_ev[1]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c.prof_ev[0], c.prof_ev[1]);
    rec.ms = ms;
    c.prof_recs.push_back(rec);
  }
};

// ------------------------------------------------------------------ kernel launchers (kernels_poly.hip)
void k_fr_from_u64(Ctx&, const uint64_t* in, size_t n, Fr* out);
void k_fr_from_u32(Ctx&, const uint32_t* in, size_t n, Fr* out);
void k_fr_to_repr(Ctx&, const Fr* in, size_t n, Fr* out);
void k_fr_from_repr(Ctx&, const Fr* in, size_t n, Fr* out);
void k_fr_binop(Ctx&, int op, const Fr* a, const Fr* b, size_t n, Fr* out);  // 0 add 1 sub 2 mul
void k_fr_mul_chain(Ctx&, const Fr* a, const Fr* b, size_t n, int iters, Fr* out);
void k_fr_batch_invert(Ctx&, const Fr* in, size_t n, Fr* out);
void k_fix_var(Ctx&, const Fr* in, size_t n_in, const Fr& x, Fr* out);
// binds `count` tables of n_in entries each in one launch
void k_fix_var_multi(Ctx&, const Fr* const* in, Fr* const* out, size_t count, size_t n_in, const Fr& x);
void k_eq_xy(Ctx&, const Fr* y, size_t num_vars, Fr* out);  // y: host array
// out_host (pinned) [i] = in[i][0] + (in[i][1] - in[i][0]) * x ; synchronises
void k_bind_first(Ctx&, const Fr* const* in, size_t count, const Fr& x, Fr* out_host);
// out_host (pinned) [i*k + j] = in[i][j], j < k ; synchronises
void k_gather_heads(Ctx&, const Fr* const* in, size_t count, int k, Fr* out_host);
void k_lincomb(Ctx&, const Fr* const* polys, const Fr* w, size_t count, size_t n, Fr* out);
// out[b] = lo + x (hi - lo), lo = sum_k w_k p_k[b], hi = sum_k w_k p_k[b + half]
void k_lincomb_fold(Ctx&, const Fr* const* polys, const Fr* w, size_t count, size_t half, const Fr& x, Fr* out);
// the same first step of g = sum_k coef_k col_k over 32-bit columns (entries beyond lens[k] are zero); false: too many columns
bool k_lincomb_fold_small(Ctx&, const uint32_t* const* cols, const size_t* lens, const Fr* coef, size_t count, size_t half,
                          const Fr& x, Fr* out);
// out[i] = <polys[i], weights>, i < count ; result on host
void k_inner_products(Ctx&, const Fr* const* polys, size_t count, const Fr* weights, size_t n, Fr* out_host);
// same with u32-valued polys
void k_inner_products_u32(Ctx&, const uint32_t* const* polys, size_t count, const Fr* weights, size_t n,
                          Fr* out_host);
// small-valued columns without their field-element views: 8 multiply-adds per term into a wide accumulator
void k_inner_products_small(Ctx&, const uint32_t* const* polys, size_t count, const Fr* weights, size_t n, Fr* out_host);
// the same against eq(y) given as the eq table of y[1..] (`half` entries) and y0
void k_inner_products_small_half(Ctx&, const uint32_t* const* polys, size_t count, const Fr* eq_half, size_t half,
                                 const Fr& y0, Fr* out_host);
// An eq-factored degree-2 sum-check over ONE table that still is a 32-bit column (Surge over the output column, lasso.cpp):
// out_host[0..3] = sum_b e0[b] col[2b], sum_b e0[b] col[2b+1], sum_q (e0[2q] + e0[2q+1]) col[4q+2], the same with col[4q+3]
// (the claim's two halves - and the sums of rounds 0 and 1), b < 2 quads, q < quads
void k_inner_products_small_quads(Ctx&, const uint32_t* col, const Fr* e0, size_t quads, Fr* out_host);
// d_out[4 k + t] (DEVICE, queued on the stream) = sum_q e1[q] cols[k][4 q + t], t = 0..3, q < quads; lens: multiples of 4
void k_inner_products_quads(Ctx&, const uint32_t* const* cols, const size_t* lens, size_t count, const Fr* e1, size_t quads,
                            Fr* d_out);
// out[i] = sum_k w[k] (cols[k][4i..4i+3] bound with (r0, r1)), i < 2 size; out_host[e] = sum_b eq_level[b] out[2b + e]
void k_lincomb_bind2(Ctx&, const uint32_t* const* cols, const size_t* lens, const Fr* w, size_t count, const Fr& r0, const Fr& r1,
                     const Fr* eq_level, size_t size, Fr* out, Fr* out_host);
// round 2 binds r0 and r1 at once: out[i] = the column's entries 4i..4i+3 bound with (r0, r1), i < 2 size;
// out_host[0] = sum_b eq_level[b] out[2b+1]
void k_sc_round_u32_bind2(Ctx&, const uint32_t* col, const Fr* eq_level, const Fr& r0, const Fr& r1, size_t size, Fr* out,
                          Fr* out_host);
// out[i] = sum_k wfr[k] fr[k][i] + sum_k wsm[k] sm[k][i], i < n; u32 column k has sm_len[k] entries (zero beyond)
void k_lincomb_mixed(Ctx&, const Fr* const* fr, const Fr* wfr, size_t num_fr, const uint32_t* const* sm,
                     const size_t* sm_len, const Fr* wsm, size_t num_sm, size_t n, Fr* out);
// product tree level: out[i] = in[i] * in[half + i]
void k_tree_up(Ctx&, const Fr* in, size_t half, Fr* out);
void k_tree_up_multi(Ctx&, const Fr* const* in, Fr* const* out, size_t count, size_t half);
// every level above level H[i] (2^(H+1) nodes at in[i], H <= 9) of `count` product trees, one launch:
// level h < H lands at out[i] + (2^(h+1) - 2)
void k_tree_tops(Ctx&, const Fr* const* in, Fr* const* out, const int* H, size_t count);
// fractional layer: (p_l q_r + p_r q_l, q_l q_r)
void k_frac_up(Ctx&, const Fr* p, const Fr* q, size_t half, Fr* vp, Fr* vq);
// KZG quotient step at level i: q = hi - lo ; lo' = lo + (hi - lo) * x
void k_quotient_step(Ctx&, const Fr* rem, size_t half, const Fr& x, Fr* q, Fr* rem_out);
// Lasso fingerprints: rs = dim*g2 + e*g + ts - tau ; ws = rs + 1
void k_lasso_rw_leaves(Ctx&, const uint32_t* dim, const uint32_t* e, const uint32_t* ts, size_t n,
                       const Fr& gamma, const Fr& gamma2, const Fr& tau, Fr* rs, Fr* ws);
// the same plus the level above the leaves of both trees (rs_up[i] = rs[i] * rs[i + n/2], n/2 entries each)
void k_lasso_rw_leaves_up(Ctx&, const uint32_t* dim, const uint32_t* e, const uint32_t* ts, size_t n, const Fr& gamma,
                          const Fr& gamma2, const Fr& tau, Fr* rs, Fr* ws, Fr* rs_up, Fr* ws_up);
// init = m*g2 + T[m]*g - tau ; fin = init + final_cts[m]
void k_lasso_if_leaves(Ctx&, int subtable, uint32_t chunk_bits, const uint32_t* final_cts, size_t m,
                       const Fr& gamma, const Fr& gamma2, const Fr& tau, Fr* init, Fr* fin);
// Lasso witness: counters and subtable reads
// keep_sorted / keep_index (optional, n entries each): the column sorted by value and the positions it came from
// OR of all entries of every column (its bit length bounds the values); synchronises
void k_or_u32(Ctx&, const uint32_t* const* cols, size_t count, size_t n, uint32_t* out_host);
void k_fill_u32(Ctx&, uint32_t* out, uint32_t value, size_t n);
// out[i] = a[i] | b[i] << shift
void k_pack_u32(Ctx&, const uint32_t* a, const uint32_t* b, uint32_t shift, size_t n, uint32_t* out);
// v = col[i + half] - col[i] + offset (entries beyond `len` are zero; 0 < v < 2^33): out_lo[i] = v, or with out_hi:
// out_lo[i] = v & 0xffff, out_hi[i] = v >> 16
void k_delta_u32(Ctx&, const uint32_t* col, size_t len, size_t half, uint64_t offset, uint32_t* out_lo, uint32_t* out_hi);
// access counters of all `cc` chunk columns (n lookups into m cells each) in one launch set, one bad-index readback per call;
// keep_sorted / keep_index: null, or per column where to leave the sorted values and their positions
void k_lasso_counters(Ctx&, const uint32_t* const* dims, size_t cc, size_t n, size_t m, uint32_t* const* read_ts,
                      uint32_t* const* final_cts, uint32_t* const* keep_sorted = nullptr, uint32_t* const* keep_index = nullptr);
// the sharded counters' steps (lasso.cpp lasso_counters_sharded), ALL `cc` chunk columns per call: partition the local
// lookups by address owner (sidx[q]: local indices in send order; send[q]: (address on the owner) << hi_bits | local index >>
// shard_bit; start_host[q * (R + 1) + o]: first send position of owner o in column q), rank the received lookups on the
// owner (recv[q]: n_recv[q] keys, one segment per sender; counts: cc * m_loc), scatter the returned ranks, assemble final_cts
void k_cs_partition(Ctx&, const uint32_t* const* dims, size_t cc, size_t n, size_t m, unsigned rho, unsigned j, unsigned hi_bits,
                    uint32_t* const* sidx, uint32_t* const* send, uint32_t* start_host, bool* bad_out);
void k_cs_rank(Ctx&, const uint32_t* const* recv, const size_t* n_recv, size_t cc, unsigned hi_bits, unsigned a_bits, size_t m_loc,
               uint32_t* const* ret, uint32_t* counts);
void k_cs_scatter(Ctx&, const uint32_t* const* back, const uint32_t* const* sidx, size_t cc, size_t n, uint32_t* const* read_ts);
void k_cs_final(Ctx&, const uint32_t* all_counts, size_t cc, size_t m, unsigned rho, size_t m_loc, uint32_t* const* final_cts);
void k_lasso_subtable_read(Ctx&, int subtable, uint32_t chunk_bits, const uint32_t* dim, size_t n, uint32_t* e);
// a[k] = g(E_0[k],..): small-integer evaluation into Fr
struct LassoG {
  uint32_t num_terms;
  Fr coeff[LH_LASSO_MAX_TERMS];
  uint8_t nfac[LH_LASSO_MAX_TERMS];
  uint8_t fac[LH_LASSO_MAX_TERMS][LH_SC_MAX_FACTORS];
  const uint32_t* e[LH_LASSO_MAX_MEMORIES];
};
void k_lasso_output(Ctx&, const LassoG& g, size_t n, Fr* a);
struct LassoGSmall {  // g = sum_t coeff[t] * E_{fac[t]} with 32-bit coefficients and a value that fits 32 bits
  uint32_t num_terms;
  uint32_t coeff[LH_LASSO_MAX_TERMS];
  uint8_t fac[LH_LASSO_MAX_TERMS];
  const uint32_t* e[LH_LASSO_MAX_MEMORIES];
};
void k_lasso_output_small(Ctx&, const LassoGSmall& g, size_t n, uint32_t* a);
// canonical value of every entry as u32; false if an entry is not below 2^bits
bool k_fr_to_index(Ctx&, const Fr* in, size_t n, uint32_t bits, uint32_t* out);
// both tables hold Montgomery residues in [0, r): equal values have equal limbs
bool k_fr_tables_equal(Ctx&, const Fr* a, const Fr* b, size_t n);

// ------------------------------------------------------------------ communicator (comm.cpp)
void rccl_unique_id(uint8_t out[LH_RCCL_UNIQUE_ID_BYTES]);
void comm_attach_rccl(Ctx&, int rank, int size, const uint8_t id[LH_RCCL_UNIQUE_ID_BYTES], size_t shard_bit);
void comm_probe_host_recv(Ctx&);  // attach-time self-test of the all-reduce into pinned host memory (sets Ctx::comm_host_recv_ok)
void comm_attach_loopback(Ctx&, int rank, int size, size_t shard_bit);
void comm_detach(Ctx&);
// recv = size * bytes, rank-major.  _dev: device buffers, enqueued on the ctx's stream (staged through the host when
// the communicator has no device collective); _host: host buffers, synchronous
void comm_all_gather_dev(Ctx&, const void* d_send, void* d_recv, size_t bytes);
void comm_all_gather_host(Ctx&, const void* send, void* recv, size_t bytes);
// personalised exchange of `nbuf` device buffers at once (elements of `elem` bytes; the chunk columns of the sharded access
// counters): from buffer b this rank sends send_cnt[b * R + p] elements at d_send[b] + send_off[b * R + p] to every peer p and
// receives recv_cnt[b * R + p] from it at d_recv[b] + recv_off[b * R + p] - ONE collective (RCCL: one group of sends and
// receives).  peer_off[b * R + p]: where, inside p's send buffer b, the segment for this rank starts; the send buffers are
// slices of ONE allocation, buffer b at send_base + b * send_stride elements, the same stride on every rank - both only used
// by transports without point-to-point sends (the whole allocation is staged through an all-gather).
void comm_all_to_all_multi(Ctx&, size_t nbuf, const void* const* d_send, const size_t* send_off, const size_t* send_cnt,
                           void* const* d_recv, const size_t* recv_off, const size_t* recv_cnt, const size_t* peer_off,
                           const void* send_base, size_t send_stride, size_t elem);
// v[i] = sum over the ranks of v[i], `count` field elements in a device buffer the ctx's stream owns; the sums are left in
// out_host (pinned) followed by the flag `seq` (the closing steps of a sharded sum-check round)
void comm_sum_publish(Ctx&, const Fr* d_part, Fr* d_scratch, size_t count, Fr* out_host, uint32_t seq);
// the all-reduce variant of the same step (Options::comm_round 1 / 2): d_lanes = the 8 * count tagged u64 lanes the round
// kernel left (ScFinishArgs::wide, tag = c.sc_tag); ONE collective adds them over the ranks into pinned host memory, the
// host waits until every lane carries the ranks' tags and reduces mod r.  Synchronous: out_host holds the sums on return.
void comm_sum_lanes(Ctx&, uint64_t* d_lanes, uint64_t* d_scratch, size_t count, Fr* out_host);
// a fresh tag for the next all-reduce round (never 0, below 2^24 / R)
uint32_t comm_next_tag(Ctx&);
// loopback all-reduce: out[i] = in[i] * R (every peer is a copy of this rank), one launch (kernels_poly.hip)
void k_loopback_allreduce_lanes(Ctx&, const uint64_t* d_in, size_t n, size_t R, uint64_t* out);

// ------------------------------------------------------------------ sharding helpers (kernels_poly.hip)
// inverse of k_shard_extract over the all-gathered shards: global[g] = gathered[s(g) * n_local + local(g)]
void k_shard_merge(Ctx&, const void* gathered, size_t n_local, size_t j, size_t rho, size_t elem, void* global);
// out[t][(hi * R + s) * block + lo] = gathered[(s * count + t) * n_local + hi * block + lo]: the residual tables of a
// sharded sum-check (block = 1 once the shard bits have reached bit 0) or tree levels / remainders at the replication
// point (block = n_local: concatenation)
void k_gather_interleave(Ctx&, const Fr* gathered, size_t count, size_t n_local, size_t R, size_t block, Fr* const* out);
// out_host[x] = sum_s all[s * D + x], x < D; then the flag: the closing step of a sharded sum-check round
void k_sum_publish(Ctx&, const Fr* all, size_t R, size_t D, Fr* out_host, uint32_t seq);
// local[idx] = global[((idx >> j) << (j + rho)) | (s << j) | (idx & (2^j - 1))], elements of `elem` bytes (4, 32, 64)
void k_shard_extract(Ctx&, const void* global, size_t n_local, size_t j, size_t rho, size_t s, size_t elem, void* local);
// the loopback communicator's all-gather: recv block s = send rotated by 32 s bytes (comm.cpp), one launch
void k_loopback_gather(Ctx&, const void* d_send, void* d_recv, size_t bytes, size_t R);
// out[i] = in[i] * w
void k_scale(Ctx&, const Fr* in, const Fr& w, size_t n, Fr* out);

// ------------------------------------------------------------------ sum-check round (kernels_sumcheck.hip)
constexpr int SC_MAX_TABLES = 72;  // 32 product trees (8 memories) + eq fit one GKR batch; kernel arguments stay < 4 KB
struct ScRound {
  // tables of the current round: BIND ? 4*size entries in, 2*size out : 2*size entries in
  const Fr* in[SC_MAX_TABLES];
  Fr* out[SC_MAX_TABLES];
  uint32_t num_tables;
  uint32_t num_terms;
  int32_t global_eq;  // table id multiplied onto the sum, or -1
  Fr coeff[LH_SC_MAX_TERMS];
  uint8_t coeff_is_one[LH_SC_MAX_TERMS];
  uint8_t nfac[LH_SC_MAX_TERMS];
  uint8_t fac[LH_SC_MAX_TERMS][LH_SC_MAX_FACTORS];
  Fr r;  // challenge of the previous round (BIND only)
  // "eq factoring" (sumcheck.cpp): when set, global_eq is -1 and the eq factor of the expression is eq_level[b], the eq
  // table over the variables AFTER this round's; the kernel returns q(X) = sum_b eq_level[b] * g(X, b)
  const Fr* eq_level;
  // product-pair shape (every term coeff_m * l_m * r_m over 2 num_terms distinct tables, factored eq; the generic layers of
  // the grand products): 1 = the factored degree-2 rounds run sc_round_pp_kernel (four products per Montgomery reduction);
  // 2 = this BIND round also FOLDS the coefficients into the left factors (l'_m = coeff_m l_m is what it stores: the
  // caller treats the coefficients as one from then on and divides them out of the final evaluations)
  uint8_t pp;
};
// evals_host[0..degree) receives sum_b expr at X = 1..degree (X = 0 is derived by the caller)
void k_sc_round(Ctx&, const ScRound& rd, int degree, bool bind, size_t size, Fr* evals_host);
// true when k_sc_round would run the streaming one-thread-per-pair kernel for this shape (not the LDS-staged one)
bool k_sc_round_streams(const ScRound& rd, int degree, size_t size);
// out[i] = in[2 i] + in[2 i + 1]: the eq table over one variable less (eq factoring)
// levels[k] = pair sums of levels[k - 1] (levels[-1] = `in`, n_in entries, a power of two), k < nlev: the levels of a
// factored eq table (host.hpp EqFactoring), 9 per launch
void k_eq_levels(Ctx&, const Fr* in, size_t n_in, Fr* const* levels, size_t nlev);
// Batch-opening shape sum_m eq_m * poly_m with every eq factored: per term q_m(0) = sum_b E_m[b] v0, q_m(1) = sum_b E_m[b] v1
constexpr int SC_OPEN_MAX_TERMS = 6;
struct ScOpenRound {
  const Fr* in[SC_OPEN_MAX_TERMS];
  Fr* out[SC_OPEN_MAX_TERMS];
  const Fr* eq_level[SC_OPEN_MAX_TERMS];
  uint32_t num_terms;
  Fr r;
};
// out_host[2 m], out_host[2 m + 1] = q_m(0), q_m(1)
void k_sc_round_open(Ctx&, const ScOpenRound& rd, bool bind, size_t size, Fr* out_host);

// Streaming round of a grand-product layer whose trees come in pairs (A_i, A_i + 1) - Lasso's read set and write set at
// the leaf level, write = read + 1 entry by entry, and binding keeps the "+ 1".  With cs_i = c_A + c_B, k_i = c_B / cs_i:
//   c_A l r + c_B (l + 1)(r + 1) = cs_i (l + k_i)(r + k_i) + const_i
// so only the A tables are read and bound and a pair of trees costs one product per point.  Factored eq (eq_level as in
// ScRound); out_host[x - 1] = sum_b eq_level[b] * sum_i cs_i (l_i + k_i)(r_i + k_i) at X = x, x = 1, 2 (the caller adds
// the constants).
constexpr int SC_RW_MAX_PAIRS = 8;
struct ScRwRound {
  const Fr* l[SC_RW_MAX_PAIRS];
  const Fr* r[SC_RW_MAX_PAIRS];
  Fr* lo[SC_RW_MAX_PAIRS];
  Fr* ro[SC_RW_MAX_PAIRS];
  Fr cs[SC_RW_MAX_PAIRS], k[SC_RW_MAX_PAIRS];
  const Fr* eq_level;
  Fr rchal;
  uint32_t num_pairs;
};
// fold (BIND rounds): store l' = cs (l + k), r' = r + k instead of the bound l, r - the later rounds are ScRound::pp ones
void k_sc_round_rw(Ctx&, const ScRwRound& rd, bool bind, size_t size, Fr* out_host, bool fold = false);

// Resident tail: once the live tables of a sum-check fit the LDS of a few CUs, ONE launch runs all remaining rounds.
// G workgroups each keep a contiguous slice of every table in LDS (binding never crosses a slice); per round every
// workgroup leaves its D partial sums in device memory and draws a ticket, the one that draws the last ticket adds
// them up and sends the message to the host; when a slice is down to one entry the slices are handed (through device
// memory, same ticket protocol) to the workgroup that arrives last, which runs the remaining log2(G) rounds alone.
// Host <-> kernel traffic is self-validating 16-byte chunks {seq, 3 limbs} (TailChunk): a chunk is written with ONE
// 16-byte store and read with ONE 16-byte load on either side, so a message needs no flag and no fence (one PCIe
// crossing per direction and round instead of two).  After the last challenge the kernel publishes the final
// evaluation of the first `num_out` tables (plain values, flag = seq0 + rounds).
// No launch and no completion latency per round, tables never leave LDS.
struct TailChunk {
  uint32_t w[4];  // {seq, limb 3j, limb 3j+1, limb 3j+2} (chunk j = 0..2 of a field element; the last holds two limbs)
};
struct TailMbox {      // pinned, written by the host only: the challenge of round i carries seq0 + i in every chunk;
  TailChunk c[3];      // SC_TAIL_ABORT makes the kernel exit
  uint32_t pad[4];
};
constexpr uint32_t SC_TAIL_ABORT = 0xffffffffu;
constexpr int SC_TAIL_MAX_DEGREE = 6;
// largest resident table length (power of two, 0 = tail not applicable) for this expression
size_t k_sc_tail_capacity(const Ctx&, const ScRound& rd, int degree);
// rd.in: entry tables; first_bind: they hold 2*n0 entries and are bound with rd.r first.  Returns immediately.
// msg_host: 3 * degree chunks (value x at chunks 3x..3x+2), out_host: num_out field elements.
void k_sc_tail_launch(Ctx&, const ScRound& rd, int degree, size_t n0, bool first_bind, size_t num_out, uint32_t seq0,
                      TailChunk* msg_host, Fr* out_host);
// after a tail that ended early: the ticket counter is wherever the workgroups left it
void k_sc_tail_resync(Ctx&);

// ------------------------------------------------------------------ resident grand-product layers (kernels_gkr.hip)
// The layers of a product-tree argument (prove_grand_product) whose tables fit on-chip, in ONE launch: layer loop, eq
// tables and rounds inside the kernel, host <-> kernel traffic in TailChunk messages.  Per layer the host sends
// [c_0 .. c_{B-1}, y_0 .. y_{h-1}] (sequence number `seq`), the kernel answers every round j with q(1), q(2) of the
// eq-factored round polynomial (seq + 1 + j; the challenge comes back under the same number through the ctx's TailMbox)
// and ends the layer with the 2 B final evaluations (l'_k = c_k l_k and r_k per tree; flag = seq + h + 1).
constexpr int GKR_MAX_TREES = 16;
constexpr int GKR_MAX_VARS = 16;          // a resident layer has at most GKR_MAX_VARS - 2 variables
constexpr uint32_t GKR_CAP = 128;         // entries of a table one workgroup holds
constexpr uint32_t GKR_THREADS = 256;
constexpr uint32_t GKR_MSG_CHUNKS = 3 * (GKR_MAX_TREES + GKR_MAX_VARS);
constexpr uint32_t GKR_START_FAILED = 0xfffffffeu;  // first message chunk: the workgroups did not all start (see the kernel)
// A layer can also be the TAIL of a sum-check that ran its first rounds elsewhere (GKR_F_* flags): the tables come as
// separate left / right pointers, possibly still to be bound with the previous challenge, the eq level of the first
// resident round is read from memory, the coefficients (and the constant offsets of the read/write leaf layer,
// cs (l + k)(r + k)) come with the descriptor instead of a layer message.
enum { GKR_F_SPLIT = 1, GKR_F_BIND = 2, GKR_F_EQ = 4, GKR_F_NOMSG = 8, GKR_F_KOFF = 16 };
struct GkrLayerDev {
  const Fr* lv[GKR_MAX_TREES];  // the level of tree k: 2^(h + 1) nodes, left factors first (GKR_F_SPLIT: the left table)
  const Fr* rv[GKR_MAX_TREES];  // GKR_F_SPLIT: the right table
  uint32_t h, B;                // variables of the layer's tables, trees
  uint32_t g, s_log;            // g workgroups x 2^s_log entries = 2^h
  uint32_t seq;
  uint32_t flags;
  const Fr* eq_level;           // GKR_F_EQ: E of the first round, 2^(h - 1) entries
  Fr r_prev;                    // GKR_F_BIND: the tables hold 2^(h + 1) entries and are bound with this first
  Fr coef[GKR_MAX_TREES];       // GKR_F_NOMSG: the coefficients c_k
  Fr koff[GKR_MAX_TREES];       // GKR_F_KOFF: l'_k = c_k (l_k + koff_k), r'_k = r_k + koff_k
};
bool k_gkr_resident_geometry(uint32_t h, uint32_t* g, uint32_t* s_log);
// returns immediately; msg_host: 6 chunks, out_host: 2 * GKR_MAX_TREES field elements (pinned)
void k_gkr_resident_launch(Ctx&, const GkrLayerDev* layers, size_t num_layers, TailChunk* msg_host, Fr* out_host);

// ------------------------------------------------------------------ general expressions (kernels_expr.hip)
struct ExtRound {
  const Fr* in[SC_MAX_TABLES];
  Fr* out[SC_MAX_TABLES];
  uint32_t num_tables, num_terms;
  // device arrays, constant over the rounds of one sum-check
  const Fr* coeff;          // [num_terms]
  const uint8_t* is_one;    // [num_terms]
  const uint32_t* off;      // [num_terms + 1] into fac / store
  const uint8_t* fac;       // table ids
  const uint8_t* store;     // first occurrence of a table stores its bound pair
  Fr r;
};
void k_sc_round_ext(Ctx&, const ExtRound& rd, int degree, bool bind, size_t size, Fr* evals_host);
void k_rotate_gather(Ctx&, const Fr* poly, size_t num_vars, int rot, uint32_t primitive, uint32_t x_inv, Fr* out);
void k_identity_table(Ctx&, size_t n, Fr* out);
void k_identity_table_shard(Ctx&, size_t n_local, size_t j, size_t rho, size_t rank, Fr* out);
// BooleanHypercube::iter() order and its inverse (bh.rs:127-141), 2^num_vars entries each
void k_bh_order(Ctx&, size_t num_vars, uint32_t primitive, uint32_t* order, uint32_t* nth);
void k_one_hot_table(Ctx&, size_t n, size_t hot, Fr* out);
// Straight-line program over tables, constants and a small register file (the device counterpart of the
// reference's ExpressionRegistry calculations, util/expression/evaluator.rs:135-323): instruction i is
//   code[2i]   = op | dst << 4 | a_kind << 8 | b_kind << 10        code[2i+1] = a_idx | b_idx << 16
enum { PROG_ADD = 0, PROG_SUB = 1, PROG_MUL = 2, PROG_NEG = 3, PROG_MOV = 4 };
// (PROG_PAIR: table `idx` holds ONE entry per pair, the same at every evaluation point - the level of a factored eq table)
enum { PROG_REG = 0, PROG_ATOM = 1, PROG_CONST = 2, PROG_PAIR = 3 };
constexpr int PROG_MAX_REGS = 8;
struct ProgRound {
  const Fr* in[SC_MAX_TABLES];  // tables of 2 * size entries (already bound)
  uint32_t num_tables, num_instrs, num_regs, result_reg;
  const uint32_t* code;  // device
  const Fr* consts;      // device
};
// runtime-compiled form of a program (jit.cpp); nullptr = not compiled (disabled, too small, or compilation failed)
struct JitKernel;
bool jit_enabled(size_t num_vars);
// development / tests: the source the runtime compiler would be given (host code only)
std::string jit_debug_source(const uint32_t* code, size_t num_instrs, uint32_t num_regs, uint32_t result_reg, int degree);
const JitKernel* jit_sc_round(const Ctx&, const uint32_t* host_code, size_t num_instrs, uint32_t num_regs, uint32_t result_reg, int degree);
unsigned jit_blocks_per_cu(const JitKernel*);
void jit_launch(Ctx&, const JitKernel*, const ProgRound& pr, unsigned points, unsigned grid, size_t size, Fr* partials, const ScFinishArgs& fin);
// evals_host[0..points) = sum over pairs of program(tables at X), X = 1..points; `jit`: run the compiled form
void k_sc_round_prog(Ctx&, const ProgRound& pr, int points, size_t size, Fr* evals_host, const JitKernel* jit = nullptr);
// the linear part of a zero-check beside its eq-factored part (expr.cpp): out_host[0] = sum_i coeff_i sum_b t_i[2 b],
// out_host[1] = the same over the odd entries; queued in FRONT of the round's kernel, nobody waits for it alone
constexpr int LIN_MAX_TABLES = 8;
struct LinSums {
  const Fr* t[LIN_MAX_TABLES];
  Fr coeff[LIN_MAX_TABLES];
  uint32_t count;
};
void k_lin_sums(Ctx&, const LinSums& ls, size_t size, Fr* out_host);
enum { ROWS_ATOM_POLY = 0, ROWS_ATOM_IDENTITY = 1, ROWS_ATOM_LAGRANGE = 2 };
struct RowsAtom {
  const Fr* table;
  int32_t rot;
  uint32_t kind;
  uint64_t hot;  // Lagrange: the row where it is 1
};
struct RowsExpr {
  uint32_t num_terms, num_vars, primitive, x_inv;
  const Fr* coeff;       // device [num_terms]
  const uint32_t* off;   // device [num_terms + 1]
  const uint8_t* fac;    // device: atom ids
  const RowsAtom* atoms; // device
};
void k_expr_rows(Ctx&, const RowsExpr& e, size_t n, Fr* out);

// ------------------------------------------------------------------ HyperPlonk witness polys (kernels_plonk.hip)
// m[j] = #{i : input[i] == table[j]} on the LAST row j holding that value; false if an input is missing
bool k_lookup_m(Ctx&, const Fr* input, const Fr* table, size_t n, Fr* m_out);
void k_lookup_h(Ctx&, const Fr* input, const Fr* table, const Fr* m, const Fr& gamma, size_t n, Fr* h);
void k_permutation_z(Ctx&, const Fr* const* values, const Fr* const* perms, size_t num_perm, size_t num_chunks,
                     size_t num_vars, const Fr& beta, const Fr& gamma, const uint32_t* d_order, const uint32_t* d_nth,
                     Fr* const* z_out);
void k_permutation_products(Ctx&, const Fr* const* values, const Fr* const* perms, size_t num_perm, size_t num_chunks,
                            size_t num_vars, const Fr& beta, const Fr& gamma, size_t n, size_t sj, size_t srho, size_t srank,
                            Fr* const* prod_out);
void k_permutation_z_from_products(Ctx&, const Fr* const* prods, size_t num_chunks, size_t num_vars, const uint32_t* d_order,
                                   const uint32_t* d_nth, Fr* const* z_out, size_t sj, size_t srho, size_t srank);
void k_scatter_rows(Ctx&, const uint32_t* d_rows, const Fr* d_vals, size_t count, size_t n, Fr* table);

// ------------------------------------------------------------------ Zeromorph / univariate KZG (kernels_zm.hip)
void k_powers(Ctx&, const Fr& s, size_t n, Fr* out);  // out[i] = s^i
// q: quotients flat (q_k at offset 2^k - 1); ypow / q_scalars: host arrays of num_vars elements
void k_zm_qhat(Ctx&, const Fr* q, size_t num_vars, const Fr* ypow, Fr* q_hat);
void k_zm_combine(Ctx&, const Fr* poly, const Fr* q_hat, const Fr* q, size_t num_vars, const Fr& z,
                  const Fr* q_scalars, Fr* f);
// out[i] = sum_{j >= i} f_j x^(j - i): out[1..] is the quotient of f by (X - x), out[0] = f(x)
void k_suffix_horner(Ctx&, const Fr* f, size_t n, const Fr& x, Fr* out);

// ------------------------------------------------------------------ radix sort (sort.hip)
// stable sort of (u32 key, u32 value) pairs by the low `bits` bits of the key; inputs are preserved
struct SortSlab {
  const uint32_t* keys_in;
  uint32_t* keys_out;
  const uint32_t* vals_in;
  uint32_t* vals_out;
  size_t n;
  unsigned bits;
  unsigned first_bit = 0;
};
// (vals_in == nullptr: the values are the positions 0 .. n - 1; the sort is by the key bits [first_bit, first_bit + bits))
void sort_pairs_u32(Ctx&, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                    unsigned bits, unsigned first_bit = 0);
// `count` independent sorts as ONE launch set per radix pass (temporary storage from the arena: the caller's ArenaScope)
void sort_pairs_u32_batched(Ctx&, const SortSlab* slabs, size_t count);
void sort_pairs_u64(Ctx&, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                    unsigned bits);

// ------------------------------------------------------------------ MSM (msm.hip)
struct MsmJob {
  const void* scalars;  // Fr (Montgomery) or u32
  bool scalars_u32;
  const G1Affine* bases;
  size_t n;
  // Optional: this column is a table of another column of the batch, scalars[i] = table[parent.scalars[i]] (Lasso's
  // E = T[dim]).  When the parent ends up with ONE window whose bucket index is the parent's value, this job's buckets
  // are sums of the PARENT'S buckets (S_v = sum of B_d over T[d] = v) and no pass over its n points is made: 2^in_bits
  // bucket additions instead of n point additions.  Otherwise the job runs as an ordinary u32 column.
  int derived_parent = -1;             // index of the parent job in the same batch
  const uint32_t* d_table = nullptr;   // device: T[d], d < 2^table_in_bits; T[0] must be 0
  const uint32_t* d_order = nullptr;   // device: 0 .. 2^table_in_bits - 1 sorted by T (any order inside equal T)
  uint32_t table_in_bits = 0, table_out_bits = 0;
  // Optional (u32 columns): the column's values in ascending order and the positions they came from (Lasso's access
  // counters sort every dim column anyway).  A job that ends up with one window per (job, window) slab takes its sorted
  // entry stream from these instead of emitting and sorting it again.
  const uint32_t* sorted_scalars = nullptr;
  const uint32_t* sorted_index = nullptr;
  // Optional (u32 columns): TWO small-valued columns over the same bases packed into one, scalars[i] = s1[i] | s2[i] <<
  // pack_shift (4 <= pack_shift, at most MSM_PACK_MAX_BITS significant bits together).  One pass over the points fills one
  // bucket set indexed by the packed value; the reduction weighs it twice: out[j] = sum s1[i] B_i, *out_second = sum s2[i] B_i.
  uint32_t pack_shift = 0;
  G1Affine* out_second = nullptr;  // host
  // Optional: every scalar < 2^known_bits, promised by the caller (0: measured by a pass over the scalars - a column
  // that happens to be narrower then gets fewer windows; a promise that is too small loses the upper bits)
  uint32_t known_bits = 0;
  // Optional (Fr columns): the window table of `bases` (k_msm_window_table: entry w * n + i = 2^(win_table_c w) * bases[i],
  // w < win_table_W).  A column that needs 2 .. win_table_W windows of win_table_c bits then files all of them into one
  // bucket set (one bucket reduction, no doublings) - same sum, W-fold fewer buckets.
  const G1Affine* win_table = nullptr;
  uint32_t win_table_c = 0, win_table_W = 0;
};
constexpr uint32_t MSM_PACK_MAX_BITS = 20;
// two columns share a pass only while the pass saved (one mixed addition per point) outweighs the bucket set it adds to the
// reduction (2^bits buckets at ~3.5 full additions each): from 8 points per bucket on.  (The single-GPU sizes that pack are
// far above it; a rank's shard of a level - 2^20 points against 2^20 buckets at 8 ranks - is not.)
constexpr size_t MSM_PACK_MIN_POINTS_PER_BUCKET = 8;
Ctx& ctx_helper(Ctx&);            // the ctx's helper ctx (created on first use, destroyed with the ctx)
void open_precommit_cancel(Ctx&);  // waits for a running precommit and drops it (open_columns.cpp)
uint32_t msm_window_bits(size_t n);  // window width msm_batch picks for a full-width (254-bit) column of n points
void k_msm_window_table(Ctx&, const G1Affine* bases, size_t n, uint32_t cbits, uint32_t W, G1Affine* out);
// Runs all jobs as one batched Pippenger; out[j] is the affine sum (identity = (0,0)).
// `overlap` (optional): host work that needs none of this batch's results - run once, after the batch's kernels are
// queued and before the host waits for the window sums (it overlaps the device's work instead of following it)
// Returns whether the host waited for the ctx's stream on the way (false only for a batch without a single entry and with every
// width promised: then nothing queued before the call is known to have run).
bool msm_batch(Ctx&, const MsmJob* jobs, size_t num_jobs, G1Affine* out_host, const std::function<void()>* overlap = nullptr);
int msm_slab_log();  // jobs of >= 2^this points are sorted slab by slab (and can take MsmJob::sorted_*)
// out[i] = scalars[i] * G (fixed-base), normalised to affine; all on device
void k_fixed_base_mul_g(Ctx&, const Fr* scalars, size_t n, G1Affine* out);

}  // namespace lh
