// Device context and workspace arena.
#include <string.h>
#include <stdlib.h>
#include <ctype.h>
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <functional>
#include <memory>
#include <emmintrin.h>
#include <hip/hip_ext.h>
#include <vector>
#include <chrono>
#include "dev.hpp"

namespace lh {

static thread_local std::string g_last_error;
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
const char* get_last_error() { return g_last_error.c_str(); }

Arena::~Arena() {
  for (auto& b : blocks_) (void)hipFree(b.p);
}

void* Arena::alloc(size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  return alloc_raw(bytes);
}
void* Arena::alloc_raw(size_t bytes) {
  while (true) {
    if (!blocks_.empty()) {
      Block& b = blocks_[cur_];
      if (b.used + bytes <= b.size) {
        void* p = b.p + b.used;
        b.used += bytes;
        size_t tot = 0;
        for (size_t i = 0; i <= cur_; i++) tot += blocks_[i].used;
        if (tot > high_) high_ = tot;
        return p;
      }
      if (cur_ + 1 < blocks_.size()) {
        cur_++;
        blocks_[cur_].used = 0;
        continue;
      }
    }
    size_t want = blocks_.empty() ? ((size_t)64 << 20) : blocks_.back().size * 2;
    if (want < bytes) want = bytes;
    void* p = nullptr;
    LH_HIP(hipMalloc(&p, want));
    blocks_.push_back(Block{(char*)p, want, 0});
    cur_ = blocks_.size() - 1;
  }
}

void Arena::release(Mark m) {
  if (blocks_.empty()) return;
  for (size_t i = m.block + 1; i < blocks_.size(); i++) blocks_[i].used = 0;
  cur_ = m.block;
  blocks_[cur_].used = m.used;
}

void Ctx::opt_in_lds(const void* fn, int bytes) {
  for (const void* f : lds_opted)
    if (f == fn) return;
  LH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  lds_opted.push_back(fn);
}

static const struct {
  const char* name;
  int64_t Options::*field;
  int64_t lo, hi;  // accepted range (shift counts stay below 64, lengths are never negative)
} OPTION_TABLE[] = {
    {"open_small_min_vars", &Options::open_small_min_vars, 0, 64}, {"open_small_depth", &Options::open_small_depth, 0, 2},
    {"sc_eq_factoring", &Options::sc_eq_factoring, 0, 1},          {"lasso_pack_ts", &Options::lasso_pack_ts, 0, 1},
    {"sc_tail", &Options::sc_tail, 0, 1},                          {"sc_tail_max_len", &Options::sc_tail_max_len, 0, (int64_t)1 << 20},
    {"shard_exchange_log", &Options::shard_exchange_log, 0, 40},   {"msm_window_tables", &Options::msm_window_tables, 0, 40},
    {"open_precommit", &Options::open_precommit, 0, 64},           {"gkr_resident", &Options::gkr_resident, 0, 1},
    {"sc_pp_fold", &Options::sc_pp_fold, 0, 2},                    {"msm_half_batches", &Options::msm_half_batches, 0, 1},
    {"comm_round", &Options::comm_round, 0, 2},
};

int64_t* Options::find(const char* name) {
  if (!name) return nullptr;
  for (const auto& o : OPTION_TABLE)
    if (strcmp(o.name, name) == 0) return &(this->*o.field);
  return nullptr;
}
bool Options::in_range(const char* name, int64_t value) {
  if (!name) return false;
  for (const auto& o : OPTION_TABLE)
    if (strcmp(o.name, name) == 0) return value >= o.lo && value <= o.hi;
  return false;
}

Options::Options() {
  for (const auto& o : OPTION_TABLE) {
    std::string env = "LH_";
    for (const char* p = o.name; *p; p++) env.push_back((char)toupper((unsigned char)*p));
    const char* e = getenv(env.c_str());
    if (!e || !*e) continue;
    const int64_t v = (int64_t)atoll(e);
    if (v < o.lo || v > o.hi) {  // (an environment default out of range is ignored, loudly: nobody is there to take an error)
      fprintf(stderr, "[lasso-hip] %s=%s is outside [%lld, %lld]: ignored\n", env.c_str(), e, (long long)o.lo, (long long)o.hi);
      continue;
    }
    this->*o.field = v;
    if (o.field == &Options::open_small_min_vars) open_small_min_vars_forced = true;
  }
}

void Ctx::host_stamp(const char* tag) {
  if (!host_trace_on) return;
  host_stamps.emplace_back(tag, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
void Ctx::host_stamps_print() {
  if (!host_trace_on || host_stamps.empty()) return;
  fprintf(stderr, "[host trace]");
  for (size_t i = 0; i < host_stamps.size(); i++)
    fprintf(stderr, " %s +%.1f", host_stamps[i].first, i ? host_stamps[i].second - host_stamps[i - 1].second : 0.0);
  fprintf(stderr, " us\n");
  host_stamps.clear();
}

void Ctx::wait_flag(uint32_t seq) {
  volatile uint32_t* f = flag;
  for (uint64_t spin = 0;; spin++) {
    if (*f == seq) return;
    __builtin_ia32_pause();
    if ((spin & 0xfffff) == 0xfffff) {
      // a failed launch never publishes: ask the runtime instead of spinning forever
      hipError_t e = hipStreamQuery(stream);
      if (e == hipSuccess) {
        if (*f == seq) return;
        throw Error(LH_ERR_DEVICE, "kernel finished without publishing its result");
      }
      if (e != hipErrorNotReady) throw Error(LH_ERR_DEVICE, std::string("stream error: ") + hipGetErrorString(e));
    }
  }
}

hipEvent_t Ctx::live_event() {
  if (!live_pool.empty()) {
    hipEvent_t e = live_pool.back();
    live_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  LH_HIP(hipEventCreate(&e));
  return e;
}
void Ctx::live_resolve(std::vector<ProfRec>& out) {
  LH_HIP(hipStreamSynchronize(stream));
  if (aux_stream) LH_HIP(hipStreamSynchronize(aux_stream));
  for (size_t i = 0; i < live_recs.size();) {
    size_t j = i;
    ProfRec sum = live_recs[i].rec;
    snprintf(sum.name, sizeof sum.name, "%s/batch", live_recs[i].rec.name);
    sum.bytes = sum.muls = sum.items = 0;
    float span = 0;
    for (; j < live_recs.size() && live_recs[j].batch == live_recs[i].batch; j++) {
      LiveRec& r = live_recs[j];
      float ms = 0, end = 0;
      (void)hipEventElapsedTime(&ms, r.e0, r.e1);
      (void)hipEventElapsedTime(&end, live_recs[i].e0, r.e1);  // (the batch's first launch starts first: the others wait for an event behind it)
      r.rec.ms = ms;
      span = std::max(span, end);
      sum.bytes += r.rec.bytes, sum.muls += r.rec.muls, sum.items += r.rec.items;
      out.push_back(r.rec);
    }
    sum.ms = span;
    out.push_back(sum);
    i = j;
  }
  for (LiveRec& r : live_recs) live_pool.push_back(r.e0), live_pool.push_back(r.e1);
  live_recs.clear();
}

void Ctx::aux_streams() {
  if (aux_stream) return;
  // (the LOWEST priority the device offers: what runs here fills the wave slots the ctx's stream leaves idle)
  int prio_least = 0, prio_greatest = 0;
  LH_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  LH_HIP(hipStreamCreateWithPriority(&aux_stream, hipStreamNonBlocking, prio_least));
  LH_HIP(hipEventCreateWithFlags(&aux_ev, hipEventDisableTiming));
}
ScFinishArgs Ctx::finish_for_aux(uint32_t grid, uint32_t seq) {
  ScFinishArgs f{ticket + 16, (uint32_t)(aux_ticket_base + grid - 1), nullptr, flag + 4, seq, nullptr, 0, nullptr};
  if (grid > 1) aux_ticket_base += grid;
  return f;
}
void Ctx::wait_flag_aux(uint32_t seq) {
  volatile uint32_t* f = flag + 4;
  for (uint64_t spin = 0;; spin++) {
    if (*f == seq) return;
    __builtin_ia32_pause();
    if ((spin & 0xfffff) == 0xfffff) {
      hipError_t e = hipStreamQuery(aux_stream ? aux_stream : stream);
      if (e == hipSuccess) {
        if (*f == seq) return;
        throw Error(LH_ERR_DEVICE, "kernel on the aux stream finished without publishing its result");
      }
      if (e != hipErrorNotReady) throw Error(LH_ERR_DEVICE, std::string("aux stream error: ") + hipGetErrorString(e));
    }
  }
}

ScFinishArgs Ctx::finish_for(uint32_t grid, Fr* out_host, uint32_t seq, double stored_bytes) {
  // (the lane buffer serves launches of at most 16 sums per workgroup, one at a time: the kernels of this ctx's stream)
  // Every launch hands over in lanes (measured on 2^24 AND lookups: no launch 58.9-59.0 ms, launches that store >= 2^28 bytes
  // 58.0, >= 2^25 bytes 57.4, every launch 57.1; 2^22 range 15.67 / 15.37 / 15.30; 2^20: within the noise).
  // LH_FIN_LANES_MIN_BYTES: the smallest launch (by the bytes its workgroups store) that does - development A/B; -1: none
  static const double lanes_min = getenv("LH_FIN_LANES_MIN_BYTES") ? atof(getenv("LH_FIN_LANES_MIN_BYTES")) : 0.0;
  const bool lanes = lanes_min >= 0 && stored_bytes >= lanes_min && grid > 1 && (uint64_t)grid * 16 <= FIN_LANE_SUMS;
  ScFinishArgs f{ticket, (uint32_t)(ticket_base + grid - 1), out_host, flag, seq, nullptr, 0, lanes ? fin_lanes : nullptr};
  if (sc_redirect) f.out_host = sc_redirect, f.flag = ticket + 8, f.wide = sc_wide, f.tag = sc_tag;  // sharded round: a device word nobody waits on
  if (grid > 1) ticket_base += grid;  // single-workgroup launches draw no ticket
  return f;
}

// one 16-byte store per chunk (aligned SSE stores are single-copy atomic on every x86-64 with AVX)
static inline void store_chunk(TailChunk* dst, uint32_t a, uint32_t b, uint32_t c_, uint32_t d) {
  _mm_store_si128((__m128i*)dst, _mm_set_epi32((int)d, (int)c_, (int)b, (int)a));
}
static inline void load_chunk(const TailChunk* src, uint32_t w[4]) {
  _mm_store_si128((__m128i*)w, _mm_load_si128((const __m128i*)src));
}

void Ctx::mbox_send(const Fr& r, uint32_t seq) {
  TailMbox* m = mbox();
  store_chunk(&m->c[0], seq, r.l[0], r.l[1], r.l[2]);
  store_chunk(&m->c[1], seq, r.l[3], r.l[4], r.l[5]);
  store_chunk(&m->c[2], seq, r.l[6], r.l[7], 0);
}

void Ctx::mbox_abort() {
  TailMbox* m = mbox();
  for (int j = 0; j < 3; j++) store_chunk(&m->c[j], SC_TAIL_ABORT, 0, 0, 0);
}

void Ctx::gkr_boxes() {
  if (gkr_mbox) return;
  LH_HIP(hipHostMalloc((void**)&gkr_mbox, GKR_MSG_CHUNKS * sizeof(TailChunk), hipHostMallocCoherent | hipHostMallocMapped));
  memset((void*)gkr_mbox, 0, GKR_MSG_CHUNKS * sizeof(TailChunk));
  LH_HIP(hipMalloc((void**)&gkr_relay, (4 + GKR_MSG_CHUNKS) * sizeof(TailChunk)));
  LH_HIP(hipMemset(gkr_relay, 0, (4 + GKR_MSG_CHUNKS) * sizeof(TailChunk)));
}
void Ctx::gkr_send_layer(const Fr* vals, size_t count, uint32_t seq) {
  LH_REQUIRE(3 * count <= GKR_MSG_CHUNKS, LH_ERR_ARG, "resident layers: layer message too long");
  for (size_t i = 0; i < count; i++) {
    const Fr& r = vals[i];
    store_chunk(&gkr_mbox[3 * i + 0], seq, r.l[0], r.l[1], r.l[2]);
    store_chunk(&gkr_mbox[3 * i + 1], seq, r.l[3], r.l[4], r.l[5]);
    store_chunk(&gkr_mbox[3 * i + 2], seq, r.l[6], r.l[7], 0);
  }
}
void Ctx::gkr_abort() {
  mbox_abort();
  if (gkr_mbox) store_chunk(&gkr_mbox[0], SC_TAIL_ABORT, 0, 0, 0);
}
void Ctx::gkr_resync() {
  if (gkr_relay) LH_HIP(hipMemsetAsync(gkr_relay, 0, (4 + GKR_MSG_CHUNKS) * sizeof(TailChunk), stream));
  if (gkr_mbox) memset((void*)gkr_mbox, 0, GKR_MSG_CHUNKS * sizeof(TailChunk));
  k_sc_tail_resync(*this);
}

void Ctx::wait_chunks(const TailChunk* chunks, size_t count, uint32_t seq, Fr* out) {
  (void)wait_chunks_or(chunks, count, seq, seq, out);
}
bool Ctx::wait_chunks_or(const TailChunk* chunks, size_t count, uint32_t seq, uint32_t alt, Fr* out) {
  alignas(16) uint32_t w[4];
  size_t have = 0;  // chunks 0..have-1 carry `seq` and are copied out
  bool gone = false;
  for (uint64_t spin = 0;; spin++) {
    while (have < count) {
      load_chunk(chunks + have, w);
      if (have == 0 && alt != seq && w[0] == alt) return false;
      if (w[0] != seq) break;
      Fr& f = out[have / 3];
      const size_t j = have % 3;
      f.l[3 * j] = w[1], f.l[3 * j + 1] = w[2];
      if (j < 2) f.l[3 * j + 2] = w[3];
      have++;
    }
    if (have == count) return true;
    __builtin_ia32_pause();
    if ((spin & 0xfffff) == 0xfffff) {
      hipError_t e = hipStreamQuery(stream);
      if (e == hipSuccess) {
        if (gone) throw Error(LH_ERR_DEVICE, "kernel finished without publishing its result");
        gone = true;  // one more look at the chunks: they may have landed between the last look and the query
        spin = 0xffffe;
        continue;
      }
      if (e != hipErrorNotReady) throw Error(LH_ERR_DEVICE, std::string("stream error: ") + hipGetErrorString(e));
    }
  }
}

void* Ctx::pin(size_t bytes) {
  if (bytes > pinned_bytes) {
    if (pinned) (void)hipHostFree(pinned);
    size_t want = bytes < 65536 ? 65536 : bytes;
    LH_HIP(hipHostMalloc(&pinned, want, hipHostMallocCoherent | hipHostMallocMapped));
    pinned_bytes = want;
  }
  return pinned;
}

void Ctx::d2h(void* dst, const void* d_src, size_t bytes) {
  if (!bytes) return;
  if (bytes > ((size_t)8 << 20)) {  // bulk: let the runtime stream it
    LH_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream));
    sync();
    return;
  }
  if (bytes > stage_bytes) {
    if (stage) (void)hipHostFree(stage);
    size_t want = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes;
    LH_HIP(hipHostMalloc(&stage, want, hipHostMallocCoherent | hipHostMallocMapped));
    stage_bytes = want;
  }
  LH_HIP(hipMemcpyAsync(stage, d_src, bytes, hipMemcpyDeviceToHost, stream));
  sync();
  memcpy(dst, stage, bytes);
}

// ------------------------------------------------------------------ HostWorker
struct HostWorker::Impl {
  std::mutex mu;
  std::condition_variable cv, idle_cv;
  std::vector<std::function<void()>> queue;
  size_t head = 0;
  bool busy = false, stop = false;
  std::thread th;
  void loop() {
    for (;;) {
      std::function<void()> fn;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || head < queue.size(); });
        if (head >= queue.size()) return;  // (stop, and nothing left)
        fn = std::move(queue[head++]);
        if (head == queue.size()) queue.clear(), head = 0;
        busy = true;
      }
      try {
        fn();
      } catch (...) {  // (tasks report their own errors; nothing may escape a thread)
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        busy = false;
      }
      idle_cv.notify_all();
    }
  }
};
HostWorker::HostWorker() : impl_(new Impl()) { impl_->th = std::thread([this] { impl_->loop(); }); }
HostWorker::~HostWorker() {
  {
    std::lock_guard<std::mutex> lk(impl_->mu);
    impl_->stop = true;
  }
  impl_->cv.notify_all();
  if (impl_->th.joinable()) impl_->th.join();
  delete impl_;
}
void HostWorker::submit(std::function<void()> fn) {
  {
    std::lock_guard<std::mutex> lk(impl_->mu);
    impl_->queue.push_back(std::move(fn));
  }
  impl_->cv.notify_one();
}
void HostWorker::wait() {
  std::unique_lock<std::mutex> lk(impl_->mu);
  impl_->idle_cv.wait(lk, [&] { return !impl_->busy && impl_->head >= impl_->queue.size(); });
}

void Ctx::phase_times_resolve() {
  if (!phase_ev_pending) return;
  phase_ev_pending = false;
  // ev[0] = start of the prove, ev[k + 1] = end of phase k (k = 0..6); [7] stays 0, [8] (total) is the host's wall clock
  if (!phase_ev[0]) return;
  LH_HIP(hipEventSynchronize(phase_ev[7]));
  for (int k = 0; k < 7; k++) {
    float ms = 0;
    LH_HIP(hipEventElapsedTime(&ms, phase_ev[k], phase_ev[k + 1]));
    lasso_ms[k] = (double)ms;
  }
}

// ------------------------------------------------------------------ host worker pool
// Every parallel_for publishes ONE immutable job object (function, item count, its own claim and completion
// counters).  A worker copies the shared_ptr under the mutex and claims indices only from that job's counter, so a
// worker that wakes late works on its own (already finished) job and can never draw an index of the next generation.
namespace {
struct HostJob {
  std::function<void(size_t)> fn;
  size_t n = 0;
  std::atomic<size_t> next{0}, done{0};
  void run() {
    size_t i;
    while ((i = next.fetch_add(1, std::memory_order_relaxed)) < n) {
      fn(i);
      done.fetch_add(1, std::memory_order_release);
    }
  }
};
struct HostPool {
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<uint64_t> gen{0};
  uint64_t spin_gen = 0;       // bumped by prewake(): `spin_want` sleeping workers wake up and poll `gen` until `spin_until`
  int spin_want = 0;
  std::chrono::steady_clock::time_point spin_until;
  std::atomic<int> spinning{0};
  std::shared_ptr<HostJob> job;
  unsigned workers = 0;
  // the CPUs this process may run on (taskset, a cgroup, Context.bind_host: fewer than the machine's)
  static unsigned usable_cpus() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) return (unsigned)CPU_COUNT(&set);
    const unsigned hw = std::thread::hardware_concurrency();
    return hw ? hw : 1;
  }
  HostPool() {
    unsigned hw = usable_cpus();
    // (an MSM batch hands ~25-40 window combines of ~70 us each to this pool at the end of a commit or an opening - the
    // GPU idles meanwhile: one round of them, not three, where the host has the cores; LH_HOST_THREADS overrides)
    unsigned cap = 47;
    if (const char* e = getenv("LH_HOST_THREADS")) cap = (unsigned)std::max(0, atoi(e) - 1);
    workers = hw > 1 ? std::min(cap, hw - 1) : 0;
    for (unsigned i = 0; i < workers; i++) std::thread([this] { worker(); }).detach();
  }
  void worker() {
    uint64_t seen = 0, seen_spin = 0;
    bool was_spinning = false;
    for (;;) {
      std::shared_ptr<HostJob> mine;
      bool spin = false;
      std::chrono::steady_clock::time_point until;
      {
        std::unique_lock<std::mutex> lk(mu);
        // (`spinning` only changes under the lock: a poster that counts the pollers under the lock counts exactly the
        // workers that will look at `gen` again before they sleep)
        if (was_spinning) spinning--, was_spinning = false;
        cv.wait(lk, [&] { return gen.load(std::memory_order_relaxed) != seen || (spin_gen != seen_spin && spin_want > 0); });
        if (gen.load(std::memory_order_relaxed) != seen) {
          seen = gen.load(std::memory_order_relaxed);
          seen_spin = spin_gen;
          mine = job;
        } else {  // woken ahead of a job that is about to come: poll for it instead of sleeping through its arrival
          seen_spin = spin_gen;
          spin_want--;
          spin = true;
          until = spin_until;
          spinning++, was_spinning = true;
        }
      }
      if (spin) {
        while (gen.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
        continue;  // (takes the job under the lock if one came, sleeps again otherwise)
      }
      if (mine) mine->run();
    }
  }
  // a parallel_for of about `count` items will be posted within `us` microseconds: have workers awake and polling by then
  // (waking a sleeping thread costs the poster ~50 us of the ~70 us an item takes)
  void prewake(size_t count, unsigned us) {
    if (!workers || count <= 1) return;
    // (polling threads burn a core each: at most an eighth of the host's - eight ranks may share it, one process per GPU)
    static const size_t spin_cap = std::max<size_t>(1, usable_cpus() / 8);
    int want;
    {
      std::lock_guard<std::mutex> lk(mu);
      want = (int)std::min<size_t>(std::min<size_t>(count - 1, workers), spin_cap) - spinning;
      if (want <= 0) return;
      spin_gen++;
      spin_want = want;
      spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(us);
    }
    if (want >= (int)workers) cv.notify_all();
    else
      for (int i = 0; i < want; i++) cv.notify_one();
  }
  void parallel_for(size_t count, const std::function<void(size_t)>& f) {
    auto j = std::make_shared<HostJob>();
    j->fn = f;  // a copy: the job outlives the caller's frame for workers that wake late
    j->n = count;
    bool enough_awake;
    {
      std::lock_guard<std::mutex> lk(mu);
      enough_awake = (size_t)spinning + 1 >= count;  // (decided under the lock: see worker())
      job = j;
      gen.fetch_add(1, std::memory_order_release);
      spin_want = 0;
    }
    if (!enough_awake) cv.notify_all();  // (polling workers see `gen` move by themselves)
    j->run();
    while (j->done.load(std::memory_order_acquire) != count) __builtin_ia32_pause();
    {
      std::lock_guard<std::mutex> lk(mu);
      if (job == j) job.reset();
    }
  }
};
}  // namespace

static HostPool& host_pool() {
  static HostPool* pool = new HostPool();  // never destroyed: workers sleep on the condition variable
  return *pool;
}
void host_parallel_for(size_t n, const std::function<void(size_t)>& fn) {
  if (n <= 1) {
    for (size_t i = 0; i < n; i++) fn(i);
    return;
  }
  host_pool().parallel_for(n, fn);
}
void host_parallel_prewake(size_t n, unsigned us) { host_pool().prewake(n, us); }

}  // namespace lh
