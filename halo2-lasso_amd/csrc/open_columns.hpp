// The column route of a batch opening over small-valued columns (pcs/multilinear.rs:72-107 by linearity): which quotient
// levels are committed column by column, the MSM jobs that do it (the challenge-free half - open_columns.cpp), and the plan
// committed ahead on a helper ctx (Options::open_precommit).  Shared by mkzg.cpp (the opening) and open_columns.cpp.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include "host.hpp"

namespace lh {

extern std::mutex srs_cache_mu;  // guards an Srs's lazily built members (level sums, shard levels, window tables)

// ------------------------------------------------------------------ the challenge-free half of the column route
// Which columns take part, how wide their differences are, how many levels go column by column and the MSM jobs that
// commit them depend on the witness columns alone; only the linear combination of the jobs' results uses the batch
// opening's coefficients and the fold weights.  mkzg_open builds this plan itself - or finds it already committed by
// open_precommit_start (a helper ctx on its own stream and host thread, beside the GKR phase).
struct ColTerm {  // result of job `job` (or its second output) times coef[k] * w(sidx) * factor goes into the commitment
  size_t job;
  bool second;
  size_t k, sidx;
  int factor;  // -1: only the low half is populated (hi - lo = -lo); 1; 65536: the high limb of a 33-bit difference
};
struct ColOffset {  // coef[k] * w(sidx) * off: times the level's base sum, subtracted
  size_t k, sidx;
  uint64_t off;
};
struct ColLevel {
  size_t level = 0;  // quotient level (number of variables)
  std::vector<ColTerm> terms;
  std::vector<ColOffset> offsets;
  bool need_sum = false;
  size_t sum_job = (size_t)-1;  // the all-ones MSM when the level's base sum is not cached yet
  HG1 base_sum;
};
struct ColumnPlan {
  std::vector<uint32_t> ors;  // per column: OR of its entries (a bound when the caller knows the width)
  size_t depth = 0;           // quotient levels that go column by column
  std::vector<MsmJob> jobs;
  std::vector<ColLevel> levels;
  std::vector<HG1> seconds;   // second outputs of packed jobs (MsmJob::out_second points in here: sized before the jobs)
};
// ---- the plan committed ahead (Options::open_precommit): the helper ctx builds the same plan from the same columns and
// runs its jobs on its own stream, driven by its own host thread, while the ctx goes through the sum-checks.  mkzg_open
// takes the results only if columns, widths and depth are what it arrives at itself; otherwise it commits as before.
struct OpenPrecommit {
  HostWorker* worker = nullptr;  // the helper ctx's host thread, busy with this plan until wait() returns
  const Srs* srs = nullptr;
  size_t num_vars = 0;
  std::vector<SmallPoly> cols;
  std::vector<char> zero;
  ColumnPlan plan;
  std::vector<HG1> out;
  bool ok = false;
  std::string err;
  void join() {
    if (worker) worker->wait();
    worker = nullptr;
  }
  ~OpenPrecommit() { join(); }
};

void column_shape(Ctx& c, const std::vector<SmallPoly>& cols, const std::vector<char>& zero, size_t n, size_t num_vars, size_t cut,
                  ColumnPlan& plan);
void column_jobs(Ctx& c, const Srs& srs, const std::vector<SmallPoly>& cols, const std::vector<char>& zero, size_t num_vars,
                 size_t lsh, bool sharded, const std::function<const G1Affine*(size_t)>& level_bases, ColumnPlan& plan);
void column_sums_store(const Srs& srs, bool sharded, ColumnPlan& plan, const HG1* out);
bool column_route_on(const Ctx& c, const std::vector<SmallPoly>& cols, const std::vector<char>& zero, size_t num_vars, size_t lsh,
                     size_t n, bool sharded, size_t cut);
bool small_open_columns(const SmallPoly* small, size_t num_polys, const lh_evaluation* evals, size_t num_evals, size_t n,
                        const HFr* coef_in, SmallOpen& so);
std::unique_ptr<OpenPrecommit> open_precommit_take(Ctx& c, const Srs& srs, size_t num_vars, const std::vector<SmallPoly>& cols,
                                                   const std::vector<char>& zero, const ColumnPlan& own);

}  // namespace lh
