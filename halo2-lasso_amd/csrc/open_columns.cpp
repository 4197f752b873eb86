// The challenge-free half of the opening's column route and its run on the helper ctx (open_columns.hpp).
#include <algorithm>
#include <functional>
#include <chrono>
#include <memory>
#include <thread>
#include "open_columns.hpp"

namespace lh {

static inline uint32_t bits_of_u32(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

// ---- which quotient levels go column by column: the largest always; the second largest too when few columns take
// part.  After d folds the remainder is sum_s w_s g'[s 2^(n-d) + .] over the 2^d settings s of the top d index bits
// (w_s = the product of x_j or 1 - x_j over those bits), so the quotient of level n-1-d is the same kind of sum of
// 2^d differences of sub-columns per column: 2^d times the passes of the top level, against ~15 windows.
void column_shape(Ctx& c, const std::vector<SmallPoly>& cols, const std::vector<char>& zero, size_t n,
                         size_t num_vars, size_t cut, ColumnPlan& plan) {
  const size_t K = cols.size(), half_top = n >> 1;
  std::vector<uint32_t>& ors = plan.ors;
  ors.assign(K, 0);
  std::map<size_t, std::vector<size_t>> by_len;  // one OR pass per column length (normally none: the widths are known)
  for (size_t k = 0; k < K; k++) {
    // (every column that can reach past the low half of a sub-column at either depth needs its width)
    if (cols[k].len <= (half_top >> 1) || zero[k]) continue;
    if (cols[k].bits) ors[k] = cols[k].bits >= 32 ? 0xffffffffu : (1u << cols[k].bits) - 1u;  // known bound
    else by_len[cols[k].len].push_back(k);
  }
  for (const auto& grp : by_len) {
    std::vector<const uint32_t*> ptrs;
    for (size_t k : grp.second) ptrs.push_back(cols[k].ptr);
    std::vector<uint32_t> o(ptrs.size(), 0);
    k_or_u32(c, ptrs.data(), ptrs.size(), grp.first, o.data());
    for (size_t f = 0; f < ptrs.size(); f++) ors[grp.second[f]] = o[f];
  }
  size_t passes = 0, narrow_cols = 0;
  for (size_t k = 0; k < K; k++) {
    if (cols[k].len <= half_top || zero[k] || !ors[k]) continue;
    const uint32_t b1 = bits_of_u32(ors[k]) + 1;
    if (b1 > 32) passes += 2;
    else if (b1 <= MSM_PACK_MAX_BITS - 4) narrow_cols++;
    else passes += 1;
  }
  passes += (narrow_cols + 1) / 2;
  const int forced_depth = (int)c.opt.open_small_depth;  // 1 or 2 levels column by column, whatever the shape
  plan.depth = 1;
  if (num_vars >= cut + 3 && (forced_depth ? forced_depth >= 2 : 2 * passes <= 10)) plan.depth = 2;
}

// the MSM jobs of the column-wise levels (difference columns, limbs, packed pairs, the levels' base sums); temporaries
// from c's arena, kernels on c's stream
void column_jobs(Ctx& c, const Srs& srs, const std::vector<SmallPoly>& cols, const std::vector<char>& zero,
                        size_t num_vars, size_t lsh, bool sharded, const std::function<const G1Affine*(size_t)>& level_bases,
                        ColumnPlan& plan) {
  const size_t depth = plan.depth;
  const std::vector<uint32_t>& ors = plan.ors;
  std::vector<MsmJob>& jobs = plan.jobs;
  plan.levels.assign(depth, ColLevel());
  size_t num_seconds = 0;
  for (size_t d = 0; d < depth; d++) num_seconds += cols.size() << d;
  plan.seconds.assign(num_seconds, HG1());
  size_t next_second = 0;
  for (size_t d = 0; d < depth; d++) {
    ColLevel& cl = plan.levels[d];
    cl.level = num_vars - 1 - d;
    const size_t half = (size_t)1 << (cl.level - lsh);
    const G1Affine* bases = level_bases(cl.level);
    struct Narrow {
      uint32_t bits;  // of the shifted difference
      uint32_t* col;
      size_t k, sidx;
    };
    std::vector<Narrow> narrow;
    for (size_t sidx = 0; sidx < ((size_t)1 << d); sidx++) {
      const size_t off_idx = sidx << (cl.level + 1 - lsh);
      for (size_t k = 0; k < cols.size(); k++) {
        const SmallPoly& sp = cols[k];
        if (zero[k] || sp.len <= off_idx) continue;
        const uint32_t* sub = sp.ptr + off_idx;
        const size_t sub_len = std::min(sp.len - off_idx, half << 1);
        if (sub_len <= half) {  // only the low half is populated: hi - lo = -lo, no offset
          jobs.push_back(MsmJob{sub, true, bases, sub_len});
          if (sp.bits) jobs.back().known_bits = sp.bits;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, -1});
          continue;
        }
        const uint32_t b = bits_of_u32(ors[k]);
        if (!b) continue;  // an all-zero column
        const uint64_t off = (uint64_t)1 << b;
        cl.offsets.push_back(ColOffset{k, sidx, off});
        cl.need_sum = true;
        if (b + 1 > 32) {  // 33-bit shifted differences: a 16-bit limb and a 17-bit limb
          uint32_t* lo = c.arena.alloc_n<uint32_t>(half);
          uint32_t* hi = c.arena.alloc_n<uint32_t>(half);
          k_delta_u32(c, sub, sub_len, half, off, lo, hi);
          jobs.push_back(MsmJob{lo, true, bases, half});
          jobs.back().known_bits = 16;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, 1});
          jobs.push_back(MsmJob{hi, true, bases, half});
          jobs.back().known_bits = b + 1 - 16;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, 65536});
          continue;
        }
        uint32_t* dcol = c.arena.alloc_n<uint32_t>(half);
        k_delta_u32(c, sub, sub_len, half, off, dcol, nullptr);
        if (b + 1 <= MSM_PACK_MAX_BITS - 4) {
          narrow.push_back(Narrow{b + 1, dcol, k, sidx});
        } else {
          jobs.push_back(MsmJob{dcol, true, bases, half});
          jobs.back().known_bits = b + 1;
          cl.terms.push_back(ColTerm{jobs.size() - 1, false, k, sidx, 1});
        }
      }
    }
    // narrow columns two by two (narrowest first) while the packed value stays within MSM_PACK_MAX_BITS
    std::stable_sort(narrow.begin(), narrow.end(), [](const Narrow& x, const Narrow& y) { return x.bits < y.bits; });
    for (size_t i = 0; i < narrow.size(); i++) {
      const Narrow& x = narrow[i];
      const uint32_t shift = std::max(x.bits, 4u);
      // (a packed pair saves one pass over `half` points and reduces a bucket set indexed by the packed value - ~5 curve
      // additions' worth per bucket: on a rank's shard of a level the second can outweigh the first)
      if (i + 1 < narrow.size() && shift + narrow[i + 1].bits <= MSM_PACK_MAX_BITS &&
          half >= ((size_t)MSM_PACK_MIN_POINTS_PER_BUCKET << (shift + narrow[i + 1].bits))) {
        const Narrow& y = narrow[i + 1];
        uint32_t* packed = c.arena.alloc_n<uint32_t>(half);
        k_pack_u32(c, x.col, y.col, shift, half, packed);
        MsmJob jb{packed, true, bases, half};
        jb.pack_shift = shift;
        jb.known_bits = shift + y.bits;
        jb.out_second = (G1Affine*)&plan.seconds[next_second++];
        jobs.push_back(jb);
        cl.terms.push_back(ColTerm{jobs.size() - 1, false, x.k, x.sidx, 1});
        cl.terms.push_back(ColTerm{jobs.size() - 1, true, y.k, y.sidx, 1});
        i++;
      } else {
        jobs.push_back(MsmJob{x.col, true, bases, half});
        jobs.back().known_bits = x.bits;
        cl.terms.push_back(ColTerm{jobs.size() - 1, false, x.k, x.sidx, 1});
      }
    }
    // the level's base sum (an MSM with all-one scalars, once per SRS and level)
    if (cl.need_sum) {
      std::lock_guard<std::mutex> lock(srs_cache_mu);
      std::map<size_t, HG1>& sums = sharded ? srs.shard_level_sums : srs.level_sums;  // (sharded: of this rank's share)
      auto it = sums.find(cl.level);
      if (it != sums.end()) {
        cl.base_sum = it->second;
      } else {
        uint32_t* ones = c.arena.alloc_n<uint32_t>(half);
        k_fill_u32(c, ones, 1u, half);
        jobs.push_back(MsmJob{ones, true, bases, half});
        jobs.back().known_bits = 1;
        cl.sum_job = jobs.size() - 1;
      }
    }
  }
}
// after the jobs ran: the levels' base sums that were computed go into the SRS's cache
void column_sums_store(const Srs& srs, bool sharded, ColumnPlan& plan, const HG1* out) {
  for (ColLevel& cl : plan.levels)
    if (cl.sum_job != (size_t)-1) {
      cl.base_sum = out[cl.sum_job];
      std::lock_guard<std::mutex> lock(srs_cache_mu);
      (sharded ? srs.shard_level_sums : srs.level_sums)[cl.level] = cl.base_sum;
    }
}

// does an opening over these columns take the column route (mkzg_open's gating, on the local sizes)
bool column_route_on(const Ctx& c, const std::vector<SmallPoly>& cols, const std::vector<char>& zero, size_t num_vars,
                            size_t lsh, size_t n, bool sharded, size_t cut) {
  if (num_vars - lsh < (size_t)c.opt.open_small_min_vars) {
    // below the general threshold the route still pays when only a few columns take part (the range check: two dim
    // and two read_ts columns - 2^20 lookups 11.7 -> 11.0 ms; the AND table's twelve columns lose there)
    size_t full = 0;
    for (size_t k = 0; k < cols.size(); k++) full += cols[k].len > (n >> 1) && !zero[k];
    if (!(num_vars - lsh >= 17 && full <= 4 && !c.opt.open_small_min_vars_forced)) return false;
  }
  if (num_vars < 2 || cols.empty()) return false;
  if (sharded && num_vars < cut + 2) return false;  // (the column-wise levels must be sharded ones)
  return true;
}

void open_precommit_cancel(Ctx& c) {
  if (!c.precommit) return;
  delete (OpenPrecommit*)c.precommit;  // (joins)
  c.precommit = nullptr;
}
// the columns of a batch opening whose polys are all small-valued columns: `used` polys, a linear column's coefficient
// handed to the columns it combines, the same column under two names merged.  coef == nullptr: the structure alone
// (every column that survives gets coefficient one)
bool small_open_columns(const SmallPoly* small, size_t num_polys, const lh_evaluation* evals, size_t num_evals,
                               size_t n, const HFr* coef_in, SmallOpen& so) {
  for (size_t i = 0; i < num_evals; i++)
    if (evals[i].poly >= num_polys || !small[evals[i].poly].ptr) return false;
  std::vector<HFr> coef(num_polys, HFr::zero());
  std::vector<char> used(num_polys, 0);
  for (size_t i = 0; i < num_evals; i++) {
    coef[evals[i].poly] = coef_in ? coef_in[evals[i].poly] : HFr::one();
    used[evals[i].poly] = 1;
  }
  // a column that is a linear combination of other opened columns hands its coefficient over to them
  for (size_t pi = 0; pi < num_polys; pi++) {
    const SmallLinear* lin = small[pi].linear;
    if (!used[pi] || !lin) continue;
    bool ok = !lin->poly.empty() && lin->poly.size() == lin->coeff.size();
    for (size_t k = 0; k < lin->poly.size() && ok; k++) ok = lin->poly[k] < num_polys && lin->poly[k] != pi && used[lin->poly[k]];
    if (!ok) continue;
    if (coef_in)
      for (size_t k = 0; k < lin->poly.size(); k++) coef[lin->poly[k]] += coef[pi] * lin->coeff[k];
    coef[pi] = HFr::zero();
    used[pi] = 0;
  }
  for (size_t pi = 0; pi < num_polys; pi++) {
    if (!used[pi]) continue;
    const SmallPoly sp{small[pi].ptr, std::min(small[pi].len, n), small[pi].bits};
    size_t k = 0;  // the same column under two names (Lasso's E = dim for an identity subtable): one coefficient
    while (k < so.cols.size() && !(so.cols[k].ptr == sp.ptr && so.cols[k].len == sp.len)) k++;
    if (k < so.cols.size()) {
      if (coef_in) so.coef[k] += coef[pi];
      so.cols[k].bits = so.cols[k].bits && sp.bits ? std::max(so.cols[k].bits, sp.bits) : 0;
    } else {
      so.cols.push_back(sp), so.coef.push_back(coef[pi]);
    }
  }
  return true;
}
void open_precommit_start(Ctx& c, const Srs& srs, size_t num_vars, const SmallPoly* small, size_t num_polys,
                          const lh_evaluation* evals, size_t num_evals) {
  open_precommit_cancel(c);
  // (the option is the smallest proof that does it; default 1: every proof that takes the column route - 2^17..2^19 range
  // lookups gain too: 7.0 -> 6.4-6.9, 8.1-9.1 -> 7.3-7.7, 9.7-9.9 -> 9.3-9.4 ms)
  if (c.opt.open_precommit <= 0 || (int64_t)num_vars < c.opt.open_precommit || !small || num_vars > srs.num_vars) return;
  // inside a sharded proof the columns are this rank's shards and the column-wise levels are committed against this rank's
  // share of the levels' bases (mkzg_open's geometry): nothing in this half of the route needs a peer
  const Shard sh(c);
  const bool sharded = sh.sharded(num_vars);
  const size_t lsh = sharded ? sh.rho : 0, cut = sharded ? sh.j + sh.rho : 0;
  const size_t n = (size_t)1 << (num_vars - lsh);
  SmallOpen so;
  if (!small_open_columns(small, num_polys, evals, num_evals, n, nullptr, so)) return;
  std::vector<char> zero(so.cols.size(), 0);
  if (!column_route_on(c, so.cols, zero, num_vars, lsh, n, sharded, cut)) return;
  // the bases of the (at most two) column-wise levels, resolved here: a rank's share of a level is made on first use by a
  // ctx that knows the shard geometry - this one
  std::vector<const G1Affine*> bases_of(num_vars + 1, nullptr);
  for (size_t d = 0; d < 2 && d + 1 <= num_vars; d++) {
    const size_t lvl = num_vars - 1 - d;
    if (sharded && lvl < cut) break;
    bases_of[lvl] = sharded ? srs_shard_level(c, srs, lvl) : srs.eq(lvl);
  }
  Ctx& h = ctx_helper(c);
  h.opt = c.opt;
  h.prof = c.prof;
  h.live = c.live;  // (live records of the helper's accumulation launches are read with the owner's: capi.cpp)
  h.prof_recs.clear();
  OpenPrecommit* pc = new OpenPrecommit();
  pc->srs = &srs, pc->num_vars = num_vars, pc->cols = so.cols, pc->zero = zero;
  c.precommit = pc;
  // the witness columns are written by kernels queued on this ctx's stream and read by the helper's: an event between the
  // streams (not a host sync), and the helper's long-lived host thread (not a thread per proof)
  if (!c.handoff_ev) LH_HIP(hipEventCreateWithFlags(&c.handoff_ev, hipEventDisableTiming));
  LH_HIP(hipEventRecord(c.handoff_ev, c.stream));
  hipEvent_t handoff = c.handoff_ev;
  const int device = c.device;
  if (!h.worker) h.worker = new HostWorker();
  pc->worker = h.worker;
  h.worker->submit([pc, &h, &srs, num_vars, n, lsh, cut, sharded, bases_of, device, handoff] {
    try {
      LH_HIP(hipSetDevice(device));
      LH_HIP(hipStreamWaitEvent(h.stream, handoff, 0));
      ArenaScope scope(h.arena);
      column_shape(h, pc->cols, pc->zero, n, num_vars, cut, pc->plan);
      column_jobs(h, srs, pc->cols, pc->zero, num_vars, lsh, sharded,
                  [&bases_of](size_t lvl) {
                    LH_REQUIRE(lvl < bases_of.size() && bases_of[lvl], LH_ERR_ARG, "open precommit: level without bases");
                    return bases_of[lvl];
                  },
                  pc->plan);
      pc->out.resize(pc->plan.jobs.size());
      if (!pc->plan.jobs.empty()) msm_batch(h, pc->plan.jobs.data(), pc->plan.jobs.size(), (G1Affine*)pc->out.data());
      h.sync();
      column_sums_store(srs, sharded, pc->plan, pc->out.data());
      pc->ok = true;
    } catch (const std::exception& e) {
      pc->err = e.what();
    } catch (...) {
      pc->err = "unknown error";
    }
  });
  if (c.prof) pc->join();  // a profiled prove keeps its kernels one at a time (the records are merged when taken)
}
// the precommitted plan if it is the plan this opening would build (same SRS, columns, zero pattern, widths, depth)
std::unique_ptr<OpenPrecommit> open_precommit_take(Ctx& c, const Srs& srs, size_t num_vars,
                                                          const std::vector<SmallPoly>& cols, const std::vector<char>& zero,
                                                          const ColumnPlan& own) {
  std::unique_ptr<OpenPrecommit> pc((OpenPrecommit*)c.precommit);
  c.precommit = nullptr;
  if (!pc) return nullptr;
  pc->join();
  if (c.prof && c.helper) {
    c.prof_recs.insert(c.prof_recs.end(), c.helper->prof_recs.begin(), c.helper->prof_recs.end());
    c.helper->prof_recs.clear();
  }
  bool same = pc->ok && pc->srs == &srs && pc->num_vars == num_vars && pc->cols.size() == cols.size() && pc->zero == zero &&
              pc->plan.depth == own.depth && pc->plan.ors == own.ors;
  for (size_t k = 0; same && k < cols.size(); k++)
    same = pc->cols[k].ptr == cols[k].ptr && pc->cols[k].len == cols[k].len && pc->cols[k].bits == cols[k].bits;
  if (!same) return nullptr;
  return pc;
}


}  // namespace lh
