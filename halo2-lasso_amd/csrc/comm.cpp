// Communicator of a proof sharded over several GPUs (the reference is single-process, nothing to cite).
//
// Two collectives are all the sharded prover needs:
//   all_gather_dev   device buffers, enqueued on the ctx's stream - per-round partial sums, residual tables when the
//                    shard bits reach bit 0, tree levels / quotient remainders at the replication point, the u32
//                    lookup columns of the witness phase;
//   all_gather_host  a few hundred bytes that live on the host anyway (partial commitments after the host's window
//                    combine, partial evaluations).
// Backends:
//   RCCL (lh_ctx_set_comm_rccl) - one ncclComm per ctx, created from a unique id the caller distributes (the host
//       side does that over its own control plane: torch.distributed's store / broadcast in halo2-lasso_amd/dist.py).
//       ncclAllGather runs on the ctx's stream, so a sharded sum-check round is [round kernel -> all-gather -> sum and
//       publish kernel] without a host round trip in between; host-side gathers are staged through a device buffer.
//       librccl is loaded with dlopen at the first use (no link-time dependency: a single-GPU host never loads it; a
//       process that already runs torch's RCCL gets that same instance through the shared soname).
//   callbacks (lh_ctx_set_comm) - a caller-supplied host all-gather (gloo in the CPU / one-GPU tests) and optionally a
//       device all-gather; without the latter device gathers are staged through the host.
#include <dlfcn.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <rccl/rccl.h>  // types and enums only: every function is resolved with dlsym
#include <mutex>
#include "host.hpp"

namespace lh {

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

const RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    // an instance the process already runs (torch ships its own librccl.so) is reused; otherwise the ROCm one is loaded.
    // RTLD_LOCAL: a second copy loaded later by someone else must not bind to this one's symbols.
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (api.lib) break;
    }
    for (const char* n : names) {
      if (api.lib) break;
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!api.lib) {
      err = std::string("RCCL is not loadable (librccl.so.1): ") + (dlerror() ? dlerror() : "");
      return;
    }
    auto sym = [&](const char* s) {
      void* p = dlsym(api.lib, s);
      if (!p && err.empty()) err = std::string("RCCL symbol missing: ") + s;
      return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  if (!err.empty()) throw Error(LH_ERR_DEVICE, err);
  return api;
}

void rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return;
  const RcclApi& api = rccl();
  throw Error(LH_ERR_DEVICE, std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(r) : "RCCL error"));
}

void* comm_device_stage(Ctx& c, size_t bytes) {
  if (bytes > c.comm_stage_bytes) {
    if (c.comm_stage) {
      c.sync();
      (void)hipFree(c.comm_stage);
    }
    size_t want = bytes < ((size_t)1 << 16) ? ((size_t)1 << 16) : bytes;
    LH_HIP(hipMalloc(&c.comm_stage, want));
    c.comm_stage_bytes = want;
  }
  return c.comm_stage;
}
}  // namespace

void rccl_unique_id(uint8_t out[LH_RCCL_UNIQUE_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == LH_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out, &id, sizeof(id));
}

void comm_detach(Ctx& c) {
  if (c.rccl_comm) {
    (void)hipStreamSynchronize(c.stream);
    (void)rccl().CommDestroy((ncclComm_t)c.rccl_comm);
    c.rccl_comm = nullptr;
  }
  c.has_comm = false;
  c.comm_loopback = false;
  c.comm = lh_comm{0, 1, nullptr, nullptr, nullptr};
  c.shard_bit = 0;
}

void comm_attach_rccl(Ctx& c, int rank, int size, const uint8_t id_bytes[LH_RCCL_UNIQUE_ID_BYTES], size_t shard_bit) {
  comm_detach(c);
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm = nullptr;
  rccl_check(rccl().CommInitRank(&comm, size, id, rank), "ncclCommInitRank");
  c.rccl_comm = comm;
  c.comm = lh_comm{rank, size, nullptr, nullptr, nullptr};
  c.has_comm = true;
  c.shard_bit = shard_bit;
  comm_probe_host_recv(c);
}

// One-time self-test of what comm_round = 1 relies on (every rank runs it at attach, it is a collective): does this
// fabric's ncclAllReduce accept PINNED HOST memory as its receive buffer and deliver every u64 lane there?  Each rank
// contributes lanes tagged like a round's (comm_sum_lanes), the collective writes into the ctx's pinned lane block, and
// the host waits - bounded - for lanes that read size * tag.  The ranks then agree on the verdict through an all-reduce
// into DEVICE memory (ncclMin over a 0 / 1 word), so that every rank takes the same variant: Ctx::comm_host_recv_ok false
// makes comm_round = 1 behave as 2 (device receive buffer + copy), with a note on stderr.  LH_COMM_PROBE=0 skips the
// probe (the variant is then taken on trust), LH_COMM_PROBE_FAIL=1 makes it fail (tests).
void comm_probe_host_recv(Ctx& c) {
  c.comm_host_recv_ok = true;
  if (!c.rccl_comm) return;
  static const bool skip = [] { const char* e = getenv("LH_COMM_PROBE"); return e && atoi(e) == 0; }();
  static const bool force_fail = [] { const char* e = getenv("LH_COMM_PROBE_FAIL"); return e && atoi(e) != 0; }();
  if (skip) return;
  const RcclApi& api = rccl();
  const size_t R = (size_t)c.comm.size, n = 16;
  if (!c.lanes_host) {
    LH_HIP(hipHostMalloc((void**)&c.lanes_host, 128 * sizeof(uint64_t), hipHostMallocCoherent | hipHostMallocMapped));
    memset(c.lanes_host, 0, 128 * sizeof(uint64_t));
  }
  ArenaScope scope(c.arena);
  uint64_t* d = c.arena.alloc_n<uint64_t>(2 * n);
  const uint64_t tag = 0x5a5;  // (R * tag < 2^24 for R <= 256; a lane is a limb | tag << 40, as in a round)
  uint64_t h[n];
  for (size_t i = 0; i < n; i++) h[i] = (uint64_t)(i + 1) | (tag << SC_LANE_TAG_SHIFT);
  LH_HIP(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, c.stream));
  c.sync();
  bool ok = !force_fail;
  volatile uint64_t* hl = c.lanes_host;
  for (size_t i = 0; i < n; i++) hl[64 + i] = 0;
  if (ok) {
    const ncclResult_t r = api.AllReduce(d, c.lanes_host + 64, n, ncclUint64, ncclSum, (ncclComm_t)c.rccl_comm, c.stream);
    ok = r == ncclSuccess;
    if (ok) {
      const auto t0 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < n && ok; i++)
        while (hl[64 + i] != (uint64_t)R * h[i]) {
          __builtin_ia32_pause();
          if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2000)) {
            ok = false;
            break;
          }
        }
      if (hipStreamSynchronize(c.stream) != hipSuccess) ok = false;
    } else {
      (void)hipGetLastError();
    }
  }
  // the ranks agree (a device-to-device all-reduce: the path comm_round = 2 uses)
  uint32_t mine = ok ? 1u : 0u, all = 0;
  uint32_t* dv = (uint32_t*)(d + n);
  LH_HIP(hipMemcpyAsync(dv, &mine, sizeof(mine), hipMemcpyHostToDevice, c.stream));
  rccl_check(api.AllReduce(dv, dv + 1, 1, ncclUint32, ncclMin, (ncclComm_t)c.rccl_comm, c.stream), "ncclAllReduce (probe verdict)");
  LH_HIP(hipMemcpyAsync(&all, dv + 1, sizeof(all), hipMemcpyDeviceToHost, c.stream));
  c.sync();
  c.comm_host_recv_ok = all == 1;
  if (!c.comm_host_recv_ok && c.comm.rank == 0)
    fprintf(stderr, "[comm] ncclAllReduce into pinned host memory failed the attach-time probe on %zu ranks: comm_round = 1 runs as 2 "
                    "(device receive buffer + copy)\n", R);
}

// Measurement aid (lh_ctx_set_comm_loopback): every peer is a copy of this rank.  Rank `rank` of a `size`-rank proof then
// runs alone on its GPU with exactly the kernels, sizes and exchange volumes it has in the real job - the compute half of
// a scaling curve without the other GPUs.  Sums and gathered tables are NOT those of a real job: the transcript it
// produces is not a valid proof.
void comm_attach_loopback(Ctx& c, int rank, int size, size_t shard_bit) {
  comm_detach(c);
  c.comm = lh_comm{rank, size, nullptr, nullptr, nullptr};
  c.comm_loopback = true;
  c.has_comm = true;
  c.shard_bit = shard_bit;
}

static const bool COMM_DEBUG = getenv("LH_COMM_DEBUG") != nullptr;  // one stderr line per collective (development)
static void comm_trace(Ctx& c, const char* what, size_t bytes) {
  c.comm_phase_stats[c.comm_phase & 7][0]++;
  c.comm_phase_stats[c.comm_phase & 7][1] += bytes;
  if (COMM_DEBUG)
    fprintf(stderr, "[comm %d/%d] #%llu %s %zu B\n", c.comm.rank, c.comm.size,
            (unsigned long long)(c.comm_stats[0] + c.comm_stats[1]), what, bytes);
}

static void require_comm(const Ctx& c) {
  LH_REQUIRE(c.has_comm && (c.rccl_comm || c.comm_loopback || c.comm.all_gather || c.comm.all_gather_device), LH_ERR_ARG,
             "no communicator attached (lh_ctx_set_comm / lh_ctx_set_comm_rccl)");
}

void comm_all_gather_dev(Ctx& c, const void* d_send, void* d_recv, size_t bytes) {
  require_comm(c);
  if (!bytes) return;
  comm_trace(c, "all_gather_dev", bytes);
  if (c.comm_loopback) {
    // peer s's data = this rank's rotated by 32 s bytes (whole field elements): the same volume, but not R identical
    // blocks - identical halves of a gathered table would make a quotient identically zero, and its commitment, the
    // identity, cannot be written to the transcript
    // (ONE launch, like the collective it stands for - R to 2 R blits per call were 4.5 ms of an 8-rank rank's 2^24 proof)
    c.comm_stats[0]++;
    k_loopback_gather(c, d_send, d_recv, bytes, (size_t)c.comm.size);
    return;
  }
  if (c.rccl_comm) {
    c.comm_stats[0]++;
    rccl_check(rccl().AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)c.rccl_comm, c.stream), "ncclAllGather");
    return;
  }
  if (c.comm.all_gather_device) {
    c.comm_stats[0]++;
    int rc = c.comm.all_gather_device(c.comm.user, d_send, d_recv, bytes, (void*)c.stream);
    if (rc != 0) throw Error(rc < 0 ? rc : LH_ERR_DEVICE, "communicator all_gather_device failed");
    return;
  }
  // staged through the host (tests with several ranks on one GPU, gloo)
  const size_t R = (size_t)c.comm.size;
  std::vector<uint8_t> mine(bytes), all(bytes * R);
  LH_HIP(hipMemcpyAsync(mine.data(), d_send, bytes, hipMemcpyDeviceToHost, c.stream));
  c.sync();
  c.comm_stats[1]++;
  int rc = c.comm.all_gather(c.comm.user, mine.data(), all.data(), bytes);
  if (rc != 0) throw Error(rc < 0 ? rc : LH_ERR_DEVICE, "communicator all_gather failed");
  LH_HIP(hipMemcpyAsync(d_recv, all.data(), bytes * R, hipMemcpyHostToDevice, c.stream));
  c.sync();  // `all` is pageable host memory
}

void comm_all_gather_host(Ctx& c, const void* send, void* recv, size_t bytes) {
  require_comm(c);
  if (!bytes) return;
  comm_trace(c, "all_gather_host", bytes);
  if (c.comm_loopback) {
    c.comm_stats[1]++;
    for (size_t s = 0; s < (size_t)c.comm.size; s++) memcpy((char*)recv + s * bytes, send, bytes);
    return;
  }
  if (c.comm.all_gather && !c.rccl_comm) {
    c.comm_stats[1]++;
    int rc = c.comm.all_gather(c.comm.user, send, recv, bytes);
    if (rc != 0) throw Error(rc < 0 ? rc : LH_ERR_DEVICE, "communicator all_gather failed");
    return;
  }
  // device collective only: stage through a device buffer [send | recv]
  const size_t R = (size_t)c.comm.size;
  uint8_t* d = (uint8_t*)comm_device_stage(c, bytes * (R + 1));
  LH_HIP(hipMemcpyAsync(d, send, bytes, hipMemcpyHostToDevice, c.stream));
  comm_all_gather_dev(c, d, d + bytes, bytes);
  c.d2h(recv, d + bytes, bytes * R);
}

// Personalised exchange (the sharded access counters: lookups to their address owners and the ranks back), all chunk
// columns in ONE collective.  RCCL: grouped ncclSend / ncclRecv on the ctx's stream - on xGMI's full mesh every pair of GPUs
// has its own link, so an all-to-all costs each link 1/(R-1) of a rank's traffic.  Callback transports have no
// point-to-point primitive: every rank's whole send allocation is all-gathered and the segments meant for this rank are
// picked out (tests only).
void comm_all_to_all_multi(Ctx& c, size_t nbuf, const void* const* d_send, const size_t* send_off, const size_t* send_cnt,
                           void* const* d_recv, const size_t* recv_off, const size_t* recv_cnt, const size_t* peer_off,
                           const void* send_base, size_t send_stride, size_t elem) {
  require_comm(c);
  const size_t R = (size_t)c.comm.size, me = (size_t)c.comm.rank;
  size_t total = 0;
  for (size_t q = 0; q < nbuf * R; q++) total += send_cnt[q] + recv_cnt[q];
  comm_trace(c, "all_to_all_v", total * elem);
  // LH_COMM_A2A=allgather: stage the personalised exchange through ncclAllGather under RCCL as well (a fallback should
  // grouped send / recv misbehave on some fabric; R times the traffic)
  static const bool a2a_by_gather = [] {
    const char* e = getenv("LH_COMM_A2A");
    return e && strcmp(e, "allgather") == 0;
  }();
  // LH_COMM_A2A_SELF=1 (tests): the segment a rank keeps for itself ALSO travels by ncclSend / ncclRecv - on a one-GPU box
  // the only way the grouped point-to-point path (symbols, group semantics, stream ordering) ever executes before the
  // first multi-GPU run
  static const bool a2a_self = [] {
    const char* e = getenv("LH_COMM_A2A_SELF");
    return e && atoi(e) != 0;
  }();
  if (c.rccl_comm && !a2a_by_gather) {
    const RcclApi& api = rccl();
    c.comm_stats[0]++;
    // (the segment a rank keeps for itself is a device copy: no self-send)
    if (!a2a_self)
      for (size_t b = 0; b < nbuf; b++)
        if (send_cnt[b * R + me])
          LH_HIP(hipMemcpyAsync((char*)d_recv[b] + recv_off[b * R + me] * elem, (const char*)d_send[b] + send_off[b * R + me] * elem,
                                send_cnt[b * R + me] * elem, hipMemcpyDeviceToDevice, c.stream));
    if (R > 1 || a2a_self) {
      rccl_check(api.GroupStart(), "ncclGroupStart");
      // (sends and receives between a pair of ranks match in the order they are issued: buffer by buffer on both sides)
      for (size_t p = 0; p < R; p++) {
        if (p == me && !a2a_self) continue;
        for (size_t b = 0; b < nbuf; b++) {
          if (send_cnt[b * R + p])
            rccl_check(api.Send((const char*)d_send[b] + send_off[b * R + p] * elem, send_cnt[b * R + p] * elem, ncclUint8, (int)p,
                                (ncclComm_t)c.rccl_comm, c.stream), "ncclSend");
          if (recv_cnt[b * R + p])
            rccl_check(api.Recv((char*)d_recv[b] + recv_off[b * R + p] * elem, recv_cnt[b * R + p] * elem, ncclUint8, (int)p,
                                (ncclComm_t)c.rccl_comm, c.stream), "ncclRecv");
        }
      }
      rccl_check(api.GroupEnd(), "ncclGroupEnd");
    }
    return;
  }
  ArenaScope scope(c.arena);
  const size_t span = nbuf * send_stride;  // elements of a rank's send allocation
  char* all = (char*)c.arena.alloc(std::max<size_t>(span * elem * R, 256));
  comm_all_gather_dev(c, send_base, all, span * elem);
  for (size_t b = 0; b < nbuf; b++)
    for (size_t p = 0; p < R; p++)
      if (recv_cnt[b * R + p])
        LH_HIP(hipMemcpyAsync((char*)d_recv[b] + recv_off[b * R + p] * elem,
                              all + (p * span + b * send_stride + peer_off[b * R + p]) * elem, recv_cnt[b * R + p] * elem,
                              hipMemcpyDeviceToDevice, c.stream));
  c.sync();  // (`all` is released with the scope)
}

// closing steps of a sharded sum-check round: all-gather of every rank's partial sums, then the sum-and-publish kernel
void comm_sum_publish(Ctx& c, const Fr* d_part, Fr* d_scratch, size_t count, Fr* out_host, uint32_t seq) {
  comm_all_gather_dev(c, d_part, d_scratch, count * sizeof(Fr));
  k_sum_publish(c, d_scratch, (size_t)c.comm.size, count, out_host, seq);
}


// ------------------------------------------------------------------ the all-reduce variant of a sharded round (Options::comm_round)
// north_star's collective: "per-round partial sums combined via RCCL all-reduce".  A field element is 8 limbs of 32 bits;
// every limb travels in a u64 lane, so ncclSum over at most 2^8 ranks cannot lose a carry (a lane's low 40 bits hold the
// sum), and the lane's upper 24 bits carry a tag every rank stamps alike: a lane whose upper bits read R * tag IS the
// finished sum of this round.  The collective writes the lanes where the host polls them (pinned memory; comm_round 2:
// into device memory, then a copy), the host adds the limbs up as one wide integer and reduces it mod r - no kernel runs
// behind the collective.  Never run on more than one GPU: tests cover a world of one over RCCL and 2 / 4 ranks over the
// callback transports (the lanes are then all-gathered through the host and added there).
uint32_t comm_next_tag(Ctx& c) {
  const uint32_t R = (uint32_t)std::max(c.comm.size, 1);
  const uint32_t span = (1u << 24) / R;  // tags below this: R * tag < 2^24
  c.sc_tag_seq = c.sc_tag_seq % (span - 1) + 1;  // 1 .. span - 1, consecutive rounds differ
  return c.sc_tag_seq;
}

void comm_sum_lanes(Ctx& c, uint64_t* d_lanes, uint64_t* d_scratch, size_t count, Fr* out_host) {
  require_comm(c);
  const size_t R = (size_t)c.comm.size, n = 8 * count;
  LH_REQUIRE(count >= 1 && count <= 16 && R <= 256, LH_ERR_ARG, "comm_sum_lanes: bad size");
  comm_trace(c, "all_reduce_lanes", n * sizeof(uint64_t));
  if (!c.lanes_host) {
    LH_HIP(hipHostMalloc((void**)&c.lanes_host, 128 * sizeof(uint64_t), hipHostMallocCoherent | hipHostMallocMapped));
    memset(c.lanes_host, 0, 128 * sizeof(uint64_t));
  }
  volatile uint64_t* hl = c.lanes_host;
  const bool via_device = c.opt.comm_round == 2 || !c.comm_host_recv_ok;  // (comm_probe_host_recv: this fabric refused host memory)
  uint64_t* dst = via_device ? d_scratch : c.lanes_host;
  if (c.comm_loopback) {
    c.comm_stats[0]++;
    k_loopback_allreduce_lanes(c, d_lanes, n, R, dst);
    if (via_device) LH_HIP(hipMemcpyAsync(c.lanes_host, d_scratch, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
  } else if (c.rccl_comm) {
    c.comm_stats[0]++;
    rccl_check(rccl().AllReduce(d_lanes, dst, n, ncclUint64, ncclSum, (ncclComm_t)c.rccl_comm, c.stream), "ncclAllReduce");
    if (via_device) LH_HIP(hipMemcpyAsync(c.lanes_host, d_scratch, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
  } else {
    // callback transports (gloo in the tests): the lanes are all-gathered through the host and added there - the same
    // arithmetic, the same lanes, the same wait below
    std::vector<uint64_t> mine(n), all(n * R);
    c.d2h(mine.data(), d_lanes, n * sizeof(uint64_t));
    comm_all_gather_host(c, mine.data(), all.data(), n * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) {
      uint64_t acc = 0;
      for (size_t s = 0; s < R; s++) acc += all[s * n + i];
      hl[i] = acc;
    }
  }
  // every lane validates itself: wait until each carries the ranks' tags
  // (bounded: a peer that never arrives - hipErrorNotReady for ever - must not hang this rank: LH_COMM_WAIT_TIMEOUT_MS)
  static const long wait_ms = [] { const char* e = getenv("LH_COMM_WAIT_TIMEOUT_MS"); return e && atol(e) > 0 ? atol(e) : 30000L; }();
  const auto wait_t0 = std::chrono::steady_clock::now();
  const uint64_t expect = ((uint64_t)R * c.sc_tag) & 0xffffffull;
  for (size_t i = 0; i < n; i++) {
    for (uint64_t spin = 0; (hl[i] >> SC_LANE_TAG_SHIFT) != expect; spin++) {
      __builtin_ia32_pause();
      if ((spin & 0xfffff) == 0xfffff) {
        if (std::chrono::steady_clock::now() - wait_t0 > std::chrono::milliseconds(wait_ms))
          throw Error(LH_ERR_DEVICE, "all-reduce of a sharded round: the sums of all ranks did not arrive in time (a peer is gone?)");
        hipError_t e = hipStreamQuery(c.stream);
        if (e == hipSuccess) {
          if ((hl[i] >> SC_LANE_TAG_SHIFT) == expect) break;
          throw Error(LH_ERR_DEVICE, "all-reduce finished without delivering the round's sums");
        }
        if (e != hipErrorNotReady) throw Error(LH_ERR_DEVICE, std::string("stream error: ") + hipGetErrorString(e));
      }
    }
  }
  // lazy reduction: sum_k lane_k 2^(32 k) < R r as a 320-bit integer, minus r while it is not below it
  const uint64_t mask = ((uint64_t)1 << SC_LANE_TAG_SHIFT) - 1;
  for (size_t x = 0; x < count; x++) {
    uint64_t w[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 8; k++) {
      const unsigned __int128 v = (unsigned __int128)(hl[8 * x + k] & mask) << (32 * (k & 1));
      unsigned __int128 cy = v;
      for (int q = k >> 1; q < 5 && cy; q++) {
        cy += w[q];
        w[q] = (uint64_t)cy;
        cy >>= 64;
      }
    }
    while (w[4] || host::F<host::FrTag>::geq_mod(w)) {
      unsigned __int128 bw = 0;
      for (int q = 0; q < 5; q++) {
        const unsigned __int128 d = (unsigned __int128)w[q] - (q < 4 ? host::FrTag::MOD[q] : 0) - bw;
        w[q] = (uint64_t)d;
        bw = (d >> 64) & 1;
      }
    }
    memcpy(&out_host[x], w, 32);
  }
}

}  // namespace lh
