// Device-side pieces shared by the resident kernels (kernels_sumcheck.hip sc_tail_kernel, kernels_gkr.hip): 16-byte
// system-scope accesses, the self-validating message chunks (dev.hpp TailChunk), the ticket hand-off between workgroups.
#pragma once
#include <hip/hip_runtime.h>
#include "dev.hpp"

namespace lh {

__device__ __forceinline__ void publish_flag(uint32_t* flag, uint32_t seq) {
  // results were written by this thread just before: release them to the host, then the sequence number
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the closing step of a round kernel, by the ONE thread that publishes, once the launch's D sums are in f.out_host and
// visible to it (its own stores, or the workgroup's before a barrier).  Sharded rounds of the all-reduce variant
// (ScFinishArgs::wide): the sums leave a second time as tagged u64 lanes - re-read with agent-scope loads, the stores
// may be another wave's.
__device__ __forceinline__ void publish_round(const ScFinishArgs& f, int D) {
  if (f.wide) {
    for (int x = 0; x < D; x++)
      for (int k = 0; k < 8; k++) {
        const uint32_t limb = __hip_atomic_load(&f.out_host[x].l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f.wide[8 * x + k] = (uint64_t)limb | ((uint64_t)f.tag << SC_LANE_TAG_SHIFT);
      }
  }
  publish_flag(f.flag, f.seq);
}

__device__ __forceinline__ Fr shfl_xor_fr(const Fr& v, int mask) {
  Fr o;
#pragma unroll
  for (int i = 0; i < 8; i++) o.l[i] = __shfl_xor(v.l[i], mask, 64);
  return o;
}

// 16-byte system-scope accesses: one request to host memory each
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load_sys_x4(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_sys_x4(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void tail_send(TailChunk* msg, uint32_t x, const Fr& s, uint32_t seq) {
  store_sys_x4(&msg[3 * x + 0], u32x4{seq, s.l[0], s.l[1], s.l[2]});
  store_sys_x4(&msg[3 * x + 1], u32x4{seq, s.l[3], s.l[4], s.l[5]});
  store_sys_x4(&msg[3 * x + 2], u32x4{seq, s.l[6], s.l[7], 0u});
}

// wave 0, thread 0 of a workgroup: release this workgroup's stores (all of wave 0), draw a ticket; true when it is
// the last of its batch (then every other workgroup's stores are visible to this CU)
__device__ __forceinline__ bool tail_ticket(uint32_t* ticket, uint32_t last_ticket) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool last = t == last_ticket;
  if (last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  return last;
}


}  // namespace lh
