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

// A workgroup's partial sum number idx on its way to the workgroup that finishes the launch (dev.hpp ScFinishArgs::lanes):
// eight self-validating 8-byte lanes when the launch has a lane buffer, a plain store (under the fences of the ticket
// protocol - or straight into the host buffer of a single-workgroup launch) when not.
__device__ __forceinline__ void fin_put(const ScFinishArgs& f, Fr* partials, size_t idx, const Fr& v) {
  if (f.lanes && gridDim.x * gridDim.y > 1) {
#pragma unroll
    for (int k = 0; k < 8; k++)
      __hip_atomic_store(&f.lanes[idx * 8 + k], (uint64_t)v.l[k] | ((uint64_t)f.seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    partials[idx] = v;
  }
}
// (by the finishing workgroup, after the last ticket: every lane was stored before its workgroup drew its ticket, so the
//  poll is a formality - a lane that never arrives is a bug, and ends the kernel rather than hanging the device)
__device__ __forceinline__ Fr fin_get(const ScFinishArgs& f, const Fr* partials, size_t idx) {
  if (!f.lanes) return partials[idx];
  uint64_t v[8];
  for (uint32_t spin = 0;; spin++) {  // all eight loads in flight, then the tags
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __hip_atomic_load(&f.lanes[idx * 8 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < 8; k++) ok = ok && (uint32_t)(v[k] >> 32) == f.seq;
    if (ok) break;
    if (spin > (1u << 22)) __builtin_trap();
    __builtin_amdgcn_s_sleep(1);
  }
  Fr p;
#pragma unroll
  for (int k = 0; k < 8; k++) p.l[k] = (uint32_t)v[k];
  return p;
}
// the ticket of a workgroup whose partial sums are on their way (its wave 0 stored them): true for the workgroup that
// finishes the launch.  With lanes: no fence (resident.cuh fin_put); without: agent-scope release by every producer,
// agent-scope acquire by the one consumer (cdna_hip_programming.md Guideline 16).  Every thread of the workgroup calls it.
__device__ __forceinline__ bool fin_ticket(const ScFinishArgs& f, int* is_last_lds) {
  if (threadIdx.x < 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial stores have left
    if (threadIdx.x == 0) {
      if (!f.lanes) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const uint32_t t = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == f.last_ticket;
      if (last && !f.lanes) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *is_last_lds = last;
    }
  }
  __syncthreads();
  return *is_last_lds != 0;
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
