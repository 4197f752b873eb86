// Wave64 / workgroup reductions of field elements (modular sums).
#pragma once
#if !defined(__HIPCC_RTC__)  // the runtime compiler (jit.cpp) has the HIP built-ins without headers
#include <hip/hip_runtime.h>
#endif
#include "ff.cuh"

namespace lh {

// sum over the 64 lanes of a wavefront via __shfl_down (8 dwords per element per step)
template <class P>
__device__ __forceinline__ Fp<P> wave_reduce_sum(Fp<P> v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Fp<P> o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(v.l[i], off, 64);
    v = add(v, o);
  }
  return v;
}

// blockDim.x must be a multiple of 64 and <= 256; lds: one slot per wave. Result valid in thread 0.
template <class P>
__device__ __forceinline__ Fp<P> block_reduce_sum(Fp<P> v, Fp<P>* lds) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();  // lds may be reused across calls
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; w++) v = add(v, lds[w]);
  }
  return v;
}

}  // namespace lh
