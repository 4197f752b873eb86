// Device radix sort of (bucket key, point index) pairs for the MSM (rocPRIM's device-wide sort:
// a utility primitive from the ROCm toolchain; the curve arithmetic around it is hand-written).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "dev.hpp"

namespace lh {

void sort_pairs_u32(Ctx& c, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                    size_t n, unsigned bits) {
  size_t temp_bytes = 0;
  LH_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, c.stream));
  void* temp = c.arena.alloc(temp_bytes ? temp_bytes : 256);  // caller's ArenaScope releases it
  LH_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, c.stream));
}

// 64-bit keys (the sharded access counters sort (address, global lookup index) on the address owner)
void sort_pairs_u64(Ctx& c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                    size_t n, unsigned bits) {
  size_t temp_bytes = 0;
  LH_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, c.stream));
  void* temp = c.arena.alloc(temp_bytes ? temp_bytes : 256);
  LH_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, c.stream));
}

size_t sort_pairs_u32_temp_bytes(size_t n, unsigned bits) {
  size_t temp_bytes = 0;
  LH_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                   (uint32_t*)nullptr, n, 0u, bits, (hipStream_t) nullptr));
  return temp_bytes ? temp_bytes : 256;
}

// the same with caller-provided temporary storage (several sorts of one stream share it)
void sort_pairs_u32_with(Ctx& c, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                         uint32_t* vals_out, size_t n, unsigned bits, void* temp, size_t temp_bytes) {
  LH_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, c.stream));
}

}  // namespace lh
