// Development aid: rocPRIM radix_sort_pairs of (u32 key, u32 value) with 17-22 significant key bits under the library's
// tuned default (8 bits per onesweep pass: 3 passes) against onesweep configurations with 9-11 bits per pass (2 passes).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/radix_bits.hip -o /tmp/radix_bits ; run: /tmp/radix_bits [log_n]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <unsigned RB, unsigned BS, unsigned IPT>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>,
                                                                           rocprim::kernel_config<BS, IPT>, RB,
                                                                           rocprim::block_radix_rank_algorithm::match>>;

template <class Config>
static double run(const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in, uint32_t* v_out, size_t n, unsigned bits,
                  bool check) {
  size_t tb = 0;
  CHECK((rocprim::radix_sort_pairs<Config>(nullptr, tb, k_in, k_out, v_in, v_out, n, 0u, bits, (hipStream_t)0)));
  void* tmp;
  CHECK(hipMalloc(&tmp, tb));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    CHECK((rocprim::radix_sort_pairs<Config>(tmp, tb, k_in, k_out, v_in, v_out, n, 0u, bits, (hipStream_t)0)));
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  if (check) {
    std::vector<uint32_t> k(n), v(n), ki(n);
    CHECK(hipMemcpy(k.data(), k_out, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(v.data(), v_out, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ki.data(), k_in, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
      if (i && k[i - 1] > k[i]) { printf("NOT SORTED at %zu\n", i); exit(2); }
      if (ki[v[i]] != k[i]) { printf("VALUE MISMATCH at %zu\n", i); exit(2); }
      if (i && k[i - 1] == k[i] && v[i - 1] > v[i]) { printf("NOT STABLE at %zu\n", i); exit(2); }
    }
  }
  CHECK(hipFree(tmp));
  return best;
}

int main(int argc, char** argv) {
  const int log_n = argc > 1 ? atoi(argv[1]) : 24;
  const size_t n = (size_t)1 << log_n;
  uint32_t *k_in, *k_out, *v_in, *v_out;
  CHECK(hipMalloc(&k_in, n * 4)); CHECK(hipMalloc(&k_out, n * 4)); CHECK(hipMalloc(&v_in, n * 4)); CHECK(hipMalloc(&v_out, n * 4));
  std::vector<uint32_t> h(n), iota(n);
  for (unsigned bits : {16u, 17u, 18u, 20u, 22u}) {
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) {
      s ^= s << 13, s ^= s >> 7, s ^= s << 17;
      h[i] = (uint32_t)(s >> 20) & ((1u << bits) - 1u);
      iota[i] = (uint32_t)i;
    }
    CHECK(hipMemcpy(k_in, h.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(v_in, iota.data(), n * 4, hipMemcpyHostToDevice));
    printf("2^%d pairs, %u key bits: default %.3f ms", log_n, bits, run<rocprim::default_config>(k_in, k_out, v_in, v_out, n, bits, true));
    if (bits <= 18) printf(" | 9 bits/pass %.3f ms", run<cfg<9, 256, 16>>(k_in, k_out, v_in, v_out, n, bits, true));
    if (bits <= 20) printf(" | 10 bits/pass %.3f ms", run<cfg<10, 256, 16>>(k_in, k_out, v_in, v_out, n, bits, true));
    printf(" | 11 bits/pass %.3f ms\n", run<cfg<11, 256, 16>>(k_in, k_out, v_in, v_out, n, bits, true));
  }
  return 0;
}
