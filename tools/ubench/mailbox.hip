// Development micro-benchmark: round trip between a RESIDENT kernel and the host through pinned memory.
// The kernel publishes a sequence number, the host answers in a mailbox, the kernel polls the mailbox:
// the turnaround a resident sum-check tail would pay per round instead of launch + completion latency.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mailbox.hip -o /tmp/mailbox
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Mbox { unsigned seq; unsigned pad[7]; unsigned r[8]; };

__global__ void resident(unsigned* to_host, unsigned* payload, const Mbox* mbox, unsigned rounds, unsigned* status) {
  __shared__ unsigned r_sh[8];
  const unsigned long long t_start = wall_clock64();
  for (unsigned i = 1; i <= rounds; i++) {
    if (threadIdx.x < 8) payload[threadIdx.x] = i * 8 + threadIdx.x;  // the "round message"
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(to_host, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      // poll the mailbox, bounded: 100 MHz clock, 2 s
      while (__hip_atomic_load(&mbox->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != i) {
        if (wall_clock64() - t_start > 200000000ull) { *status = 0xdead; r_sh[0] = 0xffffffffu; break; }
      }
      for (int k = 0; k < 8; k++) r_sh[k] = __hip_atomic_load(&mbox->r[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (r_sh[0] == 0xffffffffu) return;
  }
  if (threadIdx.x == 0) *status = r_sh[0];
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned* host = nullptr;
  CK(hipHostMalloc((void**)&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  for (int i = 0; i < 1024; i++) host[i] = 0;
  unsigned* to_host = host;             // line 0
  unsigned* payload = host + 16;        // line 1
  Mbox* mbox = (Mbox*)(host + 64);      // line 4..
  unsigned* status = host + 128;
  const unsigned rounds = 2000;
  double t0 = now_us();
  hipLaunchKernelGGL(resident, 1, 256, 0, s, to_host, payload, mbox, rounds, status);
  double worst = 0;
  for (unsigned i = 1; i <= rounds; i++) {
    double a = now_us();
    while (*(volatile unsigned*)to_host != i) {
      if (now_us() - a > 3e6) { printf("timeout at round %u\n", i); return 2; }
    }
    if (((volatile unsigned*)payload)[7] != i * 8 + 7) { printf("payload not visible at round %u\n", i); return 3; }
    for (int k = 0; k < 8; k++) ((volatile unsigned*)mbox->r)[k] = i + k;
    __atomic_store_n(&mbox->seq, i, __ATOMIC_RELEASE);
    worst = std::max(worst, now_us() - a);
  }
  CK(hipStreamSynchronize(s));
  double dt = now_us() - t0;
  printf("resident kernel <-> host: %.2f us per round trip over %u rounds (worst host wait %.1f us), status %u\n",
         dt / rounds, rounds, worst, *status);
  return 0;
}
