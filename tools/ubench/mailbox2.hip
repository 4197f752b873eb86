// Development micro-benchmark: where should the host's answer to a RESIDENT kernel live, and how should the kernel look
// for it?  A round of a resident sum-check is [kernel: message -> pinned host memory] [host: transcript, challenge]
// [kernel polls for the challenge].  Production keeps the challenge mailbox in pinned host memory and polls it with ONE
// 16-byte read in flight per lane: every look is a PCIe read round trip (~1.35 us).
//   A   mailbox in pinned host memory, one read in flight                     (production)
//   A4  the same mailbox, four reads in flight per lane (one asm block, the oldest examined and re-issued as it returns)
//   B   mailbox in fine-grained DEVICE memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)) that the host stores
//       into through the BAR (posted writes) while the kernel polls its own memory
//   C   the same in plain hipMalloc memory
// Message and answer have production's shape (dev.hpp TailChunk): three 16-byte chunks each, the sequence number in every
// chunk; the message goes to pinned host memory in every variant.
// RESULT (MI355X, round 5; profiles/README.md): with an idle host B is 0.5-0.65 us per round faster than A (2.2-2.4 against
// 2.8-3.0 us) on every XCD - but A's time is QUANTISED: the first look comes too early, the second samples the mailbox
// ~2.1 us after the message left, so any host time up to ~1.3 us between message and answer is hidden (2.8 us flat for 0 ..
// 1.0 us, 4.1 us at 1.5 us), while B and A4 pay it in full (B = 2.15 us + host time, A4 = 2.67 us + host time).  The prover's
// host spends ~0.8-0.9 us per round (two Keccak-f, a dozen field operations): A 2.8, B 3.0-3.2, A4 3.4.  Built into the
// library behind an option and measured there (rounds of the resident grand-product kernel by its device clock: pinned
// 3.0-3.2 us, device memory 3.1-3.5 us) - and taken out again: production's mailbox stays where it is.
// build: hipcc -O3 -mavx2 --offload-arch=gfx950 tools/ubench/mailbox2.hip -o /tmp/mailbox2
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <setjmp.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Chunk { unsigned seq, a, b, c; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load_sys_x4(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_sys_x4(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// production's shape: three 16-byte chunks out (one field element), three chunks back, lanes 0..2 poll one chunk each
__global__ void resident(Chunk* to_host, const Chunk* mbox, unsigned rounds, unsigned* status, unsigned work, unsigned active_wg) {
  if (blockIdx.x != active_wg) return;  // (workgroup k of a launch runs on XCD k mod 8: does the home XCD matter?)
  __shared__ unsigned r_sh[12];
  __shared__ unsigned bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const unsigned long long t_start = wall_clock64();
  unsigned acc = threadIdx.x;
  for (unsigned i = 1; i <= rounds; i++) {
    for (unsigned k = 0; k < work; k++) acc = acc * 1664525u + 1013904223u;  // stand-in for the round's arithmetic
    if (threadIdx.x < 3) store_sys_x4(&to_host[threadIdx.x], u32x4{i, acc, i * 3u, i * 5u});
    if (threadIdx.x < 3) {
      u32x4 v;
      for (;;) {
        v = load_sys_x4(&mbox[threadIdx.x]);
        if (v.x == i) break;
        if (wall_clock64() - t_start > 400000000ull) { bad = 1; break; }
      }
      r_sh[4 * threadIdx.x] = v.y, r_sh[4 * threadIdx.x + 1] = v.z, r_sh[4 * threadIdx.x + 2] = v.w;
    }
    __syncthreads();
    if (bad) { if (threadIdx.x == 0) *status = 0xdead; return; }
    acc += r_sh[threadIdx.x % 11];
    __syncthreads();
  }
  if (threadIdx.x == 0) *status = r_sh[0] + (acc & 1);
}

// PIPELINED polling of the production chunk format, ONE asm block (the registers a load lands in long after its issue
// must belong to the block for its whole duration: explicit physical registers).  The active lanes (0..2: one chunk each)
// keep DEPTH 16-byte loads in flight; the oldest is examined when it returns (loads return in order: vmcnt(DEPTH - 1)) and
// re-issued at once, so the mailbox is sampled every round-trip / DEPTH.  Returns after `iters` passes without the number
// (the caller looks for an abort marker and at its clock, then comes back).
#define POLL_STEP(R0, R1, R2, R3, LBL, WAIT)                                       \
  "s_waitcnt vmcnt(" #WAIT ")\n\t"                                                 \
  "v_cmp_ne_u32 vcc, %[seq], v" #R0 "\n\t"                                         \
  "s_cbranch_vccz " #LBL "f\n\t"                                                   \
  "global_load_dwordx4 v[" #R0 ":" #R3 "], %[p], off sc0 sc1\n\t"
#define POLL_TAKE(R0, R1, R2, R3, LBL)                                             \
  #LBL ":\n\t"                                                                     \
  "v_mov_b32 v112, v" #R0 "\n\tv_mov_b32 v113, v" #R1 "\n\tv_mov_b32 v114, v" #R2 "\n\tv_mov_b32 v115, v" #R3 "\n\t" \
  "s_mov_b32 %[found], 1\n\ts_branch 9f\n\t"
__device__ __forceinline__ bool poll_pipelined4(const void* p, unsigned seq, unsigned iters, u32x4& out) {
  unsigned found, cnt;
  u32x4 r;
  asm volatile(
      "s_mov_b32 %[cnt], %[iters]\n\t"
      "global_load_dwordx4 v[96:99], %[p], off sc0 sc1\n\ts_sleep 10\n\t"
      "global_load_dwordx4 v[100:103], %[p], off sc0 sc1\n\ts_sleep 10\n\t"
      "global_load_dwordx4 v[104:107], %[p], off sc0 sc1\n\ts_sleep 10\n\t"
      "global_load_dwordx4 v[108:111], %[p], off sc0 sc1\n\t"
      "1:\n\t"
      POLL_STEP(96, 97, 98, 99, 2, 3)
      POLL_STEP(100, 101, 102, 103, 3, 3)
      POLL_STEP(104, 105, 106, 107, 4, 3)
      POLL_STEP(108, 109, 110, 111, 5, 3)
      "s_sub_u32 %[cnt], %[cnt], 1\n\t"
      "s_cmp_lg_u32 %[cnt], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "v_mov_b32 v112, v108\n\tv_mov_b32 v113, v109\n\tv_mov_b32 v114, v110\n\tv_mov_b32 v115, v111\n\t"
      "s_mov_b32 %[found], 0\n\ts_branch 9f\n\t"
      POLL_TAKE(96, 97, 98, 99, 2)
      POLL_TAKE(100, 101, 102, 103, 3)
      POLL_TAKE(104, 105, 106, 107, 4)
      POLL_TAKE(108, 109, 110, 111, 5)
      "9:\n\t"
      "s_waitcnt vmcnt(0)"
      : [found] "=&s"(found), [cnt] "=&s"(cnt), "={v[112:115]}"(r)
      : [p] "v"(p), [seq] "s"(seq), [iters] "s"(iters)
      : "memory", "vcc", "scc", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",
        "v108", "v109", "v110", "v111");
  out = r;
  return found != 0;
}

__global__ void __launch_bounds__(256) resident_asm4(Chunk* to_host, const Chunk* mbox, unsigned rounds, unsigned* status) {
  __shared__ unsigned r_sh[12];
  __shared__ unsigned bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const unsigned long long t_start = wall_clock64();
  unsigned acc = threadIdx.x;
  for (unsigned i = 1; i <= rounds; i++) {
    if (threadIdx.x < 3) store_sys_x4(&to_host[threadIdx.x], u32x4{i, acc, i * 3u, i * 5u});
    if (threadIdx.x < 3) {
      u32x4 v;
      for (;;) {
        if (poll_pipelined4(&mbox[threadIdx.x], i, 64, v)) break;
        if (wall_clock64() - t_start > 400000000ull) { bad = 1; break; }
      }
      r_sh[4 * threadIdx.x] = v.y, r_sh[4 * threadIdx.x + 1] = v.z, r_sh[4 * threadIdx.x + 2] = v.w;
    }
    __syncthreads();
    if (bad) { if (threadIdx.x == 0) *status = 0xdead; return; }
    acc += r_sh[threadIdx.x % 11];
    __syncthreads();
  }
  if (threadIdx.x == 0) *status = r_sh[0] + (acc & 1);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static sigjmp_buf probe_env;
static void on_segv(int) { siglongjmp(probe_env, 1); }
// can the host store to p?  (a device pointer without a CPU mapping faults)
static bool host_can_write(void* q) {
  volatile unsigned long long* p = (volatile unsigned long long*)q;
  struct sigaction sa, old_segv, old_bus;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_segv;
  sigaction(SIGSEGV, &sa, &old_segv);
  sigaction(SIGBUS, &sa, &old_bus);
  bool ok = false;
  if (!sigsetjmp(probe_env, 1)) {
    p[0] = 0;
    _mm_sfence();
    ok = true;
  }
  sigaction(SIGSEGV, &old_segv, nullptr);
  sigaction(SIGBUS, &old_bus, nullptr);
  return ok;
}

static int run(const char* name, hipStream_t s, Chunk* to_host, Chunk* mbox_dev, volatile Chunk* mbox_host, unsigned* status,
               unsigned rounds, unsigned work, int store_kind, unsigned active_wg = 0, double host_delay_us = 0, int kernel_kind = 0) {
  memset((void*)to_host, 0, 4 * sizeof(Chunk));
  for (int k = 0; k < 4; k++) _mm_store_si128((__m128i*)&mbox_host[k], _mm_setzero_si128());
  _mm_sfence();
  *status = 0;
  std::vector<double> rt(rounds);
  double t0 = now_us();
  if (kernel_kind == 1) hipLaunchKernelGGL(resident_asm4, 1, 256, 0, s, to_host, mbox_dev, rounds, status);
  else hipLaunchKernelGGL(resident, 8, 256, 0, s, to_host, mbox_dev, rounds, status, work, active_wg);
  double prev = now_us();
  for (unsigned i = 1; i <= rounds; i++) {
    double a = now_us();
    for (int k = 0; k < 3; k++)
      while (((volatile Chunk*)to_host)[k].seq != i) {
        if (now_us() - a > 3e6) { printf("%s: timeout at round %u\n", name, i); return 2; }
      }
    if (host_delay_us > 0) {  // the host's share of a round (transcript, a dozen field operations)
      const double until = now_us() + host_delay_us;
      while (now_us() < until) {}
    }
    if (store_kind == 0) {
      for (int k = 0; k < 3; k++) _mm_store_si128((__m128i*)&mbox_host[k], _mm_set_epi32((int)(i + k), (int)(i * 7u), (int)(i * 3u), (int)i));
    } else {  // the whole 64-byte line by two AVX stores: the write-combining buffer leaves as one write
      __m256i lo = _mm256_set_epi32((int)(i + 1), (int)(i * 7u), (int)(i * 3u), (int)i, (int)i, (int)(i * 7u), (int)(i * 3u), (int)i);
      __m256i hi = _mm256_set_epi32(0, 0, 0, 0, (int)(i + 2), (int)(i * 7u), (int)(i * 3u), (int)i);
      _mm256_store_si256((__m256i*)&mbox_host[0], lo);
      _mm256_store_si256((__m256i*)&mbox_host[2], hi);
    }
    _mm_sfence();
    double b = now_us();
    rt[i - 1] = b - prev;
    prev = b;
  }
  CK(hipStreamSynchronize(s));
  double dt = now_us() - t0;
  std::sort(rt.begin() + 1, rt.end());
  printf("%-48s work %5u: %.2f us per round (median %.2f, p99 %.2f), status %u\n", name, work, dt / rounds, rt[rounds / 2],
         rt[(size_t)(rounds * 0.99)], *status);
  return 0;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned* host = nullptr;
  CK(hipHostMalloc((void**)&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  memset(host, 0, 4096);
  Chunk* to_host = (Chunk*)host;
  unsigned* status = host + 128;
  Chunk* mbox_a = (Chunk*)(host + 64);
  Chunk* mbox_b = nullptr;
  Chunk* mbox_c = nullptr;
  hipError_t eb = hipExtMallocWithFlags((void**)&mbox_b, 4096, hipDeviceMallocFinegrained);
  if (eb != hipSuccess) printf("fine-grained device allocation failed: %s\n", hipGetErrorString(eb)), mbox_b = nullptr, (void)hipGetLastError();
  CK(hipMalloc((void**)&mbox_c, 4096));
  CK(hipMemset(mbox_c, 0, 4096));
  if (mbox_b) CK(hipMemset(mbox_b, 0, 4096));
  CK(hipDeviceSynchronize());
  const bool b_ok = mbox_b && host_can_write(mbox_b), c_ok = host_can_write(mbox_c);
  printf("host can store to: fine-grained device memory %s, plain device memory %s\n", b_ok ? "yes" : "NO", c_ok ? "yes" : "NO");
  const unsigned rounds = 4000;
  for (unsigned work : {0u, 2000u}) {
    for (int kind = 0; kind < 2; kind++) {
      if (run(kind ? "A pinned host mailbox (avx)" : "A pinned host mailbox (16-byte stores)", s, to_host, mbox_a, mbox_a, status, rounds, work, kind)) return 1;
      if (b_ok && run(kind ? "B fine-grained device mailbox (avx)" : "B fine-grained device mailbox (16-byte stores)", s, to_host, mbox_b, mbox_b, status, rounds, work, kind)) return 1;
      if (c_ok && run(kind ? "C plain device mailbox (avx)" : "C plain device mailbox (16-byte stores)", s, to_host, mbox_c, mbox_c, status, rounds, work, kind)) return 1;
    }
  }
  for (unsigned wg = 0; wg < 8; wg++) {
    char name[64];
    snprintf(name, sizeof(name), "A pinned host mailbox, workgroup %u", wg);
    if (run(name, s, to_host, mbox_a, mbox_a, status, rounds, 0, 0, wg)) return 1;
    snprintf(name, sizeof(name), "B fine-grained device mailbox, workgroup %u", wg);
    if (b_ok && run(name, s, to_host, mbox_b, mbox_b, status, rounds, 0, 0, wg)) return 1;
  }
  // the host's share of a round between message and answer: 0 .. 1.5 us
  for (double d : {0.0, 0.3, 0.8, 1.0, 1.5}) {
    char name[80];
    snprintf(name, sizeof(name), "A  pinned, one read in flight, host busy %.1f us", d);
    if (run(name, s, to_host, mbox_a, mbox_a, status, rounds, 0, 0, 0, d)) return 1;
    snprintf(name, sizeof(name), "A4 pinned, 4 reads in flight,  host busy %.1f us", d);
    if (run(name, s, to_host, mbox_a, mbox_a, status, rounds, 0, 0, 0, d, 1)) return 1;
    snprintf(name, sizeof(name), "B  device memory,              host busy %.1f us", d);
    if (b_ok && run(name, s, to_host, mbox_b, mbox_b, status, rounds, 0, 0, 0, d)) return 1;
  }
  return 0;
}
