// Development micro-benchmark: a chain of dependent G1 additions by lone lanes (ec.cuh add) against a
// quad-cooperative addition (4 lanes share the products of one formula step), one wave per SIMD.  Checks that both give
// the same bytes.  Result (profiles/README.md): 19.7 -> 9.0 us per addition, but 4x the lanes - it only pays on launches
// well below one wave per SIMD, which the MSM tails are not; not used by the library.
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/quad_add.hip -o /tmp/quad_add
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "ec.cuh"
using namespace lh;
namespace lh {
// ------------------------------------------------------------------ quad-cooperative arithmetic (device only)
// The tails of an MSM (continuation levels, bucket segments, window sums) are chains of DEPENDENT additions on
// launches that cannot fill the chip: a lone lane multiplies at ~1 us per Montgomery product, so one addition
// (14 products) costs ~14 us however idle the machine is.  Here the 4 lanes of a quad (lanes 4k..4k+3) hold the
// SAME operands and share the work: the products of one formula step run in different lanes and are exchanged with
// DPP quad_perm moves, 4 product-steps per addition and 3 per doubling instead of 14 and 9.  Every lane of the quad
// returns the full result, so a kernel maps one logical thread to a quad and keeps its control flow (which depends
// only on the replicated data) unchanged.
template <int K>
__device__ __forceinline__ Fq quad_bcast(const Fq& v) {
  Fq o;
#pragma unroll
  for (int i = 0; i < 8; i++)
    o.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], K * 0x55, 0xf, 0xf, false);
  return o;
}

// lane-dependent operand choice as mask arithmetic: (x & m) | (y & ~m) is one v_bfi_b32 (the ternary form is turned
// into a branch per limb)
__device__ __forceinline__ Fq quad_sel(int ql, const Fq& a, const Fq& b, const Fq& c, const Fq& d) {
  const uint32_t lo = 0u - (uint32_t)(ql & 1), hi = 0u - (uint32_t)((ql >> 1) & 1);
  Fq o;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t t0 = (b.l[i] & lo) | (a.l[i] & ~lo), t1 = (d.l[i] & lo) | (c.l[i] & ~lo);
    o.l[i] = (t1 & hi) | (t0 & ~hi);
  }
  return o;
}

__device__ __forceinline__ G1Xyzz dbl_quad(const G1Xyzz& p) {
  if (p.is_identity() || p.y.is_zero()) return G1Xyzz::identity();
  const int ql = (int)(threadIdx.x & 3u);
  const Fq u = dbl(p.y);
  Fq t = quad_sel(ql, u, p.x, u, u);
  Fq m = mul(t, t);                                              // 0: V = U^2   1: XX = X^2
  const Fq v = quad_bcast<0>(m), xx = quad_bcast<1>(m);
  const Fq mm = add(dbl(xx), xx);
  m = mul(quad_sel(ql, u, p.x, mm, v), quad_sel(ql, v, v, mm, p.zz));   // 0: W = U V  1: S = X V  2: M^2  3: V ZZ
  const Fq w = quad_bcast<0>(m), s_ = quad_bcast<1>(m), m2 = quad_bcast<2>(m);
  G1Xyzz r;
  r.zz = quad_bcast<3>(m);
  r.x = sub(m2, dbl(s_));
  m = mul(quad_sel(ql, mm, w, w, w), quad_sel(ql, sub(s_, r.x), p.y, p.zzz, p.zzz));  // 0: M (S - X3)  1: W Y  2: W ZZZ
  r.y = sub(quad_bcast<0>(m), quad_bcast<1>(m));
  r.zzz = quad_bcast<2>(m);
  return r;
}

__device__ __forceinline__ G1Xyzz add_quad(const G1Xyzz& p, const G1Xyzz& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  const int ql = (int)(threadIdx.x & 3u);
  Fq m = mul(quad_sel(ql, p.x, q.x, p.y, q.y), quad_sel(ql, q.zz, p.zz, q.zzz, p.zzz));
  const Fq u1 = quad_bcast<0>(m), u2 = quad_bcast<1>(m), s1 = quad_bcast<2>(m), s2 = quad_bcast<3>(m);
  const Fq pp_ = sub(u2, u1), r_ = sub(s2, s1);
  if (pp_.is_zero()) {
    if (r_.is_zero()) return dbl_quad(p);
    return G1Xyzz::identity();
  }
  m = mul(quad_sel(ql, pp_, r_, p.zz, p.zzz), quad_sel(ql, pp_, r_, q.zz, q.zzz));   // PP, RR, ZZ1 ZZ2, ZZZ1 ZZZ2
  const Fq pp = quad_bcast<0>(m), rr = quad_bcast<1>(m), zz12 = quad_bcast<2>(m), zzz12 = quad_bcast<3>(m);
  m = mul(quad_sel(ql, pp_, u1, zz12, zz12), pp);                                        // PPP, Q, ZZ3
  const Fq ppp = quad_bcast<0>(m), qq = quad_bcast<1>(m);
  G1Xyzz r;
  r.zz = quad_bcast<2>(m);
  r.x = sub(sub(rr, ppp), dbl(qq));
  m = mul(quad_sel(ql, r_, s1, zzz12, zzz12), quad_sel(ql, sub(qq, r.x), ppp, ppp, ppp));
  r.y = sub(quad_bcast<0>(m), quad_bcast<1>(m));
  r.zzz = quad_bcast<2>(m);
  return r;
}
}  // namespace lh
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool QUAD>
__global__ __launch_bounds__(64) void chain(G1Xyzz* out, int n, int variant) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned id = QUAD ? gid >> 2 : gid;
  G1Affine g;
  g.x = Fq::one();
  g.y = add(Fq::one(), Fq::one());
  G1Xyzz a = G1Xyzz::from_affine(g), b = dbl(a);
  for (unsigned k = 0; k < (id & 7u); k++) b = add(b, a);  // different start per chain
  for (int i = 0; i < n; i++) {
    G1Xyzz c;
    if (variant == 1 && (i & 15) == 7) c = QUAD ? add_quad(b, b) : add(b, b);          // hits the doubling branch
    else if (variant == 1 && (i & 15) == 11) c = QUAD ? add_quad(b, G1Xyzz::identity()) : add(b, G1Xyzz::identity());
    else c = QUAD ? add_quad(a, b) : add(a, b);
    a = b;
    b = c;
  }
  if (!QUAD || (gid & 3u) == 0) out[id] = b;
}

// timing reference: `per` dependent multiplications per iteration, nothing else
__global__ __launch_bounds__(64) void mul_only(Fq* out, int n, int per) {
  Fq a = add(Fq::one(), Fq::one()), b = a;
  for (unsigned k = 0; k < (threadIdx.x & 7u); k++) b = add(b, a);
  for (int i = 0; i < n * per; i++) b = mul(b, a);
  out[blockIdx.x * blockDim.x + threadIdx.x] = b;
}
// the quad exchange alone: 13 broadcasts + 8 selects + 8 additions per iteration, no multiplication
__global__ __launch_bounds__(64) void glue_only(Fq* out, int n) {
  const int ql = threadIdx.x & 3;
  Fq a = add(Fq::one(), Fq::one()), b = a, c = a, d = a;
  for (unsigned k = 0; k < (threadIdx.x & 7u); k++) b = add(b, a);
  for (int i = 0; i < n; i++) {
    for (int s = 0; s < 4; s++) {
      Fq m = sub(quad_sel(ql, a, b, c, d), quad_sel(ql, b, c, d, a));
      a = quad_bcast<0>(m), b = quad_bcast<1>(m), c = quad_bcast<2>(m);
      d = add(a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = add(add(a, b), add(c, d));
}

int main() {
  const int blocks = 1024, n = 64;
  G1Xyzz *d1, *d4;
  CK(hipMalloc(&d1, sizeof(G1Xyzz) * blocks * 64));
  CK(hipMalloc(&d4, sizeof(G1Xyzz) * blocks * 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int variant = 0; variant < 2; variant++) {
    float ms1 = 0, ms4 = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(chain<false>, blocks, 64, 0, 0, d1, n, variant);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms1, e0, e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(chain<true>, blocks, 64, 0, 0, d4, n, variant);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms4, e0, e1));
    }
    std::vector<G1Xyzz> h1(blocks * 16), h4(blocks * 16);
    CK(hipMemcpy(h1.data(), d1, sizeof(G1Xyzz) * blocks * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h4.data(), d4, sizeof(G1Xyzz) * blocks * 16, hipMemcpyDeviceToHost));
    const bool same = memcmp(h1.data(), h4.data(), sizeof(G1Xyzz) * blocks * 16) == 0;
    printf("variant %d: %d dependent additions, one wave per SIMD: lone lanes %.1f us per addition, quads %.1f us; results %s\n",
           variant, n, ms1 * 1e3 / n, ms4 * 1e3 / n, same ? "identical" : "DIFFER");
    if (!same) return 2;
  }
  Fq* dm;
  CK(hipMalloc(&dm, sizeof(Fq) * blocks * 64));
  for (int per : {4, 14}) {
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(mul_only, blocks, 64, 0, 0, dm, n, per);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%d dependent multiplications per iteration: %.1f us per iteration\n", per, ms * 1e3 / n);
  }
  {
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(glue_only, blocks, 64, 0, 0, dm, n);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("quad exchange alone (4 x [2 selects, sub, 3 broadcasts, add]): %.1f us per iteration\n", ms * 1e3 / n);
  }
  return 0;
}
