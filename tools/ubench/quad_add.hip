// Development micro-benchmark: a chain of dependent G1 additions by lone lanes (ec.cuh add) against a
// quad-cooperative addition (4 lanes share the products of one formula step), one wave per SIMD.  Checks that both give
// the same bytes.  Result (profiles/README.md): 19.7 -> 9.0 us per addition at 4x the lanes: it pays on launches well
// below one wave per SIMD (the late continuation levels and small bucket reductions of an MSM, csrc/msm.hip).
// build: hipcc -O3 --offload-arch=gfx950 -I halo2-lasso_amd/csrc tools/ubench/quad_add.hip -o /tmp/quad_add
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "ec.cuh"
using namespace lh;
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
