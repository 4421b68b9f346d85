// Development microbenchmark (GPU box): divstep inversion (fp_inv.hip.h) vs the Fermat chain, one element per lane,
// one wave per SIMD over the whole chip.  Checks x * x^-1 = 1 and that both routines agree, then times each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/dev_inv_gpu.hip -o build/dev_inv_gpu && ./build/dev_inv_gpu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../snark-challenge-prover-reference_amd/csrc/msm_kernels.hip.h"
using namespace mnt753;

namespace mnt753 {
// Fermat chain x^(p-2): the cross-check of the divstep inversion (moved here from the product in round 2)
template <int M>
__device__ void fp_inv_fermat(Fp<M>& r, const Fp<M>& x) {
  // x^(p-2), exponent limbs from the constants table
  Fp<M> acc, a, b, t;
  fp_one(acc);
  bool started = false;
#pragma unroll 1
  for (int i = NL * LB - 1; i >= 0; --i) {
    const uint32_t limb = FPC[M].pm2[i / LB];
    const bool bit = (limb >> (i % LB)) & 1u;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
      if (phase == 0) { if (!started) continue; a = acc; b = acc; }
      else { if (!bit) continue; a = acc; b = x; started = true; }
      fp_mul(t, a, b);
      acc = t;
    }
  }
  r = acc;
}
}  // namespace mnt753

template <int M>
__global__ void __launch_bounds__(256, 1) k_make(uint32_t* x, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fp<M> a, one, b;
  fp_one(one);
  for (int i = 0; i < NL; ++i) a.l[i] = (uint32_t)(0x9e3779b9u * (uint32_t)(t * 31 + i + 1)) & LMASK;
  a.l[NL - 1] &= 0xfff;
  fp_mul(b, a, one);                 // some element in [0, 2p)
  if (t == 0) b = one;
  fp_store(x + (size_t)t * FPS_WORDS, b);
}
template <int M, int WHICH>
__global__ void __launch_bounds__(256, 1) k_inv(const uint32_t* x, uint32_t* out, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fp<M> a, r;
  fp_load(a, x + (size_t)t * FPS_WORDS);
  if (WHICH == 0) fp_inv(r, a); else fp_inv_fermat(r, a);
  fp_store(out + (size_t)t * FPS_WORDS, r);
}
template <int M>
__global__ void __launch_bounds__(256, 1) k_check(const uint32_t* x, const uint32_t* a, const uint32_t* b, int* bad, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fp<M> xx, ia, ib, p, one, c1, c2;
  fp_load(xx, x + (size_t)t * FPS_WORDS); fp_load(ia, a + (size_t)t * FPS_WORDS); fp_load(ib, b + (size_t)t * FPS_WORDS);
  fp_one(one);
  fp_mul(p, xx, ia);
  fp_canon(c1, p); fp_canon(c2, one);
  bool ok = true;
  for (int i = 0; i < NL; ++i) ok = ok && c1.l[i] == c2.l[i];
  fp_canon(c1, ia); fp_canon(c2, ib);
  for (int i = 0; i < NL; ++i) ok = ok && c1.l[i] == c2.l[i];
  if (!ok) atomicAdd(bad, 1);
}
template <int M> int run(const char* name) {
  const int n = 65536;     // one wave per SIMD on 256 CUs
  uint32_t *x, *a, *b; int* bad;
  hipMalloc(&x, (size_t)n * FPS_WORDS * 4); hipMalloc(&a, (size_t)n * FPS_WORDS * 4); hipMalloc(&b, (size_t)n * FPS_WORDS * 4);
  hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((k_make<M>), dim3(n / 256), dim3(256), 0, 0, x, n);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  hipLaunchKernelGGL((k_inv<M, 0>), dim3(n / 256), dim3(256), 0, 0, x, a, n);   // warm
  hipLaunchKernelGGL((k_inv<M, 1>), dim3(n / 256), dim3(256), 0, 0, x, b, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_inv<M, 0>), dim3(n / 256), dim3(256), 0, 0, x, a, n);
  hipEventRecord(e1);
  hipLaunchKernelGGL((k_inv<M, 1>), dim3(n / 256), dim3(256), 0, 0, x, b, n);
  hipEventRecord(e2);
  hipLaunchKernelGGL((k_check<M>), dim3(n / 256), dim3(256), 0, 0, x, a, b, bad, n);
  hipDeviceSynchronize();
  float t_div, t_fer; hipEventElapsedTime(&t_div, e0, e1); hipEventElapsedTime(&t_fer, e1, e2);
  int hbad = -1; hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
  printf("%s: %d inversions, one per lane: divsteps %.3f ms, Fermat %.3f ms (%.1fx), mismatches %d\n", name, n, t_div, t_fer, t_fer / t_div, hbad);
  return hbad;
}
int main() {
  int bad = run<0>("modulus A") + run<1>("modulus B");
  printf(bad ? "FAIL\n" : "OK\n");
  return bad != 0;
}
