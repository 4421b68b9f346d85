// Development experiment: throughput of the 753-bit Montgomery product as a function of waves per SIMD.
// The pairing levels run ONE wave per SIMD (registers + 36 KB of LDS per wave); this measures what a single wave can issue.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../snark-challenge-prover-reference_amd/csrc mul_occupancy.hip -o /tmp/mul_occ && /tmp/mul_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fp753.hip.h"
using namespace mnt753;

template <int VARIANT>
__global__ void __launch_bounds__(256) k_mul(uint32_t* p, int reps) {
  extern __shared__ uint4 lds[];
  Fp<1> a, b, c;
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 64;
  for (int i = 0; i < NL; ++i) { a.l[i] = p[base + i] & LMASK; b.l[i] = p[base + 32 + i] & LMASK; }
  for (int r = 0; r < reps; ++r) {
    if (VARIANT == 0) { fp_mul(c, a, b); fp_mul(a, c, b); }                       // dependent products
    if (VARIANT == 1) { fp_mul(c, a, b); fp_sub(a, c, b); fp_mul(b, a, c); fp_sub(b, b, a); }   // products + subtractions
    if (VARIANT == 2) { fp_sqr(c, a); fp_sqr(a, c); }
  }
  if (threadIdx.x == 9999) lds[0] = make_uint4(a.l[0], 0, 0, 0);
  for (int i = 0; i < NL; ++i) p[base + i] = a.l[i] ^ b.l[i];
}

template <int V>
static void run(const char* name, uint32_t* d, int waves_per_simd, double ops_per_rep) {
  // occupancy through LDS: 160 KB per CU; one 256-thread block = one wave on each SIMD
  const size_t lds = waves_per_simd == 1 ? 100 * 1024 : waves_per_simd == 2 ? 64 * 1024 : waves_per_simd == 4 ? 36 * 1024 : 16 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mul<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = 256 * waves_per_simd, reps = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mul<V>, dim3(blocks), dim3(256), lds, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_mul<V>, dim3(blocks), dim3(256), lds, 0, d, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 256 * reps * ops_per_rep;
  printf("%-28s waves/SIMD %d  %8.3f ms  %7.2f G ops/s  (%s)\n", name, waves_per_simd, ms, ops / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
  uint32_t* d; const size_t n = (size_t)256 * 8 * 256 * 64;
  hipMalloc(&d, n * 4);
  std::vector<uint32_t> h(n); for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u) >> 4;
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int w : {1, 2, 4, 8}) run<0>("mul chain (products)", d, w, 2);
  for (int w : {1, 2, 4}) run<1>("mul+sub (products)", d, w, 2);
  for (int w : {1, 2, 4}) run<2>("sqr chain (squarings)", d, w, 2);
  return 0;
}
