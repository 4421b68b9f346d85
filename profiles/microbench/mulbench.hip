#include "fp28_proto.hip.h"
#include <cstdio>
#include <vector>
// variant B: two accumulators (a*b and m*p chains separate)
__device__ __forceinline__ void fp_mul2(Fp& r, const Fp& a, const Fp& b) {
  uint64_t acc = 0, acc2 = 0; uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FQ.p[k - i];
    acc += acc2;
    m[k] = ((uint32_t)acc * FQ.inv) & MASK;
    acc += (uint64_t)m[k] * FQ.p[0];
    acc >>= LB; acc2 = 0;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FQ.p[k - i];
    acc += acc2; acc2 = 0;
    r.l[k - NL] = (uint32_t)acc & MASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}
template <int V, int WPS>
__global__ void __launch_bounds__(256 * (WPS > 4 ? 4 : WPS)) k_chain(Fp* out, const Fp* in, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  Fp x = in[i], y = in[i + 1];
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    Fp r;
    if (V == 0) fp_mul(r, x, y); else fp_mul2(r, x, y);
    y = x; x = r;
  }
  out[i] = x;
}
template <int V, int WPS>
void run(Fp* d_out, Fp* d_in, int iters) {
  int threads = 256 * (WPS > 4 ? 4 : WPS);
  int blocks = 256 * (WPS > 4 ? WPS / 4 : 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_chain<V, WPS>), dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<V, WPS>), dim3(blocks), dim3(threads), 0, 0, d_out, d_in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double muls = (double)blocks * threads * iters;
  printf("variant=%d waves/SIMD=%d  %.3f ms  %.2f Gmul/s  ns/mul/SIMD=%.1f  cycles@2.4GHz per mad=%.2f\n", V, WPS, ms, muls / ms * 1e-6,
         ms * 1e6 / ((double)iters * WPS), ms * 1e-3 * 2.4e9 / ((double)iters * WPS * 1458));
}
int main() {
  FpParams h; for (int i = 0; i < NL; ++i) h.p[i] = 0x0abcdef1u + i * 977; h.inv = 0x0fffffffu;
  hipMemcpyToSymbol(HIP_SYMBOL(FQ), &h, sizeof(h));
  size_t n = 256 * 2048 + 8;
  std::vector<Fp> hin(n);
  for (size_t i = 0; i < n; ++i) for (int j = 0; j < NL; ++j) hin[i].l[j] = (uint32_t)(i * 2654435761u + j * 40503u) & MASK;
  Fp *d_in, *d_out; hipMalloc(&d_in, n * sizeof(Fp)); hipMalloc(&d_out, n * sizeof(Fp));
  hipMemcpy(d_in, hin.data(), n * sizeof(Fp), hipMemcpyHostToDevice);
  int iters = 2000;
  run<0, 1>(d_out, d_in, iters); run<0, 2>(d_out, d_in, iters); run<0, 4>(d_out, d_in, iters); run<0, 8>(d_out, d_in, iters);
  run<1, 1>(d_out, d_in, iters); run<1, 2>(d_out, d_in, iters); run<1, 4>(d_out, d_in, iters); run<1, 8>(d_out, d_in, iters);
  return 0;
}
