// Development microbenchmark (GPU box): rocPRIM radix sort of the MSM's (bucket, entry) pairs vs the atomic counting sort.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/sort_bench.hip -o build/sort_bench && ./build/sort_bench
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_fill(uint32_t* keys, uint32_t* vals, size_t n, uint32_t mask) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t s = i * 0x9E3779B97F4A7C15ull + 12345; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
  keys[i] = (uint32_t)s & mask; vals[i] = (uint32_t)i;
}
int main() {
  const size_t n = (size_t)38 << 20;
  uint32_t *k0, *k1, *v0, *v1;
  CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
  hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, k0, v0, n, (1u << 19) - 1u);
  size_t tmp_bytes = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, 19));
  void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
  printf("temp storage %.1f MB\n", tmp_bytes / 1e6);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int bits : {19, 16, 24}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("radix_sort_pairs %zu pairs, %d key bits: %.3f ms\n", n, bits, ms);
    }
  }
  return 0;
}
