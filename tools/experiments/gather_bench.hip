// Development microbenchmark (GPU box): how fast can one wave per SIMD gather random 112-byte rows of a table far larger
// than the caches?  Variants:  A lane-per-row (7 x global_load_dwordx4 per lane, 64 rows per instruction);
//                              B row-cooperative LDS-DMA (global_load_lds_dwordx4: 9 rows x 7 quads per instruction) + ds_read
//                              C as B but two row sets in flight (prefetch one iteration ahead)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/gather_bench.hip -o build/gather_bench && ./build/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ROW_B = 224;            // table row (x | y), gather x = first 112 bytes
__global__ void __launch_bounds__(256, 1) k_A(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, int iters, int nl, uint32_t* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t r0 = idx[(size_t)(2 * it) * nl + t], r1 = idx[(size_t)(2 * it + 1) * nl + t];
    const uint4* p0 = table + (size_t)r0 * (ROW_B / 16);
    const uint4* p1 = table + (size_t)r1 * (ROW_B / 16);
#pragma unroll
    for (int q = 0; q < 7; ++q) { uint4 a = p0[q], b = p1[q]; acc += a.x ^ b.y ^ a.z ^ b.w; }
    // stand-in for the arithmetic of a slot: keeps the loop from overlapping iterations for free
    for (int k = 0; k < 64; ++k) acc = acc * 1664525u + 1013904223u;
  }
  out[t] = acc;
}
// row-cooperative: the wave's 128 rows (2 per lane) are fetched 9 rows per instruction: lane L of instruction k fetches quad (i % 7)
// of row (i / 7), i = 63 k + L (lane 63 idles) -> LDS image [row][7 quads] packed (112-byte stride: conflict-free ds_read_b128)
template <int DEPTH>
__global__ void __launch_bounds__(256, 1) k_B(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, int iters, int nl, uint32_t* out) {
  extern __shared__ uint4 lds[];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROWS = 128, QUADS = ROWS * 7, NINS = (QUADS + 62) / 63;   // 896 quads, 15 instructions of 63 lanes
  uint4* img = lds + (size_t)wave * DEPTH * (NINS * 64);                   // per wave: DEPTH images of NINS * 64 quads
  uint32_t* rows = reinterpret_cast<uint32_t*>(lds + (size_t)4 * DEPTH * (NINS * 64)) + wave * DEPTH * ROWS;
  uint32_t acc = 0;
  auto issue = [&](int it, int buf) {
    // row indices of the wave for iteration `it` -> LDS (so that every lane can look up the row its quad belongs to)
    rows[buf * ROWS + lane] = idx[(size_t)(2 * it) * nl + t];
    rows[buf * ROWS + 64 + lane] = idx[(size_t)(2 * it + 1) * nl + t];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the row indices are in LDS (same wave: no barrier needed)
#pragma unroll
    for (int k = 0; k < NINS; ++k) {
      const int i = 63 * k + (lane < 63 ? lane : 62);
      const int row = i / 7, q = i % 7;
      const uint32_t r = rows[buf * ROWS + (row < ROWS ? row : 0)];
      const uint4* src = table + (size_t)r * (ROW_B / 16) + q;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(img + (size_t)buf * (NINS * 64) + k * 64), 16, 0, 0);
    }
  };
  for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
  for (int it = 0; it < iters; ++it) {
    const int buf = it % DEPTH;
    if (DEPTH > 1) { if (it + DEPTH - 1 < iters) issue(it + DEPTH - 1, (it + DEPTH - 1) % DEPTH); } else issue(it, 0);
    // wait for the image of THIS iteration: everything but the (DEPTH - 1) younger images' instructions
    if (DEPTH == 1 || it + DEPTH - 1 >= iters) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0f70 | ((NINS + 2) & 15) | ((((NINS + 2) >> 4) & 3) << 14));   // vmcnt(NINS + 2 index loads)
    // lane L owns rows L and 64 + L: image position of row r quad q with the 63-lane packing: i = 7 r + q -> instruction i / 63, lane i % 63
    const uint4* im = img + (size_t)buf * (NINS * 64);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int i0 = 7 * lane + q, i1 = 7 * (64 + lane) + q;
      uint4 a = im[(i0 / 63) * 64 + i0 % 63], b = im[(i1 / 63) * 64 + i1 % 63];
      acc += a.x ^ b.y ^ a.z ^ b.w;
    }
    for (int k = 0; k < 64; ++k) acc = acc * 1664525u + 1013904223u;
  }
  out[t] = acc;
}

int main(int argc, char** argv) {
  const size_t n_rows = (size_t)38 << 20;    // 38 * 2^20 rows of 224 B = 8.9 GB (the window table of a 2^20-point G1 set)
  const int nl = 65536, iters = 160;
  uint4* table; uint32_t *idx, *out;
  CK(hipMalloc(&table, n_rows * ROW_B));
  CK(hipMemset(table, 1, n_rows * ROW_B));
  std::vector<uint32_t> h((size_t)2 * iters * nl);
  uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % n_rows); }
  CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, nl * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double rows = 2.0 * iters * nl;
    printf("%-44s %8.3f ms   %6.1f M rows/s   %6.2f TB/s useful (112 B/row)   %.2f us per wave-iteration\n", name, ms, rows / ms / 1e3, rows * 112 / ms / 1e9,
           ms * 1e3 / iters);
  };
  run("A lane-per-row, 7 x dwordx4", [&] { hipLaunchKernelGGL(k_A, dim3(nl / 256), dim3(256), 0, 0, table, idx, iters, nl, out); });
  const size_t lds1 = (size_t)4 * 1 * (15 * 64) * 16 + 4 * 1 * 128 * 4, lds2 = (size_t)4 * 2 * (15 * 64) * 16 + 4 * 2 * 128 * 4, lds3 = (size_t)4 * 3 * (15 * 64) * 16 + 4 * 3 * 128 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_B<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_B<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  run("B row-cooperative LDS-DMA, depth 1", [&] { hipLaunchKernelGGL(k_B<1>, dim3(nl / 256), dim3(256), lds1, 0, table, idx, iters, nl, out); });
  run("C row-cooperative LDS-DMA, depth 2", [&] { hipLaunchKernelGGL(k_B<2>, dim3(nl / 256), dim3(256), lds2, 0, table, idx, iters, nl, out); });
  return 0;
}
