// Development experiment (round 3): what a single wave per SIMD can issue, by multiplier formulation.
// The level kernels run ONE wave per SIMD; tools/experiments/mul_occupancy.hip showed the product-scanning multiplier at
// 18.3 G products/s there against 22.0 at eight waves: the serial tail of every column (add, mul_lo, and, mad, shift -- five
// dependent instructions) leaves only the m*p chain of the next column to fill the pipeline.  Variants:
//   0  fp_mul of the product (two accumulators, the a*b chain continues on the carry)
//   1  three accumulators: a fresh a*b sum per column, the carry chain separate (one more 64-bit add per column)
//   2  variant 1 written software-pipelined (the a*b sum of column k+1 ahead of the serial tail of column k)
//   3  four chains (a*b and m*p each split in two)
//   4  variant 1 with a signed product half (v_mad_i64_i32): operands with signed limbs, e.g. limb-wise differences
//   5  variant 0 followed by fp_sub            6  variant 4 followed by the lazy subtraction (limb-wise, no carries) and
//                                                 one fp_norm per two products (what a pairing slot needs)
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../snark-challenge-prover-reference_amd/csrc mul_variants.hip -o /tmp/mul_var && /tmp/mul_var
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fp753.hip.h"
using namespace mnt753;

// keeps LLVM from re-associating a separately accumulated sum back into the carry chain (it otherwise starts the a*b chain of a
// column from the carry of the previous one: every variant then compiles to the code of fp_mul)
__device__ __forceinline__ void opaque(uint64_t& x) { asm("" : "+v"(x)); }
__device__ __forceinline__ void opaque(int64_t& x) { asm("" : "+v"(x)); }

template <int M>
__device__ __forceinline__ void mul_v1(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  uint64_t carry = 0;
  uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    uint64_t ab = 0, mp = 0;
#pragma unroll
    for (int i = 0; i <= k; ++i) ab += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
    opaque(ab);
    uint64_t t = carry + mp;
    t += ab;
    m[k] = ((uint32_t)t * FPC[M].inv) & LMASK;
    t += (uint64_t)m[k] * FPC[M].p[0];
    carry = t >> LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
    uint64_t ab = 0, mp = 0;
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) ab += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
    opaque(ab);
    uint64_t t = carry + mp;
    t += ab;
    r.l[k - NL] = (uint32_t)t & LMASK;
    carry = t >> LB;
  }
  r.l[NL - 1] = (uint32_t)carry;
}

template <int M>
__device__ __forceinline__ void mul_v2(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  uint64_t carry = 0;
  uint32_t m[NL];
  uint64_t ab = (uint64_t)a.l[0] * b.l[0];
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
    // a*b sum of the NEXT column first: it depends on nothing the serial tail below produces
    uint64_t abn = 0;
    if (k + 1 < 2 * NL - 1) {
      const int kk = k + 1, lo = kk < NL ? 0 : kk - NL + 1, hi = kk < NL ? kk : NL - 1;
#pragma unroll
      for (int i = lo; i <= hi; ++i) abn += (uint64_t)a.l[i] * b.l[kk - i];
    }
    uint64_t mp = 0;
    if (k < NL) {
#pragma unroll
      for (int i = 0; i < k; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
      opaque(ab);
      uint64_t t = carry + mp;
      t += ab;
      m[k] = ((uint32_t)t * FPC[M].inv) & LMASK;
      t += (uint64_t)m[k] * FPC[M].p[0];
      carry = t >> LB;
    } else {
#pragma unroll
      for (int i = k - NL + 1; i < NL; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
      opaque(ab);
      uint64_t t = carry + mp;
      t += ab;
      r.l[k - NL] = (uint32_t)t & LMASK;
      carry = t >> LB;
    }
    ab = abn;
  }
  r.l[NL - 1] = (uint32_t)carry;
}

template <int M>
__device__ __forceinline__ void mul_v3(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  uint64_t carry = 0;
  uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    uint64_t ab0 = 0, ab1 = 0, mp0 = 0, mp1 = 0;
#pragma unroll
    for (int i = 0; i <= k; ++i) { if (i & 1) ab1 += (uint64_t)a.l[i] * b.l[k - i]; else ab0 += (uint64_t)a.l[i] * b.l[k - i]; }
#pragma unroll
    for (int i = 0; i < k; ++i) { if (i & 1) mp1 += (uint64_t)m[i] * FPC[M].p[k - i]; else mp0 += (uint64_t)m[i] * FPC[M].p[k - i]; }
    opaque(ab0); opaque(ab1); opaque(mp1);
    uint64_t t = carry + mp0;
    t += ab0; t += ab1; t += mp1;
    m[k] = ((uint32_t)t * FPC[M].inv) & LMASK;
    t += (uint64_t)m[k] * FPC[M].p[0];
    carry = t >> LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
    uint64_t ab0 = 0, ab1 = 0, mp0 = 0, mp1 = 0;
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) { if (i & 1) ab1 += (uint64_t)a.l[i] * b.l[k - i]; else ab0 += (uint64_t)a.l[i] * b.l[k - i]; }
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) { if (i & 1) mp1 += (uint64_t)m[i] * FPC[M].p[k - i]; else mp0 += (uint64_t)m[i] * FPC[M].p[k - i]; }
    opaque(ab0); opaque(ab1); opaque(mp1);
    uint64_t t = carry + mp0;
    t += ab0; t += ab1; t += mp1;
    r.l[k - NL] = (uint32_t)t & LMASK;
    carry = t >> LB;
  }
  r.l[NL - 1] = (uint32_t)carry;
}

// signed product half: limbs of a and b are int32 (|a_i| |b_j| 27 + 27 2^56 < 2^63), the value may be negative; the result
// has limbs 0..25 in [0, 2^28) and a signed top limb, value in (-|ab|/R', |ab|/R' + p)
template <int M>
__device__ __forceinline__ void mul_v4(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  int64_t carry = 0;
  uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    int64_t ab = 0;
    uint64_t mp = 0;
#pragma unroll
    for (int i = 0; i <= k; ++i) ab += (int64_t)(int32_t)a.l[i] * (int32_t)b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
    opaque(ab);
    int64_t t = carry + (int64_t)mp;
    t += ab;
    m[k] = ((uint32_t)t * FPC[M].inv) & LMASK;
    t += (int64_t)((uint64_t)m[k] * FPC[M].p[0]);
    carry = t >> LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
    int64_t ab = 0;
    uint64_t mp = 0;
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) ab += (int64_t)(int32_t)a.l[i] * (int32_t)b.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
    opaque(ab);
    int64_t t = carry + (int64_t)mp;
    t += ab;
    r.l[k - NL] = (uint32_t)t & LMASK;
    carry = t >> LB;
  }
  r.l[NL - 1] = (uint32_t)carry;
}


// variant 7: one level of subtractive Karatsuba on the product half (14 + 13 limbs): a0 b0 (196 MADs), a1 b1 (169), and
// (a0 - a1)(b1 - b0) (196, signed) -- 561 multiply-adds instead of 729; a0 b1 + a1 b0 = (a0 - a1)(b1 - b0) + a0 b0 + a1 b1 is
// assembled per column with 64-bit adds.  The reduction half is unchanged.  Same instruction count as the schoolbook product,
// 168 fewer multiplier operations: pays only where the chip is power-limited, not issue-limited.
template <int M>
__device__ __forceinline__ void mul_v7(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
  constexpr int H = 14, L2 = NL - H;            // low half 14 limbs, high half 13
  int32_t da[H], db[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    da[i] = (int32_t)a.l[i] - (i < L2 ? (int32_t)a.l[H + i] : 0);
    db[i] = (i < L2 ? (int32_t)b.l[H + i] : 0) - (int32_t)b.l[i];
  }
  int64_t carry = 0;
  uint32_t m[NL];
  // columns of the three partial products: lo_k (k < 27), hi_k (k < 25), mid_k (k < 27, signed)
  uint64_t lo[2 * H - 1], hi[2 * L2 - 1];
  int64_t md[2 * H - 1];
#pragma unroll
  for (int k = 0; k < 2 * H - 1; ++k) {
    uint64_t s1 = 0; int64_t s2 = 0;
#pragma unroll
    for (int i = (k < H ? 0 : k - H + 1); i <= (k < H ? k : H - 1); ++i) { s1 += (uint64_t)a.l[i] * b.l[k - i]; s2 += (int64_t)da[i] * db[k - i]; }
    lo[k] = s1; md[k] = s2;
  }
#pragma unroll
  for (int k = 0; k < 2 * L2 - 1; ++k) {
    uint64_t s1 = 0;
#pragma unroll
    for (int i = (k < L2 ? 0 : k - L2 + 1); i <= (k < L2 ? k : L2 - 1); ++i) s1 += (uint64_t)a.l[H + i] * b.l[H + k - i];
    hi[k] = s1;
  }
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
    // product column k = lo_k + [mid_(k-14) + lo_(k-14) + hi_(k-14)] + hi_(k-28)
    int64_t ab = 0;
    if (k < 2 * H - 1) ab += (int64_t)lo[k];
    if (k >= H && k - H < 2 * H - 1) { ab += md[k - H] + (int64_t)lo[k - H]; if (k - H < 2 * L2 - 1) ab += (int64_t)hi[k - H]; }
    if (k >= 2 * H && k - 2 * H < 2 * L2 - 1) ab += (int64_t)hi[k - 2 * H];
    uint64_t mp = 0;
    if (k < NL) {
#pragma unroll
      for (int i = 0; i < k; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
      int64_t t = carry + (int64_t)mp + ab;
      m[k] = ((uint32_t)t * FPC[M].inv) & LMASK;
      t += (int64_t)((uint64_t)m[k] * FPC[M].p[0]);
      carry = t >> LB;
    } else {
#pragma unroll
      for (int i = k - NL + 1; i < NL; ++i) mp += (uint64_t)m[i] * FPC[M].p[k - i];
      int64_t t = carry + (int64_t)mp + ab;
      r.l[k - NL] = (uint32_t)t & LMASK;
      carry = t >> LB;
    }
  }
  r.l[NL - 1] = (uint32_t)carry;
}

// limb-wise difference, no carries: limbs in (-2^28, 2^28) for operands with 28-bit limbs
template <int M>
__device__ __forceinline__ void sub_raw(Fp<M>& r, const Fp<M>& a, const Fp<M>& b) {
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] - b.l[i];
}
// signed, un-normalised limbs (|limb| < 2^30, |value| < 4.5p) -> limbs 0..25 in [0, 2^28), value in about [0.5p, 1.5p):
// the quotient comes from the top limb (the lower limbs move it by less than 2^-20 p), one pass carries and subtracts q p
template <int M>
__device__ __forceinline__ void fp_norm(Fp<M>& r, const Fp<M>& a) {
  const float qf = floorf((float)(int32_t)a.l[NL - 1] * (1.0f / (float)FPC[M].p[NL - 1]) - 0.5f);
  const int32_t q = (int32_t)qf;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int32_t t = (int32_t)a.l[i] - q * (int32_t)FPC[M].p[i] + c;
    if (i < NL - 1) { r.l[i] = (uint32_t)t & LMASK; c = t >> LB; } else r.l[i] = (uint32_t)t;
  }
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k_mul(uint32_t* p, int reps) {
  extern __shared__ uint4 lds[];
  Fp<1> a, b, c;
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 64;
  for (int i = 0; i < NL; ++i) { a.l[i] = p[base + i] & LMASK; b.l[i] = p[base + 32 + i] & LMASK; }
#pragma nounroll
  for (int r = 0; r < reps; ++r) {
    if (VARIANT == 0) { fp_mul(c, a, b); fp_mul(a, c, b); }
    if (VARIANT == 1) { mul_v1(c, a, b); mul_v1(a, c, b); }
    if (VARIANT == 2) { mul_v2(c, a, b); mul_v2(a, c, b); }
    if (VARIANT == 3) { mul_v3(c, a, b); mul_v3(a, c, b); }
    if (VARIANT == 4) { mul_v4(c, a, b); mul_v4(a, c, b); }
    if (VARIANT == 5) { fp_mul(c, a, b); fp_sub(a, c, b); fp_mul(b, a, c); fp_sub(b, b, a); }
    if (VARIANT == 7) { mul_v7(c, a, b); mul_v7(a, c, b); }
    if (VARIANT == 6) { Fp<1> d; mul_v4(c, a, b); sub_raw(d, c, b); mul_v4(b, d, c); sub_raw(d, b, a); fp_norm(a, d); }
  }
  if (threadIdx.x == 9999) lds[0] = make_uint4(a.l[0], 0, 0, 0);
  for (int i = 0; i < NL; ++i) p[base + i] = a.l[i] ^ b.l[i];
}

template <int V>
static void run(const char* name, uint32_t* d, int waves_per_simd) {
  const size_t lds = waves_per_simd == 1 ? 100 * 1024 : waves_per_simd == 2 ? 64 * 1024 : waves_per_simd == 4 ? 36 * 1024 : 16 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mul<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = 256 * waves_per_simd, reps = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mul<V>, dim3(blocks), dim3(256), lds, 0, d, 10);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mul<V>, dim3(blocks), dim3(256), lds, 0, d, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double ops = (double)blocks * 256 * reps * 2;
  printf("variant %d %-44s waves/SIMD %d  %8.3f ms  %7.2f G products/s  (%s)\n", V, name, waves_per_simd, best, ops / best / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
  uint32_t* d; const size_t n = (size_t)256 * 8 * 256 * 64;
  hipMalloc(&d, n * 4);
  std::vector<uint32_t> h(n); for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u) >> 4;
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int w : {1, 2, 4, 8}) {
    run<0>("product fp_mul (2 accumulators)", d, w);
    run<1>("fresh a*b sum per column (3 accumulators)", d, w);
    run<2>("3 accumulators, software-pipelined source", d, w);
    run<3>("4 chains", d, w);
    run<4>("3 accumulators, signed product half", d, w);
    run<5>("fp_mul + fp_sub", d, w);
    run<6>("signed mul + lazy sub, fp_norm per 2 products", d, w);
    run<7>("Karatsuba product half (561 + 729 MADs)", d, w);
  }
  return 0;
}
