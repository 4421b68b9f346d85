// hipcc 7.2 / gfx950, round 3 (performance, not correctness).  A Montgomery product whose product half is written with signed
// 64-bit multiply-adds, (int64)(int32)a[i] * (int32)b[j] + acc, where the compiler can prove ONE operand non-negative (limbs masked
// to 28 bits): instruction selection treats it as sext x zext, which has no single instruction, and emits two v_mad_u64_u32 plus two
// v_mov_b32 per product.  With both operands opaque (empty asm) every product is one v_mad_i64_i32.
//   expected: about 729 + 729 multiply-adds for k_known          actual: see tools/compiler_repro/check.sh (static count, no GPU needed)
#include <hip/hip_runtime.h>
#include <cstdint>
constexpr int NL = 27, LB = 28;
constexpr uint32_t LMASK = 0xfffffffu;
struct P { uint32_t p[NL]; uint32_t inv; };
__constant__ P FQ;
template <bool OPAQUE>
__device__ __forceinline__ void mul_s(uint32_t (&r)[NL], const uint32_t (&a_in)[NL], const uint32_t (&b_in)[NL]) {
  int32_t a[NL], b[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t x = a_in[i], y = b_in[i];
    if (OPAQUE) { asm("" : "+v"(x)); asm("" : "+v"(y)); }
    a[i] = (int32_t)x; b[i] = (int32_t)y;
  }
  int64_t acc = 0; uint64_t acc2 = 0; uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)a[i] * b[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * FQ.p[k - i];
    acc += (int64_t)acc2; acc2 = 0;
    m[k] = ((uint32_t)acc * FQ.inv) & LMASK;
    acc += (int64_t)((uint64_t)m[k] * FQ.p[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (int64_t)a[i] * b[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc2 += (uint64_t)m[i] * FQ.p[k - i];
    acc += (int64_t)acc2; acc2 = 0;
    r[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r[NL - 1] = (uint32_t)acc;
}
// a: signed limbs (a limb-wise difference), b: limbs masked to 28 bits -> provably non-negative
template <bool OPAQUE>
__global__ void k(const uint32_t* x, const uint32_t* y, uint32_t* out) {
  uint32_t a[NL], b[NL], r[NL];
  const size_t t = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 64;
  for (int i = 0; i < NL; ++i) { a[i] = y[t + i] & LMASK; b[i] = x[t + i] - y[t + 32 + i]; }
  mul_s<OPAQUE>(r, a, b);
  for (int i = 0; i < NL; ++i) out[t + i] = r[i];
}
template __global__ void k<false>(const uint32_t*, const uint32_t*, uint32_t*);
template __global__ void k<true>(const uint32_t*, const uint32_t*, uint32_t*);
