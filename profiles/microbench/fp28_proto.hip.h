#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
// prototype: 27 x 28-bit limbs, R' = 2^756, FIPS Montgomery with 64-bit column accumulators
#define NL 27
#define LB 28
#define MASK 0x0fffffffu
struct Fp { uint32_t l[NL]; };
struct FpParams { uint32_t p[NL]; uint32_t inv; };
__constant__ FpParams FQ;

__device__ __forceinline__ void fp_mul(Fp& r, const Fp& a, const Fp& b) {
  uint64_t acc = 0; uint32_t m[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * FQ.p[k - i];
    m[k] = ((uint32_t)acc * FQ.inv) & MASK;
    acc += (uint64_t)m[k] * FQ.p[0];
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)m[i] * FQ.p[k - i];
    r.l[k - NL] = (uint32_t)acc & MASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
}
__constant__ uint32_t FQ_P2[NL];  // 2p
__device__ __forceinline__ void fp_reduce2p(Fp& r, const uint32_t s[NL]) {
  // r = s >= 2p ? s - 2p : s   (s normalized limbs)
  uint32_t d[NL]; int32_t bw = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { int32_t t = (int32_t)s[i] - (int32_t)FQ_P2[i] + bw; d[i] = (uint32_t)t & MASK; bw = t >> LB; }
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = bw < 0 ? s[i] : d[i];
}
__device__ __forceinline__ void fp_add(Fp& r, const Fp& a, const Fp& b) {
  uint32_t s[NL], c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { uint32_t t = a.l[i] + b.l[i] + c; s[i] = t & MASK; c = t >> LB; }
  fp_reduce2p(r, s);
}
__device__ __forceinline__ void fp_sub(Fp& r, const Fp& a, const Fp& b) {
  // a - b + 2p, then reduce
  uint32_t s[NL]; int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)FQ_P2[i] + c; s[i] = (uint32_t)t & MASK; c = t >> LB; }
  fp_reduce2p(r, s);
}
__device__ __forceinline__ bool fp_is_zero(const Fp& a) {  // a in [0,2p): zero iff a==0 or a==p
  uint32_t o0 = 0, o1 = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { o0 |= a.l[i]; o1 |= a.l[i] ^ FQ.p[i]; }
  return o0 == 0 || o1 == 0;
}
