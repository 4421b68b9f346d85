// Development experiment (round 4, verdict item 8): can the matrix cores take the REDUCTION half of a 753-bit modular product?
//
// The north star says "no MFMA: this is not a dense contraction", and for the product a * b of two variables that is true: every lane
// has its own operands, there is no shared matrix.  The reduction is different.  Write T = a b = T_lo + 2^756 T_hi.  Then
//     T  ==  T_lo + sum_k t_k * (2^(756 + 8k) mod p)      (mod p),      t_k = byte k of T_hi,  k < 95
// and with c_r = sum_k M[r][k] t_k, M[r][k] = byte r of (2^(756 + 8k) mod p), the sum is  sum_r c_r 2^(8r).  M is a CONSTANT 95 x 95
// byte matrix, shared by every product in flight: for the 64 products of a wave,  C[96 x 64] = M[96 x 96] * t[96 x 64]  is a dense
// int8 contraction with int32 accumulators (c_r <= 95 * 255^2 < 2^23) -- 18 v_mfma_i32_32x32x32_i8 per wave (3 row tiles x 3 depth
// tiles x 2 column tiles, 32 cycles each = 576 cycles) against the 729 v_mad_u64_u32 (2916 cycles) of the Montgomery reduction half.
// Values are then kept in PLAIN form (no Montgomery radix): a representation change of the whole field layer, not a local patch --
// hence an experiment: what does one product cost this way, everything the VALU still has to do included?
//
//   VALU: 729 multiply-adds of the product half + 2 x 54 to normalise its columns to 28-bit limbs (T_hi must be exact bytes)
//         + 48 to pack T_hi into 24 dwords + 12 v_permlane32_swap to lay the 64 vectors out as two B operands
//         + 48 v_permlane32_swap to bring every lane the 96 column sums of ITS product back + 96 multiply-adds to place them
//         + ~84 to normalise + ~90 for the final quotient step (q from the top limbs, r = R - q p, r in [0, 2p))
//       ~ 1200 instructions instead of 1643.
//   MFMA operand layouts: A (the constant matrix) and B (the byte vectors) of one instruction have the same K mapping, so any fixed
//   rule slot (lane half h, register v, byte b) -> k = 16 h + 4 v + b used for BOTH is correct whatever the hardware's own numbering
//   is; rows / columns are lane % 32; the 32 x 32 result layout is row = 8 (v / 4) + 4 (lane / 32) + v % 4, column = lane % 32.
//
// The kernel checks every product against a host big-integer a b mod p, then times dependent chains like mul_variants.hip.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I../../snark-challenge-prover-reference_amd/csrc mul_mfma.hip -o /tmp/mul_mfma && /tmp/mul_mfma
// (the option lets the 96 accumulators of the matrix instructions live in ordinary vector registers: read out of the accumulation registers
// they cost 96 v_accvgpr_read per product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fp753.hip.h"
using namespace mnt753;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// the constant operand: [row tile 3][depth tile 3][lane 64] x 16 bytes, in the slot order described above
__device__ uint4 g_matrix[3 * 3 * 64];
// 2^756 mod p as 27 limbs is not needed; the quotient step needs p only (FPC)

// C0 = 128 * sum_k (2^(756 + 8k) mod p) mod p, as 27 limbs: the constant the signed digits leave behind (see below)
__device__ uint32_t g_c0[NL];

template <int M>
__device__ __forceinline__ void mul_mfma(Fp<M>& r, const Fp<M>& a, const Fp<M>& b, const uint4* __restrict__ mat, const uint32_t* __restrict__ c0v) {
  // ---- product half: 53 columns, normalised on the fly to 28-bit limbs t[0 .. 53]
  uint32_t t[2 * NL];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
    const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
#pragma unroll
    for (int i = lo; i <= hi; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
    t[k] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  t[2 * NL - 1] = (uint32_t)acc;
  // ---- T_hi = limbs 27 .. 53 as 24 dwords (756 bits).  The matrix cores multiply SIGNED int8: byte b with bit 7 flipped, read as int8,
  //      is b - 128, exactly and without carries, and  sum_k b_k v_k = sum_k (b_k - 128) v_k + 128 sum_k v_k  -- the second term is the
  //      constant C0 (mod p), added to the result below.  The matrix itself is stored in signed byte digits (host).
  uint32_t D[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    const int i = (32 * j) / LB, s = 32 * j - LB * i;          // dword j starts inside limb i at bit s (s = 4 j mod 28)
    const uint32_t lo = t[NL + i] >> s;
    const uint32_t hi = (i + 1 < NL) ? t[NL + i + 1] << (LB - s) : 0u;
    D[j] = (lo | hi) ^ 0x80808080u;
  }
  // ---- B operands: depth tile kt takes dwords 8 kt .. 8 kt + 7; lanes 0-31 hold dwords 0..3 of a column, lanes 32-63 dwords 4..7.
  //      One v_permlane32_swap per dword pair yields the operand of BOTH column tiles (products of lanes 0-31 / 32-63).
  v16i c0[3], c1[3];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) { c0[mt] = (v16i)(0); c1[mt] = (v16i)(0); }
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    v4i b0, b1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // after the swap: first = [x.lanes 0-31 | y.lanes 0-31 moved up], second = [x.lanes 32-63 moved down | y.lanes 32-63]
      auto sw = __builtin_amdgcn_permlane32_swap(D[8 * kt + q], D[8 * kt + 4 + q], false, false);
      b0[q] = (int)sw[0];
      b1[q] = (int)sw[1];
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const uint4 av = mat[(mt * 3 + kt) * 64 + (threadIdx.x & 63u)];
      const v4i am = {(int)av.x, (int)av.y, (int)av.z, (int)av.w};
      c0[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(am, b0, c0[mt], 0, 0, 0);
      c1[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(am, b1, c1[mt], 0, 0, 0);
    }
  }
  // ---- every lane gets the 96 column sums of its own product (signed, |c_r| < 2^21) and places them:
  //      R = T_lo + C0 + sum_r c_r 2^(8r), one v_mad_i64_i32 per column sum (the powers of two sit in scalar registers behind an
  //      opaque move, so that the compiler does not turn the product into a sign extension, a 64-bit shift and a 64-bit add)
  int32_t P2[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) { P2[j] = 1 << (4 * j); asm volatile("" : "+s"(P2[j])); }
  int64_t R[NL + 2];
#pragma unroll
  for (int i = 0; i < NL; ++i) R[i] = (int64_t)(t[i] + c0v[i]);      // both below 2^28
  R[NL] = 0; R[NL + 1] = 0;
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      auto sw = __builtin_amdgcn_permlane32_swap((unsigned)c0[mt][v], (unsigned)c1[mt][v], false, false);
      const int row_lo = 32 * mt + 8 * (v / 4) + (v % 4), row_hi = row_lo + 4;   // rows with row % 8 < 4 come from c0', the others from c1'
      {
        const int i = (8 * row_lo) / LB, s = 8 * row_lo - LB * i;
        R[i] += (int64_t)(int32_t)sw[0] * (int64_t)P2[s / 4];
      }
      {
        const int i = (8 * row_hi) / LB, s = 8 * row_hi - LB * i;
        R[i] += (int64_t)(int32_t)sw[1] * (int64_t)P2[s / 4];
      }
    }
  }
  // ---- quotient from the top of the un-normalised sums (|R_i| < 2^52; everything below R_24 moves q by less than 2^-30), then
  //      r = R - q p folded into the accumulators and ONE carry pass: r in [0, 2p), limbs 0..25 in [0, 2^28)
  const float vtop = (float)R[NL + 1] * 72057594037927936.0f + (float)R[NL] * 268435456.0f + (float)R[NL - 1] + (float)R[NL - 2] * (1.0f / 268435456.0f) +
                     (float)R[NL - 3] * (1.0f / 72057594037927936.0f);
  int32_t nq = -(int32_t)floorf(vtop * (1.0f / (float)FPC[M].p[NL - 1]) - 0.5f);
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int64_t x = R[i] + (int64_t)nq * (int64_t)(int32_t)FPC[M].p[i] + c;
    if (i == NL - 1) { x += (R[NL] + R[NL + 1] * ((int64_t)1 << LB)) * ((int64_t)1 << LB); r.l[i] = (uint32_t)x; }
    else { r.l[i] = (uint32_t)x & LMASK; c = x >> LB; }
  }
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k_mul(uint32_t* p, int reps) {
  extern __shared__ uint4 lds[];
  Fp<1> a, b, c;
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 64;
  for (int i = 0; i < NL; ++i) { a.l[i] = p[base + i] & LMASK; b.l[i] = p[base + 32 + i] & LMASK; }
  a.l[NL - 1] &= 0x1ffffffu; b.l[NL - 1] &= 0x1ffffffu;     // below 2^753: about [0, 2p)
#pragma nounroll
  for (int r = 0; r < reps; ++r) {
    if (VARIANT == 0) { fp_mul(c, a, b); fp_mul(a, c, b); }
    if (VARIANT == 1) { mul_mfma(c, a, b, g_matrix, g_c0); mul_mfma(a, c, b, g_matrix, g_c0); }
  }
  if (threadIdx.x == 9999) lds[0] = make_uint4(a.l[0], 0, 0, 0);
  for (int i = 0; i < NL; ++i) p[base + i] = a.l[i] ^ b.l[i];
}

// one product per lane, operands and result to memory: the correctness check
__global__ void __launch_bounds__(256) k_check(const uint32_t* in, uint32_t* out) {
  Fp<1> a, b, c;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < NL; ++i) { a.l[i] = in[t * 64 + i]; b.l[i] = in[t * 64 + 32 + i]; }
  mul_mfma(c, a, b, g_matrix, g_c0);
  for (int i = 0; i < NL; ++i) out[t * 32 + i] = c.l[i];
}

// ---- host big integers (64-bit limbs, little endian) -----------------------------------------------------------------------
typedef std::vector<uint64_t> Big;
static int cmp(const Big& a, const Big& b) {
  for (size_t i = std::max(a.size(), b.size()); i-- > 0;) {
    const uint64_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
    if (x != y) return x < y ? -1 : 1;
  }
  return 0;
}
static void sub_in_place(Big& a, const Big& b) {
  unsigned __int128 br = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    const unsigned __int128 x = (unsigned __int128)a[i] - (i < b.size() ? b[i] : 0) - br;
    a[i] = (uint64_t)x; br = (x >> 64) & 1;
  }
}
static void shl1(Big& a) { uint64_t c = 0; for (auto& w : a) { const uint64_t n = w >> 63; w = (w << 1) | c; c = n; } }
static Big mul(const Big& a, const Big& b) {
  Big r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); ++i) {
    unsigned __int128 c = 0;
    for (size_t j = 0; j < b.size(); ++j) { c += (unsigned __int128)a[i] * b[j] + r[i + j]; r[i + j] = (uint64_t)c; c >>= 64; }
    r[i + b.size()] = (uint64_t)c;
  }
  return r;
}
static Big mod(const Big& a, const Big& p) {   // shift-subtract
  Big r(p.size() + 1, 0);
  for (size_t bit = a.size() * 64; bit-- > 0;) {
    shl1(r);
    r[0] |= (a[bit / 64] >> (bit % 64)) & 1;
    if (cmp(r, p) >= 0) sub_in_place(r, p);
  }
  r.resize(p.size());
  return r;
}
static Big from_limbs28(const uint32_t* l, int n) {
  Big r((n * 28 + 63) / 64 + 1, 0);
  for (int i = 0; i < n; ++i) {
    const int bit = 28 * i;
    r[bit / 64] |= (uint64_t)l[i] << (bit % 64);
    if (bit % 64 > 36) r[bit / 64 + 1] |= (uint64_t)l[i] >> (64 - bit % 64);
  }
  return r;
}

template <int V>
static double run(const char* name, uint32_t* d, int waves_per_simd) {
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
  printf("%-58s waves/SIMD %d  %8.3f ms  %7.2f G products/s  (%s)\n", name, waves_per_simd, best, ops / best / 1e6, hipGetErrorString(hipGetLastError()));
  return ops / best / 1e6;
}

int main() {
  constexpr int M = 1;
  // the modulus and the constant matrix
  Big p(FPC[M].p64, FPC[M].p64 + 12);
  std::vector<uint8_t> mat(96 * 96, 0);     // mat[r * 96 + k]
  {
    Big v(13, 0); v[0] = 1;
    for (int i = 0; i < 756; ++i) { shl1(v); if (cmp(v, p) >= 0) sub_in_place(v, p); }
    Big sum(13, 0);
    for (int k = 0; k < 96; ++k) {
      // signed byte digits of v = 2^(756 + 8k) mod p: bytes of v + 0x8080..80 (96 bytes) with bit 7 flipped (the carries of the
      // addition make the digits of v, not of v + bias: sum_r (e_r - 128) 2^(8r) = v)
      Big e = v;
      unsigned __int128 cy = 0;
      for (size_t i = 0; i < 12; ++i) { cy += (unsigned __int128)e[i] + 0x8080808080808080ull; e[i] = (uint64_t)cy; cy >>= 64; }
      for (int r = 0; r < 96; ++r) mat[r * 96 + k] = (uint8_t)((e[r / 8] >> (8 * (r % 8))) ^ 0x80);
      // C0 accumulates 128 * v
      Big v128 = v;
      for (int i = 0; i < 7; ++i) { shl1(v128); if (cmp(v128, p) >= 0) sub_in_place(v128, p); }
      unsigned __int128 c2 = 0;
      for (size_t i = 0; i < 13; ++i) { c2 += (unsigned __int128)sum[i] + v128[i]; sum[i] = (uint64_t)c2; c2 >>= 64; }
      if (cmp(sum, p) >= 0) sub_in_place(sum, p);
      for (int i = 0; i < 8; ++i) { shl1(v); if (cmp(v, p) >= 0) sub_in_place(v, p); }
    }
    uint32_t c0l[NL];
    for (int i = 0; i < NL; ++i) { const int bit = 28 * i; uint64_t w = sum[bit / 64] >> (bit % 64); if (bit % 64 > 36) w |= sum[bit / 64 + 1] << (64 - bit % 64); c0l[i] = (uint32_t)w & LMASK; }
    hipMemcpyToSymbol(HIP_SYMBOL(g_c0), c0l, sizeof(c0l));
  }
  std::vector<uint32_t> hm(3 * 3 * 64 * 4);
  for (int mt = 0; mt < 3; ++mt)
    for (int kt = 0; kt < 3; ++kt)
      for (int lane = 0; lane < 64; ++lane)
        for (int v = 0; v < 4; ++v) {
          uint32_t w = 0;
          for (int b = 0; b < 4; ++b) {
            const int row = 32 * mt + lane % 32, k = 32 * kt + 16 * (lane / 32) + 4 * v + b;
            w |= (uint32_t)mat[row * 96 + k] << (8 * b);
          }
          hm[((mt * 3 + kt) * 64 + lane) * 4 + v] = w;
        }
  hipMemcpyToSymbol(HIP_SYMBOL(g_matrix), hm.data(), hm.size() * 4);

  // ---- correctness: 4096 products against the host
  const size_t nchk = 4096;
  std::vector<uint32_t> hin(nchk * 64), hout(nchk * 32);
  uint64_t s = 0x9e3779b97f4a7c15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (size_t t = 0; t < nchk; ++t)
    for (int o = 0; o < 64; o += 32) {
      for (int i = 0; i < NL; ++i) hin[t * 64 + o + i] = (uint32_t)rnd() & LMASK;
      hin[t * 64 + o + NL - 1] &= 0x1ffffffu;
      if (t == 0) for (int i = 0; i < NL; ++i) hin[o + i] = 0;                               // 0 * 0
      if (t == 1) for (int i = 0; i < NL; ++i) hin[64 + o + i] = i == 0 ? 1u : 0u;             // 1 * 1
      if (t == 2) for (int i = 0; i < NL; ++i) hin[128 + o + i] = i < NL - 1 ? LMASK : 0x1ffffffu;   // the largest operands
    }
  uint32_t *din, *dout;
  hipMalloc(&din, hin.size() * 4); hipMalloc(&dout, hout.size() * 4);
  hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(nchk / 256), dim3(256), 0, 0, din, dout);
  if (hipDeviceSynchronize() != hipSuccess) { printf("check kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost);
  Big p2 = p; shl1(p2);
  size_t bad = 0, above = 0;
  for (size_t t = 0; t < nchk; ++t) {
    const Big a = from_limbs28(&hin[t * 64], NL), b = from_limbs28(&hin[t * 64 + 32], NL);
    const Big want = mod(mul(a, b), p);
    Big got = from_limbs28(&hout[t * 32], NL);
    bool limbs_ok = true;
    for (int i = 0; i < NL; ++i) if (hout[t * 32 + i] > LMASK) limbs_ok = false;
    if (cmp(got, p2) >= 0) ++above;
    const Big gm = mod(got, p);
    if (!limbs_ok || cmp(gm, want) != 0) { if (bad < 4) printf("MISMATCH at product %zu (limbs %s)\n", t, limbs_ok ? "in range" : "out of range"); ++bad; }
  }
  printf("check: %zu products, %zu wrong, %zu results not below 2p\n", nchk, bad, above);

  // ---- throughput
  uint32_t* d; const size_t n = (size_t)256 * 8 * 256 * 64;
  hipMalloc(&d, n * 4);
  std::vector<uint32_t> h(n); for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u) >> 4;
  for (int w : {1, 2, 4}) {
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const double g0 = run<0>("Montgomery product scanning, 1458 v_mad_u64_u32 (fp_mul)", d, w);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const double g1 = run<1>("product half on the VALU + reduction by 18 v_mfma_i32_32x32x32_i8", d, w);
    printf("    ratio %.3f\n", g1 / g0);
  }
  return bad ? 1 : 0;
}
