// Radix-2 NTT over Fr for gfx950.
//
// Replaces libfqfft's basic_radix2_domain on the device (reference:
// depends/libfqfft/libfqfft/evaluation_domain/domains/basic_radix2_domain.tcc:62-134 and
// basic_radix2_domain_aux.tcc:167-202 _basic_serial_radix2_FFT, :321-330 _multiply_by_coset).
// Same transform (bit-reversal, then log2 m decimation-in-time passes); GPU-native schedule:
//
//   * vectors stay in HBM in the wire format (96 B / element, Montgomery R = 2^768).  Twiddles and coset
//     powers are precomputed once per domain in the device form (Montgomery R' = 2^756), and a Montgomery
//     product  mul'(w * R', x * R) = (w x) * R  leaves the data in wire form -- so the transform never
//     converts its data between the two Montgomery radices.
//   * k_ntt_group runs up to 8 consecutive butterfly stages on a 2^ns-element tile held in LDS
//     (27 limbs + 1 pad word = 112 B per element: a 112-byte stride maps 16 consecutive lanes'
//     ds_read_b128 onto 16 distinct bank groups), so a 2^20-point transform makes 3 passes over HBM
//     instead of 20.  The first group gathers its input in bit-reversed order (no separate permutation pass).
//   * the libfqfft scale loops (1/m, g^i, g^-i, 1/Z) are table multiplies fused pairwise.
#pragma once
#include <hip/hip_runtime.h>
#include "fp753.hip.h"
#include "msm_kernels.hip.h"   // storage helpers (fp_load / fp_store / load_wire24 / store_wire24)

namespace mnt753 {

#ifndef MNT753_NTT_LAZY
#define MNT753_NTT_LAZY 1
#endif
constexpr int NTT_MAX_NS = 8;                 // stages per LDS group
constexpr int NTT_BLOCK = 256;                // threads per block = 512 elements per block
constexpr int NTT_LDS_WORDS = 2 * NTT_BLOCK * FPS_WORDS;

template <int M>
__device__ __forceinline__ void lds_load_fp(Fp<M>& r, const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint4 v = q[i];
    r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z;
    if (4 * i + 3 < NL) r.l[4 * i + 3] = v.w;
  }
}
template <int M>
__device__ __forceinline__ void lds_store_fp(uint32_t* p, const Fp<M>& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i)
    q[i] = make_uint4(a.l[4 * i], a.l[4 * i + 1], a.l[4 * i + 2], (4 * i + 3 < NL) ? a.l[4 * i + 3] : 0u);
}

// stages [s0, s0 + ns) of the decimation-in-time transform of size 2^logm.
//   src/dst: wire vectors (may alias unless bitrev is set);  tw: omega^i, i < m/2, device form.
//   in_scale / out_scale (may be null): tables in device form the elements are multiplied by as they are read (index = the element's
//   position in `src`) / written (position in `dst`) -- libfqfft's coset and 1/m loops ride on the transform's own passes over HBM
//   instead of a pass of their own (k_vec_mul_table: 0.10 ms per 2^20 elements, four of them in compute_H).  A product of a canonical
//   element and a table entry below 2p is below 1.22p, inside the range the first carry-free stage assumes.
//   (IN_SCALE / OUT_SCALE are template switches: the plain transform keeps the code and the registers it had without them -- with
//   run-time null checks alone the 2^20 FFT measured 0.80 ms against 0.76.)
template <int M, bool IN_SCALE, bool OUT_SCALE>
__global__ void __launch_bounds__(NTT_BLOCK) k_ntt_group(const uint32_t* src, uint32_t* dst,
                                                        const uint32_t* __restrict__ tw, int logm, int s0, int ns, int bitrev,
                                                        const uint32_t* __restrict__ in_scale, const uint32_t* __restrict__ out_scale) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[NTT_LDS_WORDS];
  const int tpt = 1 << (ns - 1);                       // threads (= butterflies) per tile
  const int tiles_per_block = NTT_BLOCK / tpt;
  const int tile_in_block = threadIdx.x / tpt, bt = threadIdx.x % tpt;
  const size_t n_tiles = (size_t)1 << (logm - ns);
  const size_t tile = (size_t)blockIdx.x * tiles_per_block + tile_in_block;
  const bool active = tile < n_tiles;
  const size_t lo = tile & (((size_t)1 << s0) - 1), hi = tile >> s0;
  const size_t base_idx = (hi << (s0 + ns)) + lo;      // element e of the tile sits at base_idx + (e << s0)
  uint32_t* my = lds + (size_t)tile_in_block * ((size_t)FPS_WORDS << ns);
  if (active) {
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
      const int e = bt + r * tpt;
      size_t idx = base_idx + ((size_t)e << s0);
      if (bitrev) idx = (size_t)(__brevll((unsigned long long)idx) >> (64 - logm));
      uint32_t w[24];
      load_wire24(w, src + idx * 24);
      Fp<M> x;
      fp_unpack(x, w);
      if constexpr (IN_SCALE) {
        Fp<M> k, y;
        fp_load(k, in_scale + idx * FPS_WORDS);
        fp_mul(y, x, k);
        x = y;
      }
      lds_store_fp(my + e * FPS_WORDS, x);
    }
  }
  __syncthreads();
  // the twiddle of a stage is fetched one stage ahead (its address needs nothing the stage computes): the load's latency then
  // hides behind a product instead of standing between the barrier and the product
  auto twiddle_index = [=](int q) -> size_t {
    const int e_lo = ((bt >> q) << (q + 1)) | (bt & ((1 << q) - 1));
    const size_t j = ((size_t)(e_lo & ((1 << q) - 1)) << s0) + lo;
    return j << (logm - 1 - (s0 + q));
  };
  Fp<M> w, w_next;
  fp_zero(w_next);
  if (active && s0 != 0) fp_load(w_next, tw + twiddle_index(0) * FPS_WORDS);
#pragma unroll 1
  for (int q = 0; q < ns; ++q) {
    if (active) {
      const int e_lo = ((bt >> q) << (q + 1)) | (bt & ((1 << q) - 1));
      const int e_hi = e_lo + (1 << q);
      Fp<M> xl, xh, t;
      w = w_next;
      if (q + 1 < ns) fp_load(w_next, tw + twiddle_index(q + 1) * FPS_WORDS);
      lds_load_fp(xl, my + e_lo * FPS_WORDS);
      lds_load_fp(xh, my + e_hi * FPS_WORDS);
#if MNT753_NTT_LAZY
      // Carry-free butterflies (the lazy arithmetic of the pairing levels, fp753.hip.h): x_lo +- w x_hi limb-wise with signed limbs
      // (54 instructions instead of the 540 of fp_add + fp_sub), the product through the signed multiplier, and one normalisation
      // per element every SECOND stage.  Ranges: stage A takes values in [0, 1.51p) with limbs below 2^28 (fresh from fp_unpack or
      // fp_norm) and a twiddle in [0, 2p): t in (-0.34p, 1.34p), outputs in (-0.85p, 2.85p) with |limb| < 2^29; stage B takes those:
      // t in (-0.63p, 1.63p), outputs in (-2.48p, 4.48p) with |limb| < 2^29.6 -- inside what fp_norm accepts (|value| < 5p,
      // |limb| < 2^30), which returns them to [0.49p, 1.51p).  Values mod p are those of the eager form.
      if (s0 + q == 0) {
        t = xh;                                        // the first stage's only twiddle is omega^0 (block-uniform: one product in twenty saved)
      } else {
        fp_mul_s(t, w, xh);
      }
      fp_sub_raw(xh, xl, t);
      fp_addsub_raw(xl, xl, t, false);
      if ((q & 1) || q == ns - 1) { fp_norm(xl, xl); fp_norm(xh, xh); }
#else
      if (s0 + q == 0) {
        t = xh;
      } else {
        fp_mul(t, w, xh);
      }
      fp_sub(xh, xl, t);
      fp_add(xl, xl, t);
#endif
      lds_store_fp(my + e_lo * FPS_WORDS, xl);
      lds_store_fp(my + e_hi * FPS_WORDS, xh);
    }
    __syncthreads();
  }
  if (active) {
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
      const int e = bt + r * tpt;
      const size_t idx = base_idx + ((size_t)e << s0);
      Fp<M> x, c;
      lds_load_fp(x, my + e * FPS_WORDS);
      if constexpr (OUT_SCALE) {
        Fp<M> k, y;
        fp_load(k, out_scale + idx * FPS_WORDS);
        fp_mul(y, x, k);
        x = y;
      }
      fp_canon(c, x);
      uint32_t w[24];
      fp_pack(w, c);
      store_wire24(dst + idx * 24, w);
    }
  }
}

// size-1 ... special case is handled on the host (m == 1 is the identity transform).

// a[i] = a[i] * table[i]      (table in device form; a in wire form)
template <int M>
__global__ void __launch_bounds__(256) k_vec_mul_table(uint32_t* __restrict__ a, const uint32_t* __restrict__ table, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  load_wire24(w, a + i * 24);
  Fp<M> x, t, r, c;
  fp_unpack(x, w);
  fp_load(t, table + i * FPS_WORDS);
  fp_mul(r, x, t);
  fp_canon(c, r);
  fp_pack(w, c);
  store_wire24(a + i * 24, w);
}

// a[i] = a[i] * k              (k: one element in device form, in global memory)
template <int M>
__global__ void __launch_bounds__(256) k_vec_mul_const(uint32_t* __restrict__ a, const uint32_t* __restrict__ k, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  load_wire24(w, a + i * 24);
  Fp<M> x, t, r, c;
  fp_unpack(x, w);
  fp_load(t, k);
  fp_mul(r, x, t);
  fp_canon(c, r);
  fp_pack(w, c);
  store_wire24(a + i * 24, w);
}

// a[i] = a[i] * b[i]           both wire form: mul' gives ab*R*2^12, a second mul' by 2^744 restores ab*R
template <int M>
__global__ void __launch_bounds__(256) k_vec_muleq(uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  Fp<M> x, y, t, r, k, c;
  load_wire24(w, a + i * 24);
  fp_unpack(x, w);
  load_wire24(w, b + i * 24);
  fp_unpack(y, w);
  fp_mul(t, x, y);
  fp_const_limbs(k, FPC[M].k_in);
  fp_mul(r, t, k);
  fp_canon(c, r);
  fp_pack(w, c);
  store_wire24(a + i * 24, w);
}

// dst[i] = src[i] * k          all wire form, k passed by value (one element, 24 words); dst may be src
struct WireElem { uint32_t w[24]; };
template <int M>
__global__ void __launch_bounds__(256) k_vec_scale(uint32_t* __restrict__ dst, const uint32_t* src, WireElem kw, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  Fp<M> x, y, t, r, k, c;
  load_wire24(w, src + i * 24);
  fp_unpack(x, w);
  fp_unpack(y, kw.w);
  fp_mul(t, x, y);
  fp_const_limbs(k, FPC[M].k_in);
  fp_mul(r, t, k);
  fp_canon(c, r);
  fp_pack(w, c);
  store_wire24(dst + i * 24, w);
}

// a[i] = a[i] - b[i]
template <int M>
__global__ void __launch_bounds__(256) k_vec_subeq(uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  Fp<M> x, y, r, c;
  load_wire24(w, a + i * 24);
  fp_unpack(x, w);
  load_wire24(w, b + i * 24);
  fp_unpack(y, w);
  fp_sub(r, x, y);
  fp_canon(c, r);
  fp_pack(w, c);
  store_wire24(a + i * 24, w);
}

// compute_H pointwise step, fused:  a[i] = (a[i]*b[i] - c[i]) / Z      (cuda_prover_piecewise.cu:35-45)
//   k1 = 2^12 * R'  (lifts c to the same 2^12-shifted radix as mul'(a,b)),  k2 = Z^-1 * R' * 2^-12
template <int M>
__global__ void __launch_bounds__(256) k_h_pointwise(uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                    const uint32_t* __restrict__ cvec, const uint32_t* __restrict__ k1,
                                                    const uint32_t* __restrict__ k2, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  Fp<M> x, y, z, ab, cz, d, kk, r, c;
  load_wire24(w, a + i * 24);
  fp_unpack(x, w);
  load_wire24(w, b + i * 24);
  fp_unpack(y, w);
  load_wire24(w, cvec + i * 24);
  fp_unpack(z, w);
  fp_mul(ab, x, y);
  fp_load(kk, k1);
  fp_mul(cz, z, kk);
  fp_sub(d, ab, cz);
  fp_load(kk, k2);
  fp_mul(r, d, kk);
  fp_canon(c, r);
  fp_pack(w, c);
  store_wire24(a + i * 24, w);
}

// out[i] = scale * base^i in device form.  pow2[k] = base^(2^k) and scale arrive in wire form.
template <int M>
__global__ void __launch_bounds__(256) k_pow_table(uint32_t* __restrict__ out, const uint32_t* __restrict__ pow2_wire,
                                                  const uint32_t* __restrict__ scale_wire, size_t n, int nbits) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  Fp<M> acc, f, t;
  load_wire24(w, scale_wire);
  fp_from_wire(acc, w);
#pragma unroll 1
  for (int k = 0; k < nbits; ++k) {
    if ((i >> k) & 1) {
      load_wire24(w, pow2_wire + 24 * k);
      fp_from_wire(f, w);
      fp_mul(t, acc, f);
      acc = t;
    }
  }
  fp_store(out + i * FPS_WORDS, acc);
}

// wire element -> device form (single constants)
template <int M>
__global__ void k_consts_to_internal(uint32_t* __restrict__ out, const uint32_t* __restrict__ wire, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  load_wire24(w, wire + 24 * i);
  Fp<M> v;
  fp_from_wire(v, w);
  fp_store(out + i * FPS_WORDS, v);
}

static __global__ void __launch_bounds__(256) k_copy_h(uint32_t* __restrict__ h, const uint32_t* __restrict__ a, size_t m) {
  // h[0..m) = a[0..m), h[m] = 0      (vector_Fr_zeros(m+1) + vector_Fr_copy_into, cuda_prover_piecewise.cu:50-51)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 16-byte quad
  size_t quads = m * 6;
  if (i < quads) reinterpret_cast<uint4*>(h)[i] = reinterpret_cast<const uint4*>(a)[i];
  else if (i < quads + 6) reinterpret_cast<uint4*>(h)[i] = make_uint4(0, 0, 0, 0);
}

}  // namespace mnt753
