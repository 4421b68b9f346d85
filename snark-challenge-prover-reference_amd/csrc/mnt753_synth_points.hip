// TEST INFRASTRUCTURE (libmnt753_hip_test.so, include/mnt753_hip_test.h) -- not linked into the product library.
// Deterministic synthetic inputs in the reference's wire format, generated on the host.
//
// Stands in for libsnark/generate_parameters.cpp (the reference's input generator, which needs the
// pairing-based Groth16 generator and is not part of the timed path): bases are on-curve points with
// KNOWN discrete logarithms, base[k] = (s_j + i*d) * G for k = j*CHUNK + i, so the exact value of any
// MSM over them is (sum_k scalar_k * (s_j + i*d) mod r) * G -- a size-independent check that does not
// share a line of code with the Pippenger kernels.  (The uniform scalars are mnt753_synth_scalars of the product library.)
#include <cstring>
#include <thread>
#include <vector>

#include "common_host.hpp"
#include "../../include/mnt753_hip_test.h"
#include "host_field.hpp"

using namespace mnt753;
using namespace mnt753::host;

namespace {
constexpr size_t CHUNK = 1024;

inline uint64_t splitmix64(uint64_t& state) {
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t chunk_scalar(uint64_t seed, uint64_t j) {
  uint64_t st = seed ^ (0xA5A5A5A5A5A5A5A5ull + j * 0x9E3779B97F4A7C15ull);
  uint64_t v = splitmix64(st);
  return v | 1;
}
inline uint64_t step_scalar(uint64_t seed) {
  uint64_t st = seed ^ 0x5EEDF00DCAFEBABEull;
  return splitmix64(st) | 1;
}

// generator points in wire (Montgomery) affine form, produced from decimal literals by tools/gen_constants.py
#include "mnt753_generators.h"

template <class HC>
const uint64_t* generator_words();
template <> const uint64_t* generator_words<HMnt4G1>() { return GEN_MNT4_G1; }
template <> const uint64_t* generator_words<HMnt4G2>() { return GEN_MNT4_G2; }
template <> const uint64_t* generator_words<HMnt6G1>() { return GEN_MNT6_G1; }
template <> const uint64_t* generator_words<HMnt6G2>() { return GEN_MNT6_G2; }

template <class HC>
HPoint<HC> generator() {
  typedef typename HC::F F;
  HPoint<HC> p;
  const uint64_t* w = generator_words<HC>();
  for (int k = 0; k < F::DEG; ++k) {
    p.X.comp(k) = F::B::from_words(w + 12 * k);
    p.Y.comp(k) = F::B::from_words(w + 12 * (F::DEG + k));
  }
  p.Z = F::one();
  return p;
}

template <class HC>
void synth_chunk(uint64_t seed, size_t j, size_t count, uint64_t* out) {
  typedef typename HC::F F;
  typedef HPoint<HC> P;
  const int AW = 24 * F::DEG;
  P G = generator<HC>();
  uint64_t sj = chunk_scalar(seed, j), d = step_scalar(seed);
  P cur = G.mul_words(&sj, 1);
  P D = G.mul_words(&d, 1);
  std::vector<P> pts(count);
  for (size_t i = 0; i < count; ++i) { pts[i] = cur; cur = cur.add(D); }
  // batch inversion of Z (Montgomery's trick)
  std::vector<F> pre(count);
  F acc = F::one();
  for (size_t i = 0; i < count; ++i) { pre[i] = acc; acc = acc * pts[i].Z; }
  F inv = acc.inverse();
  for (size_t i = count; i-- > 0;) {
    F zi = inv * pre[i];
    inv = inv * pts[i].Z;
    F x = pts[i].X * zi, y = pts[i].Y * zi;
    uint64_t* o = out + i * AW;
    for (int k = 0; k < F::DEG; ++k) {
      memcpy(o + 12 * k, x.comp(k).l, 96);
      memcpy(o + 12 * (F::DEG + k), y.comp(k).l, 96);
    }
  }
}

template <class HC>
int synth_points_t(uint64_t seed, size_t n, uint64_t* out, int threads) {
  const int AW = 24 * HC::F::DEG;
  const size_t n_chunks = (n + CHUNK - 1) / CHUNK;
  if (threads < 1) threads = 1;
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([=]() {
      for (size_t j = t; j < n_chunks; j += threads) {
        size_t lo = j * CHUNK, cnt = std::min(CHUNK, n - lo);
        synth_chunk<HC>(seed, j, cnt, out + lo * AW);
      }
    });
  }
  for (auto& th : pool) th.join();
  return 0;
}

template <class HC>
int synth_expected_t(uint64_t seed, size_t n, const uint64_t* scalars, uint64_t* out_proj) {
  typedef HFp<HC::FR> Fr;
  Fr acc = Fr::zero();
  const uint64_t d = step_scalar(seed);
  Fr dM = Fr::from_uint(d);
  for (size_t j = 0; j * CHUNK < n; ++j) {
    Fr e = Fr::from_uint(chunk_scalar(seed, j));
    size_t cnt = std::min(CHUNK, n - j * CHUNK);
    for (size_t i = 0; i < cnt; ++i) {
      acc = acc + Fr::from_words(scalars + 12 * (j * CHUNK + i)) * e;
      e = e + dM;
    }
  }
  uint64_t k[12];
  acc.to_integer(k);
  generator<HC>().mul_words(k, 12).to_wire(out_proj);
  return 0;
}
}  // namespace

#define DISPATCH_CG(fn, ...)                                                                                   \
  (curve == MNT753_CURVE_MNT4753 ? (group == MNT753_G1 ? fn<HMnt4G1>(__VA_ARGS__) : fn<HMnt4G2>(__VA_ARGS__)) \
                                 : (group == MNT753_G1 ? fn<HMnt6G1>(__VA_ARGS__) : fn<HMnt6G2>(__VA_ARGS__)))

extern "C" {
int mnt753_synth_points(int curve, int group, uint64_t seed, size_t n, uint64_t* out_affine, int threads) {
  if (curve < 0 || curve > 1 || (group != MNT753_G1 && group != MNT753_G2) || (n && !out_affine)) return set_error(MNT753_EINVAL, "synth_points: bad argument");
  return DISPATCH_CG(synth_points_t, seed, n, out_affine, threads);
}
int mnt753_synth_expected_msm(int curve, int group, uint64_t seed, size_t n, const uint64_t* scalars, uint64_t* out_projective) {
  if (curve < 0 || curve > 1 || (group != MNT753_G1 && group != MNT753_G2) || !out_projective || (n && !scalars)) return set_error(MNT753_EINVAL, "synth_expected_msm: bad argument");
  return DISPATCH_CG(synth_expected_t, seed, n, scalars, out_projective);
}
}
