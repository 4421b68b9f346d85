// Development harness: the XYZZ / projective VMs executed on the GPU, lane-divergent on purpose, vs host_field.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../snark-challenge-prover-reference_amd/csrc/msm_kernels.hip.h"
#include "../snark-challenge-prover-reference_amd/csrc/host_field.hpp"
using namespace mnt753;

// lane l adds points pts[0..len_l) in order (len_l = 1 + l % maxlen); optionally flushes (TOPROJ) at the end
template <class C>
__global__ void __launch_bounds__(256, 1) k_chain(const uint32_t* pts, uint32_t* out, int maxlen) {
  using F = typename C::F;
  constexpr int EW = F::DEG * FPS_WORDS;
  const int lane = threadIdx.x;
  const int len = 1 + lane % maxlen;
  Xyzz<C> acc; Aff<C> Q;
  F::zero(acc.X); F::zero(acc.Y); F::zero(acc.ZZ); F::zero(acc.ZZZ); F::zero(Q.x); F::zero(Q.y);
  bool acc_zero = true, done = false;
  int e = 0;
#pragma nounroll
  while (!done) {
    int pc = PC_END;
    const bool flush = (e == len);
    if (flush) {
      pc = PCX_TOPROJ;
    } else {
      const uint32_t* src = pts + (size_t)((e + lane) % maxlen) * aff_words<C>();
      e_load<F>(Q.x, src); e_load<F>(Q.y, src + EW);
      if ((e + lane) & 1) F::neg(Q.y, Q.y);
      if (acc_zero) { acc.X = Q.x; acc.Y = Q.y; F::one(acc.ZZ); F::one(acc.ZZZ); acc_zero = false; }
      else pc = PCX_MADD;
    }
    pt_vm_xyzz<C>(acc, Q, pc);
    if (flush) {
      uint32_t* o = out + (size_t)lane * proj_words<C>();
      e_store<F>(o, acc.X); e_store<F>(o + EW, acc.Y); e_store<F>(o + 2 * EW, acc.ZZ);
      done = true;
    } else ++e;
  }
}

template <class C, class HC>
int run(const char* name, const char* golden) {
  using F = typename C::F;
  const int aw = 24 * F::DEG, maxlen = 7;
  FILE* f = fopen(golden, "rb"); if (!f) return 1;
  std::vector<uint64_t> rec(6 * aw + 12);
  std::vector<std::vector<uint64_t>> wire;
  for (int i = 0; i < 8 && (int)wire.size() < maxlen; ++i) {
    if (fread(rec.data(), 8, rec.size(), f) != rec.size()) break;
    if (i == 0 || i >= 5) { wire.emplace_back(rec.begin(), rec.begin() + aw); if ((int)wire.size() < maxlen) wire.emplace_back(rec.begin() + aw, rec.begin() + 2 * aw); }
  }
  fclose(f);
  std::vector<uint32_t> h_pts((size_t)maxlen * aff_words<C>(), 0);
  for (int i = 0; i < maxlen; ++i)
    for (int k = 0; k < 2 * F::DEG; ++k) {
      Fp<F::MOD> v; fp_from_wire(v, (const uint32_t*)(wire[i].data() + 12 * k));
      memcpy(&h_pts[(size_t)i * aff_words<C>() + k * FPS_WORDS], v.l, NL * 4);
    }
  uint32_t *d_pts, *d_out;
  hipMalloc(&d_pts, h_pts.size() * 4); hipMalloc(&d_out, 256 * proj_words<C>() * 4);
  hipMemcpy(d_pts, h_pts.data(), h_pts.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_chain<C>), dim3(1), dim3(256), 0, 0, d_pts, d_out, maxlen);
  std::vector<uint32_t> h_out(256 * proj_words<C>());
  hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 256; ++lane) {
    const int len = 1 + lane % maxlen;
    host::HPoint<HC> ref = host::HPoint<HC>::zero();
    for (int e = 0; e < len; ++e) {
      const uint64_t* w = wire[(e + lane) % maxlen].data();
      host::HPoint<HC> hp;
      for (int k = 0; k < F::DEG; ++k) { hp.X.comp(k) = HC::F::B::from_words(w + 12 * k); hp.Y.comp(k) = HC::F::B::from_words(w + 12 * (F::DEG + k)); }
      hp.Z = HC::F::one();
      if ((e + lane) & 1) hp.Y = -hp.Y;
      ref = ref.add(hp);
    }
    uint64_t proj[108];
    for (int k = 0; k < 3 * F::DEG; ++k) {
      Fp<F::MOD> v; memcpy(v.l, &h_out[(size_t)lane * proj_words<C>() + k * FPS_WORDS], NL * 4);
      fp_to_wire((uint32_t*)(proj + 12 * k), v);
    }
    typename HC::F x, y, rx, ry;
    host::HPoint<HC>::from_wire(proj).to_affine(x, y); ref.to_affine(rx, ry);
    if (!(x == rx && y == ry)) { if (bad < 5) printf("%s lane %d len %d MISMATCH\n", name, lane, len); ++bad; }
  }
  printf("%s: %d / 256 lanes wrong\n", name, bad);
  return bad;
}
// projective VM on the GPU: lane l computes op (l % 3): 0 = A + B (PC_ADD), 1 = A + A (PC_ADD -> doubling), 2 = 2A (PC_DBL)
template <class C>
__global__ void __launch_bounds__(256, 1) k_projops(const uint32_t* pts, uint32_t* out, int npts) {
  const int lane = threadIdx.x;
  Proj<C> A, B;
  proj_load<C>(A, pts + (size_t)(lane % npts) * proj_words<C>());
  proj_load<C>(B, pts + (size_t)((lane / 3 + 1) % npts) * proj_words<C>());
  int pc;
  const int op = lane % 3;
  if (op == 0) pc = PC_ADD; else if (op == 1) { B = A; pc = PC_ADD; } else pc = PC_DBL;
  pt_vm<C, true>(A, B, pc);
  proj_store<C>(out + (size_t)lane * proj_words<C>(), A);
}
template <class C, class HC>
int run_projops(const char* name, const char* golden) {
  using F = typename C::F;
  const int aw = 24 * F::DEG, npts = 7;
  FILE* f = fopen(golden, "rb"); if (!f) return 1;
  std::vector<uint64_t> rec(6 * aw + 12);
  std::vector<std::vector<uint64_t>> wire;
  for (int i = 0; i < 8 && (int)wire.size() < npts; ++i) {
    if (fread(rec.data(), 8, rec.size(), f) != rec.size()) break;
    if (i == 0 || i >= 5) { wire.emplace_back(rec.begin(), rec.begin() + aw); if ((int)wire.size() < npts) wire.emplace_back(rec.begin() + aw, rec.begin() + 2 * aw); }
  }
  fclose(f);
  // projective inputs with Z != 1: (x z, y z, z), z = x of the next point
  std::vector<uint32_t> h_pts((size_t)npts * proj_words<C>(), 0);
  std::vector<host::HPoint<HC>> hp(npts);
  for (int i = 0; i < npts; ++i) {
    Aff<C> p, q;
    for (int k = 0; k < F::DEG; ++k) {
      fp_from_wire(F::comp(p.x, k), (const uint32_t*)(wire[i].data() + 12 * k));
      fp_from_wire(F::comp(p.y, k), (const uint32_t*)(wire[i].data() + 12 * (F::DEG + k)));
      fp_from_wire(F::comp(q.x, k), (const uint32_t*)(wire[(i + 1) % npts].data() + 12 * k));
      hp[i].X.comp(k) = HC::F::B::from_words(wire[i].data() + 12 * k);
      hp[i].Y.comp(k) = HC::F::B::from_words(wire[i].data() + 12 * (F::DEG + k));
    }
    hp[i].Z = HC::F::one();
    Proj<C> P; F::mul(P.X, p.x, q.x); F::mul(P.Y, p.y, q.x); P.Z = q.x;
    for (int k = 0; k < F::DEG; ++k) {
      memcpy(&h_pts[(size_t)i * proj_words<C>() + (0 * F::DEG + k) * FPS_WORDS], F::comp(P.X, k).l, NL * 4);
      memcpy(&h_pts[(size_t)i * proj_words<C>() + (1 * F::DEG + k) * FPS_WORDS], F::comp(P.Y, k).l, NL * 4);
      memcpy(&h_pts[(size_t)i * proj_words<C>() + (2 * F::DEG + k) * FPS_WORDS], F::comp(P.Z, k).l, NL * 4);
    }
  }
  uint32_t *d_pts, *d_out;
  hipMalloc(&d_pts, h_pts.size() * 4); hipMalloc(&d_out, 256 * proj_words<C>() * 4);
  hipMemcpy(d_pts, h_pts.data(), h_pts.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_projops<C>), dim3(1), dim3(256), 0, 0, d_pts, d_out, npts);
  std::vector<uint32_t> h_out(256 * proj_words<C>());
  hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost);
  int bad[3] = {0, 0, 0};
  for (int lane = 0; lane < 256; ++lane) {
    const int op = lane % 3, ia = lane % npts, ib = (lane / 3 + 1) % npts;
    host::HPoint<HC> ref = op == 0 ? hp[ia].add(hp[ib]) : hp[ia].dbl();
    uint64_t proj[108];
    for (int k = 0; k < 3 * F::DEG; ++k) { Fp<F::MOD> v; memcpy(v.l, &h_out[(size_t)lane * proj_words<C>() + k * FPS_WORDS], NL * 4); fp_to_wire((uint32_t*)(proj + 12 * k), v); }
    typename HC::F x, y, rx, ry;
    host::HPoint<HC>::from_wire(proj).to_affine(x, y); ref.to_affine(rx, ry);
    if (!(x == rx && y == ry)) ++bad[op];
  }
  printf("%s projops: wrong ADD %d, ADD(P,P) %d, DBL %d\n", name, bad[0], bad[1], bad[2]);
  return bad[0] + bad[1] + bad[2];
}
// the real accumulate kernel on handcrafted buckets
template <class C, class HC, bool PROJ>
int run_acc(const char* name, const char* golden, uint32_t T, uint32_t REP = 1) {
  using F = typename C::F;
  const int aw = 24 * F::DEG, npts = 7;
  FILE* f = fopen(golden, "rb"); if (!f) return 1;
  std::vector<uint64_t> rec(6 * aw + 12);
  std::vector<std::vector<uint64_t>> wire;
  for (int i = 0; i < 8 && (int)wire.size() < npts; ++i) {
    if (fread(rec.data(), 8, rec.size(), f) != rec.size()) break;
    if (i == 0 || i >= 5) { wire.emplace_back(rec.begin(), rec.begin() + aw); if ((int)wire.size() < npts) wire.emplace_back(rec.begin() + aw, rec.begin() + 2 * aw); }
  }
  fclose(f);
  std::vector<uint32_t> h_pts((size_t)npts * aff_words<C>(), 0);
  for (int i = 0; i < npts; ++i)
    for (int k = 0; k < 2 * F::DEG; ++k) {
      Fp<F::MOD> v; fp_from_wire(v, (const uint32_t*)(wire[i].data() + 12 * k));
      memcpy(&h_pts[(size_t)i * aff_words<C>() + k * FPS_WORDS], v.l, NL * 4);
    }
  const int base_sizes[] = {1, 2, 3, 4, 5, 1, 0, 3, 7, 2, 6, 3, 0, 0, 9, 1};
  const uint32_t nb = 16 * REP;
  std::vector<int> sizes(nb);
  for (uint32_t i = 0; i < nb; ++i) sizes[i] = base_sizes[(i * 7 + i / 16) % 16];
  std::vector<uint32_t> offsets(nb + 1, 0), sorted;
  for (uint32_t b = 0; b < nb; ++b) {
    offsets[b + 1] = offsets[b] + sizes[b];
    for (int k = 0; k < sizes[b]; ++k) { uint32_t idx = (b * 3 + k * 5) % npts; sorted.push_back(idx | (((b + k) % 3 == 0) ? 0x80000000u : 0u)); }
  }
  const uint32_t total = offsets[nb], n_lanes = (total + T - 1) / T;
  const size_t PW = proj_words<C>();
  uint32_t *d_pts, *d_sorted, *d_off, *d_b, *d_e, *d_eb;
  hipMalloc(&d_pts, h_pts.size() * 4); hipMalloc(&d_sorted, sorted.size() * 4); hipMalloc(&d_off, offsets.size() * 4);
  hipMalloc(&d_b, nb * PW * 4); hipMalloc(&d_e, 2 * n_lanes * PW * 4); hipMalloc(&d_eb, 2 * n_lanes * 4);
  hipMemset(d_b, 0xff, nb * PW * 4);
  hipMemcpy(d_pts, h_pts.data(), h_pts.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_sorted, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_off, offsets.data(), offsets.size() * 4, hipMemcpyHostToDevice);
  if constexpr (PROJ) hipLaunchKernelGGL((k_bucket_accumulate_proj<C>), dim3((n_lanes + 255) / 256), dim3(256), 0, 0, d_pts, d_sorted, d_off, nb, d_b, d_e, d_eb, T, n_lanes);
  else hipLaunchKernelGGL((k_bucket_accumulate<C>), dim3((n_lanes + 255) / 256), dim3(256), 0, 0, d_pts, d_sorted, d_off, nb, d_b, d_e, d_eb, T, n_lanes);
  if (hipDeviceSynchronize() != hipSuccess) printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
  std::vector<uint32_t> hb(nb * PW), he(2 * n_lanes * PW), heb(2 * n_lanes);
  hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(he.data(), d_e, he.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(heb.data(), d_eb, heb.size() * 4, hipMemcpyDeviceToHost);
  auto to_host = [&](const uint32_t* p) {
    uint64_t proj[108];
    for (int k = 0; k < 3 * F::DEG; ++k) { Fp<F::MOD> v; memcpy(v.l, p + k * FPS_WORDS, NL * 4); fp_to_wire((uint32_t*)(proj + 12 * k), v); }
    return host::HPoint<HC>::from_wire(proj);
  };
  int bad = 0;
  for (uint32_t b = 0; b < nb; ++b) {
    if (!sizes[b]) continue;
    host::HPoint<HC> ref = host::HPoint<HC>::zero();
    for (uint32_t e = offsets[b]; e < offsets[b + 1]; ++e) {
      const uint64_t* w = wire[sorted[e] & 0x7fffffffu].data();
      host::HPoint<HC> hp;
      for (int k = 0; k < F::DEG; ++k) { hp.X.comp(k) = HC::F::B::from_words(w + 12 * k); hp.Y.comp(k) = HC::F::B::from_words(w + 12 * (F::DEG + k)); }
      hp.Z = HC::F::one();
      if (sorted[e] >> 31) hp.Y = -hp.Y;
      ref = ref.add(hp);
    }
    host::HPoint<HC> got = host::HPoint<HC>::zero();
    int pieces = 0;
    for (uint32_t j = 0; j < 2 * n_lanes; ++j) if (heb[j] == b) { got = got.add(to_host(&he[(size_t)j * PW])); ++pieces; }
    if (!pieces) got = to_host(&hb[(size_t)b * PW]);
    typename HC::F x, y, rx, ry;
    got.to_affine(x, y); ref.to_affine(rx, ry);
    bool ok = x == rx && y == ry;
    if (!ok && bad < 6) { printf("%s T=%u bucket %u (size %d, offsets %u..%u, %d edge pieces) MISMATCH\n", name, T, b, sizes[b], offsets[b], offsets[b + 1], pieces); }
    if (!ok) ++bad;
  }
  printf("%s T=%u REP=%u lanes=%u: %d buckets wrong\n", name, T, REP, n_lanes, bad);
  return bad;
}
int main() {
  int bad = 0;
  bad += run<Mnt4G1, host::HMnt4G1>("mnt4 g1", "tests/golden/group_mnt4_g1.bin");
  bad += run<Mnt4G2, host::HMnt4G2>("mnt4 g2", "tests/golden/group_mnt4_g2.bin");
  bad += run_projops<Mnt4G1, host::HMnt4G1>("mnt4 g1", "tests/golden/group_mnt4_g1.bin");
  bad += run_projops<Mnt4G2, host::HMnt4G2>("mnt4 g2", "tests/golden/group_mnt4_g2.bin");
  for (uint32_t REP : {40u}) for (uint32_t T : {16u, 64u}) {
    bad += run_acc<Mnt4G1, host::HMnt4G1, false>("acc xyzz mnt4 g1", "tests/golden/group_mnt4_g1.bin", T, REP);
    bad += run_acc<Mnt4G1, host::HMnt4G1, true>("acc proj mnt4 g1", "tests/golden/group_mnt4_g1.bin", T, REP);
    bad += run_acc<Mnt4G2, host::HMnt4G2, true>("acc proj mnt4 g2", "tests/golden/group_mnt4_g2.bin", T, REP);
  }
  for (uint32_t T : {7u}) {
    bad += run_acc<Mnt4G1, host::HMnt4G1, false>("acc xyzz mnt4 g1", "tests/golden/group_mnt4_g1.bin", T);
    bad += run_acc<Mnt4G1, host::HMnt4G1, true>("acc proj mnt4 g1", "tests/golden/group_mnt4_g1.bin", T);
    bad += run_acc<Mnt4G2, host::HMnt4G2, true>("acc proj mnt4 g2", "tests/golden/group_mnt4_g2.bin", T);
  }
  return bad != 0;
}
