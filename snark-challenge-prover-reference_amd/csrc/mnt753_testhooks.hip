// TEST INFRASTRUCTURE (libmnt753_hip_test.so, include/mnt753_hip_test.h) -- not linked into the product library.
// Test hooks beside the C ABI: the device field layer exposed element-wise, so that tests/ can pin fp_mul / fp_add / fp_sub /
// fp_inv / fp_neg / fp_canon / the wire conversions directly against the reference's golden field vectors
// (tests/golden/field_A.bin, field_B.bin: minted from libff's Fp_model, fields/fp.tcc:161-186, 405-417, 491-508, 641-685)
// instead of only through MSM / FFT results; the extension fields of G2 (one-lane Karatsuba forms and the lane-split forms the
// point kernels run: FieldFp2S / FieldFp3S, curve753.hip.h) against extfield_<curve>.bin (fp2.tcc:79-142, fp3.tcc:83-143); and
// every form of the group law the MSM kernels contain -- the point VM's full / mixed addition and doubling, the straight-line
// mixed and full additions, the two-point-lanes addition -- against groupkat_<curve>_g<k>.bin (mnt4753_g1.cpp:134-346,
// mnt4753_g2.cpp:150-362, mnt6753_g2.cpp:156-368).  Not used by the prover.
#include <hip/hip_runtime.h>

#include "common_host.hpp"
#include "../../include/mnt753_hip_test.h"
#include "msm_kernels.hip.h"

using namespace mnt753;

namespace {
// ops follow oracle_field_op (oracle/mnt753_oracle.h): 0 a*b, 1 a+b, 2 a-b, 3 a^-1, 4 as_bigint(a), 5 -a; 6 a^2 (fp_sqr);
// 7 wire -> device -> wire round trip; 8 (a*b + a*a) through the fused two-product multiplier; 9 13*a (fp_mul_small)
template <int M>
__global__ void __launch_bounds__(64) k_field_op(int op, const uint32_t* __restrict__ a_wire, const uint32_t* __restrict__ b_wire,
                                                uint32_t* __restrict__ out_wire, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[24], wb[24], wo[24];
  load_wire24(wa, a_wire + 24 * i);
  load_wire24(wb, b_wire + 24 * i);
  if (op == 4) {
    fp_wire_to_integer<M>(wo, wa);
    store_wire24(out_wire + 24 * i, wo);
    return;
  }
  Fp<M> a, b, r;
  fp_from_wire(a, wa);
  fp_from_wire(b, wb);
  switch (op) {
    case 0: fp_mul(r, a, b); break;
    case 1: fp_add(r, a, b); break;
    case 2: fp_sub(r, a, b); break;
    case 3: fp_inv(r, a); break;
    case 5: fp_neg(r, a); break;
    case 6: fp_sqr(r, a); break;
    case 8: fp_mul2(r, a, b, a, a); break;
    case 9: fp_mul_small(r, a, 13u); break;
    default: r = a; break;
  }
  fp_to_wire(wo, r);
  store_wire24(out_wire + 24 * i, wo);
}
}  // namespace

extern "C" int mnt753_test_field_op(int mod, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  if (mod < 0 || mod > 1 || op < 0 || op > 9 || (n && (!a || !b || !out))) return set_error(MNT753_EINVAL, "test_field_op: bad argument");
  if (int rc = require_device()) return rc;
  if (n == 0) return 0;
  struct Buf {   // freed on every path out of this function
    uint32_t* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
  } da, db, dout;
  HIP_TRY(hipMalloc(&da.p, 96 * n));
  HIP_TRY(hipMalloc(&db.p, 96 * n));
  HIP_TRY(hipMalloc(&dout.p, 96 * n));
  HIP_TRY(hipMemcpy(da.p, a, 96 * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db.p, b, 96 * n, hipMemcpyHostToDevice));
  const unsigned g = (unsigned)((n + 63) / 64);
  if (mod == MOD_A) hipLaunchKernelGGL((k_field_op<MOD_A>), dim3(g), dim3(64), 0, 0, op, da.p, db.p, dout.p, n);
  else hipLaunchKernelGGL((k_field_op<MOD_B>), dim3(g), dim3(64), 0, 0, op, da.p, db.p, dout.p, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, 96 * n, hipMemcpyDeviceToHost));
  return 0;
}


// ---- extension-field elements, element-wise ------------------------------------------------------------------------------
namespace {
// thread -> (element, component): one-lane fields hold every component in one thread, lane-split fields one component per thread
// (logical_lane / lane_comp of msm_kernels.hip.h).  Threads beyond the list repeat the last element -- the exchanges of the
// lane-split fields need every lane of a group to run -- and do not store.
template <class F>
__device__ __forceinline__ void ext_load(typename F::E& r, const uint32_t* wire) {
  uint32_t w[24];
  if constexpr (F::LANES == 1) {
#pragma unroll 1
    for (int k = 0; k < F::DEG; ++k) { load_wire24(w, wire + 24 * k); fp_from_wire(F::comp(r, k), w); }
  } else {
    load_wire24(w, wire + 24 * lane_comp<F>());
    fp_from_wire(r, w);
  }
}
template <class F>
__device__ __forceinline__ void ext_store(uint32_t* wire, const typename F::E& a) {
  uint32_t w[24];
  if constexpr (F::LANES == 1) {
#pragma unroll 1
    for (int k = 0; k < F::DEG; ++k) { fp_to_wire(w, F::comp(a, k)); store_wire24(wire + 24 * k, w); }
  } else {
    fp_to_wire(w, a);
    store_wire24(wire + 24 * lane_comp<F>(), w);
  }
}

// op: 0 a*b, 1 a*a (through the multiplier the point kernels use), 2 a^-1, 3 a+b, 4 a-b, 5 -a, 6 is_zero(a - b) as 0 / 1 in word 0
template <class F>
__global__ void __launch_bounds__(256, 1) k_ext_op(int op, const uint32_t* __restrict__ a_wire, const uint32_t* __restrict__ b_wire,
                                                  uint32_t* __restrict__ out_wire, uint32_t n) {
  using E = typename F::E;
  const uint32_t t = logical_lane<F>();
  const uint32_t i = t < n ? t : n - 1u;
  constexpr int EWW = 24 * F::DEG;   // wire words of one element
  E a, b, r;
  ext_load<F>(a, a_wire + (size_t)i * EWW);
  ext_load<F>(b, b_wire + (size_t)i * EWW);
  switch (op) {
    case 0: F::mul(r, a, b); break;
    case 1: F::mul(r, a, a); break;
    case 2:
      if constexpr (has_inv<F>::value) F::inv(r, a); else e_inv(r, a, (F*)nullptr);
      break;
    case 3: F::add(r, a, b); break;
    case 4: F::sub(r, a, b); break;
    case 5: F::neg(r, a); break;
    default: {
      E d;
      F::sub(d, a, b);
      const bool z = F::is_zero(d);
      F::zero(r);
      if (z) F::one(r);
    } break;
  }
  if (t < n) ext_store<F>(out_wire + (size_t)i * EWW, r);
}

struct DevBuf {   // freed on every path out of a hook
  uint32_t* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};
template <class F>
int run_ext_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  const size_t bytes = 96 * (size_t)F::DEG * n;
  DevBuf da, db, dout;
  HIP_TRY(hipMalloc(&da.p, bytes)); HIP_TRY(hipMalloc(&db.p, bytes)); HIP_TRY(hipMalloc(&dout.p, bytes));
  HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_ext_op<F>), dim3(blocks_for<F>(n)), dim3(256), 0, 0, op, da.p, db.p, dout.p, (uint32_t)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

// ---- the forms of the group law inside the MSM kernels ----------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void proj_load_wire(Proj<C>& P, const uint32_t* wire) {
  using F = typename C::F;
  ext_load<F>(P.X, wire); ext_load<F>(P.Y, wire + 24 * F::DEG); ext_load<F>(P.Z, wire + 48 * F::DEG);
}
template <class C>
__device__ __forceinline__ void proj_store_wire(uint32_t* wire, const Proj<C>& P) {
  using F = typename C::F;
  ext_store<F>(wire, P.X); ext_store<F>(wire + 24 * F::DEG, P.Y); ext_store<F>(wire + 48 * F::DEG, P.Z);
}
// op: 0  P + Q, both projective: the VM's addition behind add_pc (k_reduce_step, k_edge_level_sum, k_pair_fix)
//     1  2P: the VM's doubling (k_precompute_windows, and what every addition turns into for equal points)
//     2  P + Q, Q affine (its Z is ignored): the VM's mixed addition, identities handled as k_bucket_accumulate does
//     3  the same through pt_madd_line (k_bucket_accumulate of the base fields and the two-lane Fq2)
//     5  P + Q, both projective, through pt_add_line (k_reduce_step_line)
template <class C>
__global__ void __launch_bounds__(256, 1) k_point_op(int op, const uint32_t* __restrict__ p_wire, const uint32_t* __restrict__ q_wire,
                                                    uint32_t* __restrict__ out_wire, uint32_t n) {
  using F = typename C::F;
  const uint32_t t = logical_lane<F>();
  const uint32_t i = t < n ? t : n - 1u;
  constexpr int PWW = 72 * F::DEG;
  Proj<C> P, Q;
  proj_load_wire<C>(P, p_wire + (size_t)i * PWW);
  proj_load_wire<C>(Q, q_wire + (size_t)i * PWW);
  if (op == 0) {
    const int pc = add_pc<C>(P, Q);
    pt_vm<C, true>(P, Q, pc);
  } else if (op == 1) {
    pt_vm<C, true>(P, Q, pt_is_zero(P) ? PC_END : PC_DBL);
  } else if (op == 2 || op == 3) {
    const bool zq = pt_is_zero(Q);            // the callers never feed an identity base (its digits are forced to zero)
    int pc = PC_MADD;
    if (zq) pc = PC_END;
    else if (pt_is_zero(P)) { P.X = Q.X; P.Y = Q.Y; F::one(P.Z); pc = PC_END; }
    if constexpr ((F::LANES == 1 && F::DEG == 1) || F::LANES == 2) {
      if (op == 3) pt_madd_line<C>(P, Q, pc); else pt_vm<C, true>(P, Q, pc);
    } else {
      pt_vm<C, true>(P, Q, pc);
    }
  } else {
    if constexpr (F::LANES == 1 && F::DEG == 1) {
      Proj<C> out;
      pt_add_line<C>(out, P, Q);
      P = out;
    } else {
      const int pc = add_pc<C>(P, Q);
      pt_vm<C, true>(P, Q, pc);
    }
  }
  if (pt_is_zero(P)) pt_set_zero(P);
  if (t < n) proj_store_wire<C>(out_wire + (size_t)i * PWW, P);
}
// op 4: P + Q with two point-lanes per addition (pt_add_pairlanes: k_reduce_step_pair, k_edge_level_sum_pair)
template <class C>
__global__ void __launch_bounds__(256, 1) k_point_op_pairlanes(const uint32_t* __restrict__ p_wire, const uint32_t* __restrict__ q_wire,
                                                              uint32_t* __restrict__ out_wire, uint32_t n) {
  using F = typename C::F;
  const PairGeom<F> g = pair_geometry<F>();
  const uint32_t item = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PairGeom<F>::PAIRS_PER_WAVE + g.pair_in_wave;
  const bool live = g.valid && item < n;
  const uint32_t i = item < n ? item : n - 1u;
  constexpr int PWW = 72 * F::DEG;
  Proj<C> S, T, out;
  // the odd half holds the operands swapped
  proj_load_wire<C>(S, (g.odd ? q_wire : p_wire) + (size_t)i * PWW);
  proj_load_wire<C>(T, (g.odd ? p_wire : q_wire) + (size_t)i * PWW);
  pt_add_pairlanes<C>(out, S, T, g.odd, g.partner4);
  if (g.odd || !live) return;
  if (pt_is_zero(out)) pt_set_zero(out);
  proj_store_wire<C>(out_wire + (size_t)i * PWW, out);
}
// op 6: P + Q with one group of lanes per addition (flow_add, msm_flow.hip.h: k_reduce_step_flow, k_edge_tree_level_flow).  The
// kernels work on points in the device layout: conversion kernels either side.
template <class C>
__global__ void __launch_bounds__(256) k_wire_to_dev(const uint32_t* __restrict__ wire, uint32_t* __restrict__ dev, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Proj<C> P;
  proj_load_wire<C>(P, wire + (size_t)t * 72 * C::F::DEG);
  proj_store<C>(dev + (size_t)t * proj_words<C>(), P);
}
template <class C>
__global__ void __launch_bounds__(256) k_dev_to_wire(const uint32_t* __restrict__ dev, uint32_t* __restrict__ wire, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Proj<C> P;
  proj_load<C>(P, dev + (size_t)t * proj_words<C>());
  if (pt_is_zero(P)) pt_set_zero(P);
  proj_store_wire<C>(wire + (size_t)t * 72 * C::F::DEG, P);
}
template <class C>
int run_point_op_flow(const uint64_t* p, const uint64_t* q, size_t n, uint64_t* out) {
  using F = typename C::F;
  if constexpr (F::LANES != 1) {
    return set_error(MNT753_EINVAL, "test_point_op: the lane-group addition is instantiated with the one-lane configuration (split = 0)");
  } else {
    const size_t bytes = 288 * (size_t)F::DEG * n, dev_bytes = sizeof(uint32_t) * proj_words<C>() * n;
    DevBuf wp, wq, wout, dp, dq, dout;
    HIP_TRY(hipMalloc(&wp.p, bytes)); HIP_TRY(hipMalloc(&wq.p, bytes)); HIP_TRY(hipMalloc(&wout.p, bytes));
    HIP_TRY(hipMalloc(&dp.p, dev_bytes)); HIP_TRY(hipMalloc(&dq.p, dev_bytes)); HIP_TRY(hipMalloc(&dout.p, dev_bytes));
    HIP_TRY(hipMemcpy(wp.p, p, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(wq.p, q, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dout.p, 0, dev_bytes));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL((k_wire_to_dev<C>), dim3(gb), dim3(256), 0, 0, wp.p, dp.p, (uint32_t)n);
    hipLaunchKernelGGL((k_wire_to_dev<C>), dim3(gb), dim3(256), 0, 0, wq.p, dq.p, (uint32_t)n);
    hipLaunchKernelGGL((k_flow_add_list<C>), dim3((unsigned)((n + Flow<C>::PER_WAVE - 1) / Flow<C>::PER_WAVE)), dim3(64), 0, 0, dp.p, dq.p, dout.p, (uint32_t)n);
    hipLaunchKernelGGL((k_dev_to_wire<C>), dim3(gb), dim3(256), 0, 0, dout.p, wout.p, (uint32_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, wout.p, bytes, hipMemcpyDeviceToHost));
    return 0;
  }
}
template <class C>
int run_point_op(int op, const uint64_t* p, const uint64_t* q, size_t n, uint64_t* out) {
  if (op == 6) return run_point_op_flow<C>(p, q, n, out);
  using F = typename C::F;
  const size_t bytes = 288 * (size_t)F::DEG * n;
  DevBuf dp, dq, dout;
  HIP_TRY(hipMalloc(&dp.p, bytes)); HIP_TRY(hipMalloc(&dq.p, bytes)); HIP_TRY(hipMalloc(&dout.p, bytes));
  HIP_TRY(hipMemcpy(dp.p, p, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dq.p, q, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(dout.p, 0, bytes));
  if (op == 4) {
    if constexpr ((F::LANES == 1 && F::DEG == 1) || F::LANES > 1) {
      const unsigned per_block = 4u * PairGeom<F>::PAIRS_PER_WAVE;
      hipLaunchKernelGGL((k_point_op_pairlanes<C>), dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), 0, 0, dp.p, dq.p, dout.p, (uint32_t)n);
    } else {
      return set_error(MNT753_EINVAL, "test_point_op: the two-lanes addition exists for base fields and lane-split fields only");
    }
  } else {
    hipLaunchKernelGGL((k_point_op<C>), dim3(blocks_for<F>(n)), dim3(256), 0, 0, op, dp.p, dq.p, dout.p, (uint32_t)n);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}
}  // namespace

extern "C" int mnt753_test_ext_op(int curve, int split, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  if (curve < 0 || curve > 1 || op < 0 || op > 6 || (n && (!a || !b || !out)) || n > 0x7fffffffu) return set_error(MNT753_EINVAL, "test_ext_op: bad argument");
  if (int rc = require_device()) return rc;
  if (n == 0) return 0;
  if (curve == MNT753_CURVE_MNT4753) return split ? run_ext_op<Mnt4G2S::F>(op, a, b, n, out) : run_ext_op<Mnt4G2::F>(op, a, b, n, out);
  return split ? run_ext_op<Mnt6G2S::F>(op, a, b, n, out) : run_ext_op<Mnt6G2::F>(op, a, b, n, out);
}

extern "C" int mnt753_test_point_op(int curve, int group, int split, int op, const uint64_t* p_proj, const uint64_t* q_proj, size_t n, uint64_t* out_proj) {
  if (curve < 0 || curve > 1 || (group != MNT753_G1 && group != MNT753_G2) || op < 0 || op > 6 || (n && (!p_proj || !q_proj || !out_proj)) || n > 0x7fffffffu)
    return set_error(MNT753_EINVAL, "test_point_op: bad argument");
  if (split && group == MNT753_G1) return set_error(MNT753_EINVAL, "test_point_op: G1 has no lane-split form");
  if (int rc = require_device()) return rc;
  if (n == 0) return 0;
  if (curve == MNT753_CURVE_MNT4753) {
    if (group == MNT753_G1) return run_point_op<Mnt4G1>(op, p_proj, q_proj, n, out_proj);
    return split ? run_point_op<Mnt4G2S>(op, p_proj, q_proj, n, out_proj) : run_point_op<Mnt4G2>(op, p_proj, q_proj, n, out_proj);
  }
  if (group == MNT753_G1) return run_point_op<Mnt6G1>(op, p_proj, q_proj, n, out_proj);
  return split ? run_point_op<Mnt6G2S>(op, p_proj, q_proj, n, out_proj) : run_point_op<Mnt6G2>(op, p_proj, q_proj, n, out_proj);
}
