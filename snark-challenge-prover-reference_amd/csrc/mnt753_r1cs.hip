// Witness-map front end: the evaluations of the constraint system on the assignment that
// compute_H starts from.  The reference computes them on the CPU before the prover runs -- libsnark/generate_parameters.cpp:44-57
// writes them into the input file as ca / cb / cc; they are the first loop of r1cs_to_qap_witness_map
// (libsnark/reductions/r1cs_to_qap/r1cs_to_qap.tcc:223-237):
//     ca[i] = <a_i, (1, w)>,  cb[i] = <b_i, (1, w)>,  cc[i] = <c_i, (1, w)>         for the nc constraints
//     ca[nc + i] = (1, w)[i]                                                        for i = 0 .. num_inputs  (input consistency rows)
// and zero above.  Here the constraint system stays on the device as three CSR matrices; the assignment is the vector w of the input
// file (w[0] = 1), so a term with variable index k multiplies w[k].  Built for circuit scale (round 3):
//   * one thread per (matrix, row), rows taken in order of DECREASING length (a work list built once at create time): the 64 rows
//     of a wave have the same number of terms to within one, so ragged systems do not idle lanes behind their longest row;
//   * ONE product per term and none per row: coefficients are stored in the device radix (c R'), the assignment is used as it
//     lies in the file (w R, unpacked to 28-bit limbs with shifts), and mul'(c R', w R) = c w R is already the wire form of the
//     term -- no conversion of w, none of the result (the trick of the NTT twiddles);
//   * terms are summed limb-wise without carries, one normalisation (fp_norm) per two terms, one canonicalisation per row.
// HBM-bound by design: per term 96 B of w gathered + 112 B coefficient + 4 B index, per row 96 B written; bench.py (extras) times a
// 2^20-row system and reports the achieved GB/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <new>
#include <vector>

#include "common_host.hpp"
#include "msm_kernels.hip.h"

using namespace mnt753;

struct mnt753_r1cs {
  int curve = 0, frm = 0;
  uint64_t num_inputs = 0, m = 0, nc = 0;
  uint64_t* row_ptr[3] = {nullptr, nullptr, nullptr};   // device, nc + 1 each
  uint32_t* col[3] = {nullptr, nullptr, nullptr};       // device
  uint32_t* coeff[3] = {nullptr, nullptr, nullptr};     // device radix, FPS_WORDS per term
  uint64_t nnz[3] = {0, 0, 0};
  uint64_t* work = nullptr;                             // device, 3 nc items (which * nc + row), longest rows first
};

namespace {
template <int M>
__global__ void __launch_bounds__(256) k_coeff_to_internal(const uint32_t* __restrict__ wire, uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  load_wire24(w, wire + 24 * i);
  Fp<M> v;
  fp_from_wire(v, w);
  fp_store(out + i * FPS_WORDS, v);
}
// work[t] = which * nc + row for the t-th longest row of the three matrices (rows < nc); threads beyond the list fill the tail of the
// three outputs: rows [nc, nc + num_inputs] of ca copy w, everything else above nc is zero
template <int M>
__global__ void __launch_bounds__(256) k_r1cs_evaluate(const uint64_t* __restrict__ rp_a, const uint32_t* __restrict__ col_a, const uint32_t* __restrict__ cf_a,
                                                      const uint64_t* __restrict__ rp_b, const uint32_t* __restrict__ col_b, const uint32_t* __restrict__ cf_b,
                                                      const uint64_t* __restrict__ rp_c, const uint32_t* __restrict__ col_c, const uint32_t* __restrict__ cf_c,
                                                      const uint64_t* __restrict__ work, const uint32_t* __restrict__ w_wire, uint32_t* __restrict__ ca,
                                                      uint32_t* __restrict__ cb, uint32_t* __restrict__ cc, uint64_t nc, uint64_t num_inputs, uint64_t out_len) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t wv[24];
  if (t >= 3 * nc) {                       // tail rows: t - 3 nc enumerates (which, row - nc)
    const uint64_t u = t - 3 * nc, tail = out_len - nc;
    if (u >= 3 * tail) return;
    const int which = (int)(u / tail);
    const uint64_t row = nc + (u - (uint64_t)which * tail);
    uint32_t* dst = (which == 0 ? ca : (which == 1 ? cb : cc)) + 24 * row;
    if (which == 0 && row <= nc + num_inputs) {
      load_wire24(wv, w_wire + 24 * (row - nc));
    } else {
#pragma unroll
      for (int j = 0; j < 24; ++j) wv[j] = 0;
    }
    store_wire24(dst, wv);
    return;
  }
  const uint64_t item = work[t];
  const int which = (int)(item / nc);
  const uint64_t row = item - (uint64_t)which * nc;
  uint32_t* dst = (which == 0 ? ca : (which == 1 ? cb : cc)) + 24 * row;
  const uint64_t* rp = which == 0 ? rp_a : (which == 1 ? rp_b : rp_c);
  const uint32_t* col = which == 0 ? col_a : (which == 1 ? col_b : col_c);
  const uint32_t* cf = which == 0 ? cf_a : (which == 1 ? cf_b : cf_c);
  Fp<M> acc, c, x, p;
  fp_zero(acc);
  uint32_t pending = 0;
  for (uint64_t k = rp[row]; k < rp[row + 1]; ++k) {
    load_wire24(wv, w_wire + 24 * (size_t)col[k]);
    fp_unpack(x, wv);                      // w R as an integer < r, 28-bit limbs
    fp_load(c, cf + k * FPS_WORDS);        // c R'
    fp_mul(p, c, x);                       // c w R, lazily in [0, 2r)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc.l[i] += p.l[i];
    if (++pending == 2u) { fp_norm(acc, acc); pending = 0; }   // value < 1.51 r + 2 * 2r: inside fp_norm's range
  }
  if (pending) fp_norm(acc, acc);
  Fp<M> canon;
  fp_canon(canon, acc);
  fp_pack(wv, canon);
  store_wire24(dst, wv);
}
}  // namespace

extern "C" {

int mnt753_r1cs_create(int curve, uint64_t num_inputs, uint64_t m, uint64_t nc, const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                       const uint64_t* const coeff[3], mnt753_r1cs** out) {
  if (curve < 0 || curve > 1 || !out || !row_ptr || !col || !coeff) return set_error(MNT753_EINVAL, "r1cs_create: bad argument");
  if (int rc = require_device()) return rc;
  if (num_inputs > m) return set_error(MNT753_EINVAL, "r1cs_create: more inputs than variables");
  for (int k = 0; k < 3; ++k) {
    if (!row_ptr[k] || row_ptr[k][0] != 0) return set_error(MNT753_EINVAL, "r1cs_create: row_ptr must start at 0");
    for (uint64_t i = 0; i < nc; ++i)
      if (row_ptr[k][i + 1] < row_ptr[k][i]) return set_error(MNT753_EINVAL, "r1cs_create: row_ptr not monotone");
    const uint64_t nnz = row_ptr[k][nc];
    if (nnz && (!col[k] || !coeff[k])) return set_error(MNT753_EINVAL, "r1cs_create: null matrix");
    for (uint64_t i = 0; i < nnz; ++i)
      if (col[k][i] > m) return set_error(MNT753_EINVAL, "r1cs_create: variable index out of range");
  }
  mnt753_r1cs* r = new (std::nothrow) mnt753_r1cs();
  if (!r) return set_error(MNT753_ENOMEM, "r1cs_create: host allocation failed");
  r->curve = curve; r->frm = curve == MNT753_CURVE_MNT4753 ? MOD_A : MOD_B;
  r->num_inputs = num_inputs; r->m = m; r->nc = nc;
  for (int k = 0; k < 3; ++k) {
    const uint64_t nnz = row_ptr[k][nc];
    r->nnz[k] = nnz;
    uint32_t* staged = nullptr;
    if (hipMalloc(&r->row_ptr[k], 8 * (nc + 1)) != hipSuccess || hipMalloc(&r->col[k], 4 * (nnz + 1)) != hipSuccess ||
        hipMalloc(&r->coeff[k], 4 * FPS_WORDS * (nnz + 1)) != hipSuccess || hipMalloc(&staged, 96 * (nnz + 1)) != hipSuccess) {
      (void)hipGetLastError();
      mnt753_r1cs_free(r);
      return set_error(MNT753_ENOMEM, "r1cs_create: device allocation failed");
    }
    // any failure below frees the staging buffer and the partly built system before it returns
    hipError_t e = hipMemcpy(r->row_ptr[k], row_ptr[k], 8 * (nc + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) {
      e = hipMemcpy(r->col[k], col[k], 4 * nnz, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(staged, coeff[k], 96 * nnz, hipMemcpyHostToDevice);
      if (e == hipSuccess) {
        const unsigned g = (unsigned)((nnz + 255) / 256);
        if (r->frm == MOD_A) hipLaunchKernelGGL((k_coeff_to_internal<MOD_A>), dim3(g), dim3(256), 0, 0, staged, r->coeff[k], (size_t)nnz);
        else hipLaunchKernelGGL((k_coeff_to_internal<MOD_B>), dim3(g), dim3(256), 0, 0, staged, r->coeff[k], (size_t)nnz);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
      }
    }
    (void)hipFree(staged);
    if (e != hipSuccess) {
      mnt753_r1cs_free(r);
      return set_hip_error(e, "r1cs_create: upload / conversion", __FILE__, __LINE__);
    }
  }
  // work list: the rows of the three matrices by decreasing number of terms (counting sort over the lengths: O(rows))
  {
    std::vector<uint64_t> order((size_t)3 * nc);
    uint64_t longest = 0;
    for (int k = 0; k < 3; ++k) for (uint64_t i = 0; i < nc; ++i) longest = std::max(longest, row_ptr[k][i + 1] - row_ptr[k][i]);
    std::vector<uint64_t> start(longest + 2, 0);
    for (int k = 0; k < 3; ++k) for (uint64_t i = 0; i < nc; ++i) ++start[longest - (row_ptr[k][i + 1] - row_ptr[k][i]) + 1];
    for (uint64_t l = 1; l <= longest + 1; ++l) start[l] += start[l - 1];
    for (int k = 0; k < 3; ++k) for (uint64_t i = 0; i < nc; ++i) order[start[longest - (row_ptr[k][i + 1] - row_ptr[k][i])]++] = (uint64_t)k * nc + i;
    hipError_t e = hipMalloc(&r->work, 8 * ((size_t)3 * nc + 1));
    if (e == hipSuccess && nc) e = hipMemcpy(r->work, order.data(), 8 * (size_t)3 * nc, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      mnt753_r1cs_free(r);
      return set_hip_error(e, "r1cs_create: work list", __FILE__, __LINE__);
    }
  }
  *out = r;
  return 0;
}

int mnt753_r1cs_free(mnt753_r1cs* r) {
  if (!r) return 0;
  if (r->work) (void)hipFree(r->work);
  for (int k = 0; k < 3; ++k) {
    if (r->row_ptr[k]) (void)hipFree(r->row_ptr[k]);
    if (r->col[k]) (void)hipFree(r->col[k]);
    if (r->coeff[k]) (void)hipFree(r->coeff[k]);
  }
  delete r;
  return 0;
}

size_t mnt753_r1cs_domain_size(const mnt753_r1cs* r) { return r ? (size_t)(r->nc + r->num_inputs + 1) : 0; }
size_t mnt753_r1cs_num_variables(const mnt753_r1cs* r) { return r ? (size_t)r->m : 0; }
size_t mnt753_r1cs_num_inputs(const mnt753_r1cs* r) { return r ? (size_t)r->num_inputs : 0; }

int mnt753_r1cs_evaluate(mnt753_r1cs* r, const uint64_t* dev_w, uint64_t* dev_ca, uint64_t* dev_cb, uint64_t* dev_cc, size_t out_len, void* stream) {
  if (!r || !dev_w || !dev_ca || !dev_cb || !dev_cc) return set_error(MNT753_EINVAL, "r1cs_evaluate: null argument");
  if (out_len < r->nc + r->num_inputs + 1) return set_error(MNT753_EINVAL, "r1cs_evaluate: out_len below constraints + inputs + 1");
  if (int rc = require_device()) return rc;
  const unsigned g = (unsigned)((3 * (uint64_t)out_len + 255) / 256);   // 3 nc work items + 3 (out_len - nc) tail rows
  const uint32_t* w = reinterpret_cast<const uint32_t*>(dev_w);
  uint32_t *a = reinterpret_cast<uint32_t*>(dev_ca), *b = reinterpret_cast<uint32_t*>(dev_cb), *c = reinterpret_cast<uint32_t*>(dev_cc);
  if (r->frm == MOD_A)
    hipLaunchKernelGGL((k_r1cs_evaluate<MOD_A>), dim3(g), dim3(256), 0, (hipStream_t)stream, r->row_ptr[0], r->col[0], r->coeff[0], r->row_ptr[1], r->col[1],
                       r->coeff[1], r->row_ptr[2], r->col[2], r->coeff[2], r->work, w, a, b, c, r->nc, r->num_inputs, (uint64_t)out_len);
  else
    hipLaunchKernelGGL((k_r1cs_evaluate<MOD_B>), dim3(g), dim3(256), 0, (hipStream_t)stream, r->row_ptr[0], r->col[0], r->coeff[0], r->row_ptr[1], r->col[1],
                       r->coeff[1], r->row_ptr[2], r->col[2], r->coeff[2], r->work, w, a, b, c, r->nc, r->num_inputs, (uint64_t)out_len);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
