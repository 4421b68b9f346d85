// Test hooks of the C ABI: the device field layer exposed element-wise, so that tests/ can pin fp_mul / fp_add / fp_sub /
// fp_inv / fp_neg / fp_canon / the wire conversions directly against the reference's golden field vectors
// (tests/golden/field_A.bin, field_B.bin: minted from libff's Fp_model, fields/fp.tcc:161-186, 405-417, 491-508, 641-685)
// instead of only through MSM / FFT results.  Not used by the prover.
#include <hip/hip_runtime.h>

#include "common_host.hpp"
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
