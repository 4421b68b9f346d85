// hipcc 7.2 / gfx950, DESIGN.md 4.2 finding 1.  The point VM of round 1 selected its operands with a switch on a lane-divergent
// program counter whose `default:` held a SECOND switch (the projective-addition prologue, pc 32..36).  The G1 instantiation
// returned wrong points on the GPU while the same source compiled for the host was right; flattening the switches fixed it and the
// product only uses flat switches since.  This file is that VM (as of commit 3eeeb30^) on today's field layer, run lane-divergently
// (lanes start at PC_MADD / PC_DBL / PC_ADD by lane index) on the device and on the host.
//   expected: device words == host words for every lane        actual when it reproduces: "N lanes differ"
//   build + run: tools/compiler_repro/check.sh (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../snark-challenge-prover-reference_amd/csrc/curve753.hip.h"
using namespace mnt753;

template <class C, bool WITH_ADD>
HD void old_vm(Proj<C>& P, const Proj<C>& Q, int pc) {
  using F = typename C::F;
  using E = typename F::E;
  E u, v, t3, t4, t5, a, b, r;
  F::zero(u); F::zero(v); F::zero(t3); F::zero(t4); F::zero(t5); F::zero(a); F::zero(b);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
  while (pc != PC_END) {
    switch (pc) {
      case 0: a = P.Z; b = Q.X; break;
      case 1: a = P.Z; b = Q.Y; break;
      case 2: a = u; b = u; break;
      case 3: a = v; b = v; break;
      case 4: a = v; b = t4; break;
      case 5: a = t4; b = P.X; break;
      case 6: a = t3; b = P.Z; break;
      case 7: a = v; b = t3; break;
      case 8: a = u; b = t4; break;
      case 9: a = t5; b = P.Y; break;
      case 10: a = t5; b = P.Z; break;
      case 16: a = P.X; b = P.X; break;
      case 17: a = P.Z; b = P.Z; break;
      case 18: a = P.Y; b = P.Z; break;
      case 19: a = v; b = v; break;
      case 20: a = v; b = t4; break;
      case 21: a = P.Y; b = v; break;
      case 22: a = t4; b = t4; break;
      case 23: F::add(a, P.X, t4); b = a; break;
      case 24: a = u; b = u; break;
      case 25: a = t3; b = v; break;
      case 26: F::sub(a, t4, t3); b = u; break;
      default:
        if (WITH_ADD) {
          switch (pc) {          // <-- the nested switch
            case 32: a = P.X; b = Q.Z; break;
            case 33: a = P.Y; b = Q.Z; break;
            case 34: a = Q.X; b = P.Z; break;
            case 35: a = Q.Y; b = P.Z; break;
            default: a = P.Z; b = Q.Z; break;  // 36
          }
        }
        break;
    }
    F::mul(r, a, b);
    switch (pc) {
      case 0: F::sub(v, r, P.X); pc = 1; break;
      case 1: F::sub(u, r, P.Y); pc = (F::is_zero(u) && F::is_zero(v)) ? PC_DBL : 2; break;
      case 2: t3 = r; pc = 3; break;
      case 3: t4 = r; pc = 4; break;
      case 4: t5 = r; pc = 5; break;
      case 5: t4 = r; pc = 6; break;
      case 6: F::sub(r, r, t5); F::sub(r, r, t4); F::sub(t3, r, t4); F::sub(t4, t4, t3); pc = 7; break;
      case 7: P.X = r; pc = 8; break;
      case 8: t4 = r; pc = 9; break;
      case 9: F::sub(P.Y, t4, r); pc = 10; break;
      case 10: P.Z = r; pc = PC_END; break;
      case 16: t3 = r; pc = 17; break;
      case 17: { E az; C::mul_by_a(az, r); F::add(u, t3, t3); F::add(u, u, t3); F::add(u, u, az); pc = 18; } break;
      case 18: F::add(v, r, r); pc = 19; break;
      case 19: t4 = r; pc = 20; break;
      case 20: P.Z = r; pc = 21; break;
      case 21: t4 = r; pc = 22; break;
      case 22: t5 = r; pc = 23; break;
      case 23: F::sub(r, r, t3); F::sub(t4, r, t5); pc = 24; break;
      case 24: F::sub(r, r, t4); F::sub(t3, r, t4); pc = 25; break;
      case 25: P.X = r; pc = 26; break;
      case 26: F::sub(r, r, t5); F::sub(P.Y, r, t5); pc = PC_END; break;
      default:
        if (WITH_ADD) {
          switch (pc) {          // <-- and its twin on the result side
            case 32: P.X = r; pc = 33; break;
            case 33: P.Y = r; pc = 34; break;
            case 34: F::sub(v, r, P.X); pc = 35; break;
            case 35:
              F::sub(u, r, P.Y);
              if (F::is_zero(u) && F::is_zero(v)) { P.X = Q.X; P.Y = Q.Y; P.Z = Q.Z; pc = PC_DBL; } else pc = 36;
              break;
            default: P.Z = r; pc = 2; break;
          }
        } else {
          pc = PC_END;
        }
        break;
    }
  }
}

using C = Mnt4G1;
constexpr int W = 3 * NL;
HD int start_pc(int lane) { return lane % 3 == 0 ? PC_MADD : (lane % 3 == 1 ? PC_DBL : PC_ADD); }
HD void one_lane(const uint32_t* in, uint32_t* out, int lane) {
  Proj<C> P, Q;
  for (int i = 0; i < NL; ++i) {
    P.X.l[i] = in[i]; P.Y.l[i] = in[NL + i]; P.Z.l[i] = in[2 * NL + i];
    Q.X.l[i] = in[3 * NL + i]; Q.Y.l[i] = in[4 * NL + i]; Q.Z.l[i] = in[5 * NL + i];
  }
  if (start_pc(lane) == PC_MADD) fp_one(Q.Z);
  old_vm<C, true>(P, Q, start_pc(lane));
  for (int i = 0; i < NL; ++i) { out[i] = P.X.l[i]; out[NL + i] = P.Y.l[i]; out[2 * NL + i] = P.Z.l[i]; }
}
__global__ void __launch_bounds__(256, 1) k(const uint32_t* in, uint32_t* out, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) one_lane(in + (size_t)t * 2 * W, out + (size_t)t * W, t);
}
int main() {
  const int n = 1024;
  std::vector<uint32_t> in((size_t)n * 2 * W), dev((size_t)n * W), host((size_t)n * W);
  uint64_t s = 88172645463325252ull;
  for (auto& v : in) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & LMASK; }
  for (int t = 0; t < n; ++t) for (int e = 0; e < 6; ++e) in[(size_t)t * 2 * W + e * NL + NL - 1] &= 0x1fffu;   // values < 2^741 < p
  uint32_t *di, *dout;
  if (hipMalloc(&di, in.size() * 4) != hipSuccess) { printf("no device\n"); return 2; }
  hipMalloc(&dout, dev.size() * 4);
  hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, di, dout, n);
  if (hipDeviceSynchronize() != hipSuccess) { printf("ACTUAL: kernel fault\n"); return 1; }
  hipMemcpy(dev.data(), dout, dev.size() * 4, hipMemcpyDeviceToHost);
  for (int t = 0; t < n; ++t) one_lane(in.data() + (size_t)t * 2 * W, host.data() + (size_t)t * W, t);
  int bad = 0;
  for (int t = 0; t < n; ++t) if (memcmp(&dev[(size_t)t * W], &host[(size_t)t * W], W * 4) != 0) ++bad;
  printf(bad ? "ACTUAL: %d of %d lanes differ between device and host\n" : "expected results (%d of %d lanes differ): the miscompile does not reproduce in this harness\n", bad, n);
  return bad != 0;
}
