// VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: pick the multiply primitive for the 753-bit Montgomery multiplier
// (DESIGN.md "field layer").  Measures cycles per wave-instruction for the
// integer / fp64 candidates at 1, 2 and 4 waves per SIMD.
//
// Build:  hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
// Run:    ./valu_rates            (prints one line per instruction x occupancy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;   // independent dependency chains per lane

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U24, ADD_CO_PAIR,
          FMA_F64, LSHL_ADD_U64, ADD3_U32, MOV_B32, ACC_RW, ADD_U32, MAD_U64_CARRY, NOPS };
static const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24",
  "v_mul_hi_u32_u24", "v_add_co+v_addc_co (pair)", "v_fma_f64", "v_lshl_add_u64", "v_add3_u32",
  "v_mov_b32", "v_accvgpr_write+read (pair)", "v_add_u32", "v_mad_u64_u32+v_addc (pair)"};

template <int OP>
__global__ void __launch_bounds__(1024) k(uint64_t* out, uint32_t seed, unsigned long long* cyc) {
  uint32_t a = seed * (threadIdx.x + 1) | 1, b = seed ^ (0x9e3779b9u * (threadIdx.x + 7));
  uint64_t acc[CHAINS];
  double dacc[CHAINS];
  uint32_t w[CHAINS], top[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { acc[c] = a + c; dacc[c] = (double)(a + c); w[c] = b + c; top[c] = c; }
  double da = (double)a * 1e-9, db = (double)b * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == MAD_U64_U32)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
      else if (OP == MUL_LO_U32)
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[c]) : "v"(a));
      else if (OP == MUL_HI_U32)
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(w[c]) : "v"(a));
      else if (OP == MAD_U32_U24)
        asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(w[c]) : "v"(a), "v"(b));
      else if (OP == MUL_HI_U24)
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(w[c]) : "v"(a));
      else if (OP == ADD_CO_PAIR)
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                     : "+v"(w[c]), "+v"(top[c]) : "v"(a), "v"(b) : "vcc");
      else if (OP == FMA_F64)
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[c]) : "v"(da), "v"(db));
      else if (OP == LSHL_ADD_U64)
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
      else if (OP == ADD3_U32)
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(w[c]) : "v"(a), "v"(b));
      else if (OP == MOV_B32)
        asm volatile("v_mov_b32 %0, %1" : "=v"(w[c]) : "v"(top[c]));
      else if (OP == ACC_RW)
        asm volatile("v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "+v"(w[c]) :: "a0");
      else if (OP == ADD_U32)
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[c]) : "v"(a));
      else if (OP == MAD_U64_CARRY)
        asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                     : "+v"(acc[c]), "+v"(top[c]) : "v"(a), "v"(b) : "vcc");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t r = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) r += acc[c] + (uint64_t)dacc[c] + w[c] + top[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(int waves_per_simd, uint64_t* d_out, unsigned long long* d_cyc, int n_cu) {
  int threads = 256 * waves_per_simd;  // 4 SIMDs x waves x 64 lanes, one block per CU
  if (threads > 1024) threads = 1024;
  int blocks_per_cu = (256 * waves_per_simd) / threads;
  int blocks = n_cu * blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, d_cyc);  // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, d_cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
  double n_inst = (double)ITERS * CHAINS;  // wave-instructions (or pairs) per wave
  // cycles per instruction per SIMD = elapsed cycles / (instructions issued on that SIMD)
  double cyc_per_inst_simd = avg / (n_inst * waves_per_simd);
  printf("%-30s waves/SIMD=%d  kernel=%.3f ms  memtime_ticks/wave=%.0f  ticks/inst/wave=%.2f  ticks/inst/SIMD=%.2f  (wall-derived GHz-cycles/inst/SIMD @2.4GHz=%.2f)\n",
         names[OP], waves_per_simd, ms, avg, avg / n_inst, cyc_per_inst_simd,
         ms * 1e-3 * 2.4e9 / (n_inst * waves_per_simd));
  return 0;
}

template <int OP>
int run_all(uint64_t* d_out, unsigned long long* d_cyc, int n_cu) {
  for (int w : {1, 2, 4, 8}) if (run<OP>(w, d_out, d_cyc, n_cu)) return 1;
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int n_cu = p.multiProcessorCount;
  uint64_t* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 1024 * 8));
  CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 8));
  if (run_all<MAD_U64_U32>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<MAD_U64_CARRY>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<MUL_LO_U32>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<MUL_HI_U32>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<MAD_U32_U24>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<MUL_HI_U24>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<ADD_CO_PAIR>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<ADD_U32>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<ADD3_U32>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<FMA_F64>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<LSHL_ADD_U64>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<MOV_B32>(d_out, d_cyc, n_cu)) return 1;
  if (run_all<ACC_RW>(d_out, d_cyc, n_cu)) return 1;
  return 0;
}
