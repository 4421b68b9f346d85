// Development: only the level kernels of one group, for quick compiles and static instruction counts (tools/isa_mix.py).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I snark-challenge-prover-reference_amd/csrc [-D...] -c tools/experiments/diet/pair_only.hip -o build_exp/pair_only.o
#include "msm_kernels.hip.h"
namespace mnt753 {
#ifndef DIET_GROUP
#define DIET_GROUP Mnt4G1
#endif
template __global__ void k_pair_level<DIET_GROUP, false, false, false>(const uint32_t*, const uint32_t*, const uint4*, size_t, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint4*, size_t, uint4*, uint32_t, uint32_t, const uint32_t*, uint32_t*, const uint32_t*);
template __global__ void k_pair_level<DIET_GROUP, true, false, false>(const uint32_t*, const uint32_t*, const uint4*, size_t, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint4*, size_t, uint4*, uint32_t, uint32_t, const uint32_t*, uint32_t*, const uint32_t*);
}
