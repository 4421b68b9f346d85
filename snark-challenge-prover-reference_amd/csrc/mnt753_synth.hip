// Uniform scalars in the reference's wire format, generated on the host (mnt753_synth_scalars, include/mnt753_hip.h): the scalars of
// the warm-up MSMs at parameter-load time and the default second random element of `main_hip complete` (main.cpp:312-319 draws it
// with libff's random_element).  The synthetic base points and the expected MSM value are test infrastructure and live in
// libmnt753_hip_test.so (mnt753_synth_points.hip).
#include <cstring>

#include "common_host.hpp"
#include "host_field.hpp"

using namespace mnt753;
using namespace mnt753::host;

namespace {
inline uint64_t splitmix64(uint64_t& state) {
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
template <int FRM>
void synth_scalars_t(uint64_t seed, size_t n, uint64_t* out) {
  HFp<FRM> r2 = HFp<FRM>::from_words(FPC[FRM].r2_64);
  for (size_t i = 0; i < n; ++i) {
    uint64_t st = seed ^ (0xC0FFEE1234567ull + i * 0xD1B54A32D192ED03ull);
    HFp<FRM> v;
    do {  // uniform below the modulus by rejection on 753-bit draws
      for (int k = 0; k < 12; ++k) v.l[k] = splitmix64(st);
      v.l[11] &= (1ull << (753 - 64 * 11)) - 1;
    } while (HFp<FRM>::geq_p(v.l));
    HFp<FRM> m = v * r2;  // integer -> Montgomery
    memcpy(out + 12 * i, m.l, 96);
  }
}

}  // namespace

extern "C" int mnt753_synth_scalars(int curve, uint64_t seed, size_t n, uint64_t* out) {
  if (curve < 0 || curve > 1 || (n && !out)) return set_error(MNT753_EINVAL, "synth_scalars: bad argument");
  if (curve == MNT753_CURVE_MNT4753) synth_scalars_t<MOD_A>(seed, n, out); else synth_scalars_t<MOD_B>(seed, n, out);
  return 0;
}
