// FFT part of the C ABI -- placeholder while the NTT kernels are being written.
#include "common_host.hpp"
using namespace mnt753;
struct mnt753_domain { int curve; size_t m; };
extern "C" {
int mnt753_domain_create(int, size_t, mnt753_domain**) { return set_error(MNT753_EDOMAIN, "fft: not built yet"); }
int mnt753_domain_free(mnt753_domain*) { return 0; }
size_t mnt753_domain_size(const mnt753_domain* d) { return d ? d->m : 0; }
int mnt753_fft(mnt753_domain*, int, uint64_t*, void*) { return set_error(MNT753_EDOMAIN, "fft: not built yet"); }
int mnt753_divide_by_z_on_coset(mnt753_domain*, uint64_t*, void*) { return set_error(MNT753_EDOMAIN, "fft: not built yet"); }
int mnt753_vec_muleq(int, uint64_t*, const uint64_t*, size_t, void*) { return set_error(MNT753_EDOMAIN, "fft: not built yet"); }
int mnt753_vec_subeq(int, uint64_t*, const uint64_t*, size_t, void*) { return set_error(MNT753_EDOMAIN, "fft: not built yet"); }
int mnt753_compute_h(mnt753_domain*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, void*) { return set_error(MNT753_EDOMAIN, "fft: not built yet"); }
}
