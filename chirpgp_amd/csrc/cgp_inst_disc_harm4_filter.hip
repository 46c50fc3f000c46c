// Harmonic chirp LCD model with 4 harmonics (d = 10: the reference's bat-call analyses, real_applications/bats/), filters on the
// generic kernels -- a translation unit of its own to keep the build parallel.
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_filter_disc_harm4(int method, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    return filter_disc<HarmonicLCD<4>>(method, wave, io, ma, st);
}
}  // namespace cgp
