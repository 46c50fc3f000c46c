// Harmonic chirp LCD model with 4 harmonics (d = 10: the reference's bat-call analyses, real_applications/bats/), smoothers on the
// generic kernels -- a translation unit of its own to keep the build parallel.
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_smoother_disc_harm4(int method, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    return smoother_disc<HarmonicLCD<4>>(method, wave, io, ma, st);
}
}  // namespace cgp
