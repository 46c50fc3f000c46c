// Harmonic chirp LCD model with 5 harmonics (d = 12: the reference's bat-call analyses, real_applications/bats/), smoothers on the
// generic kernels -- a translation unit of its own to keep the build parallel.
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_smoother_disc_harm5(int method, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    return smoother_disc<HarmonicLCD<5>>(method, wave, io, ma, st);
}
}  // namespace cgp
