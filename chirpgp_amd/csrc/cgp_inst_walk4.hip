// d = 4 discrete smoothers walked by the wavefront on the matrix cores (cgp_walk4.hpp): rts / eks / sgp_smoother for the
// linear model and the chirp / La Scala LCD model.
#define CGP_COOP4_HELPERS_ONLY      // cgp_coop4.hpp: helpers only, not a second copy of ekf4_coop_kernel
#include "cgp_walk4.hpp"
#include "cgp_dispatch.hpp"
namespace cgp {
bool walk4_smoother_fits(int64_t T, const ModelArgs& ma) { return walk4_smoother_ok(T, ma); }
int dispatch_smoother_walk4_linear(int method, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    if (method == CGP_S_EKS) return hip_rc(launch_walk4_smoother<EksElement<LinearDisc<4>>>(io, ma, st));
    if (method == CGP_S_SGP) return hip_rc(launch_walk4_smoother<SgpsElement<LinearDisc<4>>>(io, ma, st));
    return CGP_E_UNSUPPORTED;
}
int dispatch_smoother_walk4_harm(int method, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    if (method == CGP_S_EKS) return hip_rc(launch_walk4_smoother<EksElement<HarmonicLCD<1>>>(io, ma, st));
    if (method == CGP_S_SGP) {
        // the collapsed quadrature as a compile-time path where the host has checked the set (cgp_dispatch.hpp)
        if (sgp_collapsible_host<HarmonicLCD<1>>(ma)) return hip_rc(launch_walk4_smoother<SgpsElement<HarmonicLCD<1>, true>>(io, ma, st));
        return hip_rc(launch_walk4_smoother<SgpsElement<HarmonicLCD<1>>>(io, ma, st));
    }
    return CGP_E_UNSUPPORTED;
}
}  // namespace cgp
