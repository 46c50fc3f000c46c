// The sigma-point filter of cgp_lane4.hpp in its own translation unit: built WITHOUT CGP_HORNER_SGPR -- with the polynomial constants as
// scalar operands the kernel fits two wavefronts per SIMD (216 registers instead of 255 + spills) but spills 360 scalars to lanes, and the
// CRLB launch (262 144 x 500, GH-3) takes 40.5 ms instead of 28.6.
#define CGP_COOP4_HELPERS_ONLY      // OobWindow, not a second copy of ekf4_coop_kernel
#include "cgp_dispatch.hpp"
#include "cgp_lane4.hpp"
namespace cgp {
int dispatch_filter_lane4_sgp(const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    using DM = HarmonicLCD<1>;
    using Meas = LinearMeasurement<4>;
    if (sigma_lds_bytes(ma, 4) > (size_t)kLane4SigLdsMaxBytes) return CGP_E_UNSUPPORTED;
    // the collapsed quadrature alone where the host has checked the set for it (fewer registers, less code)
    if (sgp_collapsible_host<DM>(ma)) return hip_rc(launch_lane4_filter<SgpPredictLane<DM, true>, Meas>(io, ma, st));
    return hip_rc(launch_lane4_filter<SgpPredictLane<DM>, Meas>(io, ma, st));
}
}  // namespace cgp
