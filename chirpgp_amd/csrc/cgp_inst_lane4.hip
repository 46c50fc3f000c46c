// Large batches at d = 4, one lane per trial (cgp_lane4.hpp): the chirp / La Scala LCD models -- EKF and the smoothers, built with the
// polynomial constants as scalar operands (CGP_HORNER_SGPR: means-only EKF 1.73 -> 1.66 ms, cd_eks 9.5 -> 8.9 ms at 262 144 x 500).
#define CGP_COOP4_HELPERS_ONLY      // OobWindow, not a second copy of ekf4_coop_kernel
#include "cgp_dispatch.hpp"
#include "cgp_lane4.hpp"
namespace cgp {
int dispatch_filter_lane4_sgp(const FilterIO& io, const ModelArgs& ma, hipStream_t st);
int dispatch_filter_lane4(int method, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    using DM = HarmonicLCD<1>;
    using Meas = LinearMeasurement<4>;
    switch (method) {
    case CGP_F_EKF: return hip_rc(launch_lane4_filter<EkfPredict<DM, false>, Meas>(io, ma, st));
    case CGP_F_SGP: return dispatch_filter_lane4_sgp(io, ma, st);          // own translation unit: cgp_inst_lane4_sgp.hip
    default: return CGP_E_UNSUPPORTED;
    }
}
int dispatch_smoother_lane4(int method, int model_id, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    if (method == CGP_S_EKS && (model_id == CGP_M_HARMONIC_LCD || model_id == CGP_M_LASCALA_LCD))
        return hip_rc(launch_lane4_smoother<EksStep<HarmonicLCD<1>, false>>(io, ma, st));
    if (method == CGP_S_CD_EKS && model_id == CGP_M_HARMONIC_SDE)
        return hip_rc(launch_lane4_smoother<CdEksStep<HarmonicSDE<1>, false>>(io, ma, st));
    return CGP_E_UNSUPPORTED;
}
}  // namespace cgp
