// cgp_dispatch.hpp -- per-model-family dispatch helpers shared by the cgp_inst_*.hip translation units.
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {

// Largest state dimension for which the time-parallel smoother (30 / 108 doubles of affine map per lane at d = 4 / 8)
// is instantiated.
#ifndef CGP_TP_MAX_D
#define CGP_TP_MAX_D 12
#endif

// d >= 6 harmonic models, one lane per fan: the collapsed quadrature is a compile-time property of the kernel (cgp_steps.hpp).
template <class DM> inline bool sgp_collapsible_host(const ModelArgs& ma) {
    return (std::is_same<DM, HarmonicLCD<1>>::value || std::is_same<DM, HarmonicLCD<2>>::value || std::is_same<DM, HarmonicLCD<3>>::value) &&
           (ma.sg.flags & CGP_SIGMA_STANDARD) && ma.sg.group_start;
}

template <class DM>
static int filter_disc(int method, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    using Meas = LinearMeasurement<DM::D>;
    constexpr bool kHarmN = std::is_same<DM, HarmonicLCD<2>>::value || std::is_same<DM, HarmonicLCD<3>>::value;
    switch (method) {
    case CGP_F_EKF:
        return hip_rc(wave ? launch_filter<EkfPredict<DM, true>, Meas>(io, ma, st) : launch_filter<EkfPredict<DM, false>, Meas>(io, ma, st));
    case CGP_F_SGP:
        if constexpr (kHarmN) { if (!wave && sgp_collapsible_host<DM>(ma)) return hip_rc(launch_filter<SgpPredict<DM, false, true>, Meas>(io, ma, st)); }
        return hip_rc(wave ? launch_filter<SgpPredict<DM, true>, Meas>(io, ma, st) : launch_filter<SgpPredict<DM, false>, Meas>(io, ma, st));
    default: return CGP_E_UNSUPPORTED;
    }
}
template <class DM>
static int smoother_disc(int method, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    // One wavefront per trial: time-parallel affine scan unless the caller asks for the step-by-step scan.
    const bool tp = wave && !(io.flags & CGP_SEQUENTIAL_SCAN) && DM::D <= CGP_TP_MAX_D;
    switch (method) {
    case CGP_S_EKS:
        if constexpr (DM::D <= CGP_TP_MAX_D) { if (tp) return hip_rc(launch_tp_smoother<EksElement<DM>>(io, ma, st)); }
        return hip_rc(wave ? launch_smoother<EksStep<DM, true>>(io, ma, st) : launch_smoother<EksStep<DM, false>>(io, ma, st));
    case CGP_S_SGP:
        if constexpr (std::is_same<DM, HarmonicLCD<1>>::value) {
            // d = 4: the time-parallel kernel with the collapsed path alone (the lane-per-trial kernel decides at run time)
            if (tp && sgp_collapsible_host<DM>(ma)) return hip_rc(launch_tp_smoother<SgpsElement<DM, true>>(io, ma, st));
        }
        if constexpr (std::is_same<DM, HarmonicLCD<2>>::value || std::is_same<DM, HarmonicLCD<3>>::value) {
            if (sgp_collapsible_host<DM>(ma)) {
                if (tp) return hip_rc(launch_tp_smoother<SgpsElement<DM, true>>(io, ma, st));
                if (!wave) return hip_rc(launch_smoother<SgpsStep<DM, false, true>>(io, ma, st));
            }
        }
        if constexpr (DM::D <= CGP_TP_MAX_D) { if (tp) return hip_rc(launch_tp_smoother<SgpsElement<DM>>(io, ma, st)); }
        return hip_rc(wave ? launch_smoother<SgpsStep<DM, true>>(io, ma, st) : launch_smoother<SgpsStep<DM, false>>(io, ma, st));
    default: return CGP_E_UNSUPPORTED;
    }
}
template <class SM>
static int filter_sde(int method, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    using Meas = LinearMeasurement<SM::D>;
    switch (method) {
    case CGP_F_CD_EKF:
        return hip_rc(wave ? launch_filter<CdEkfPredict<SM, true>, Meas>(io, ma, st) : launch_filter<CdEkfPredict<SM, false>, Meas>(io, ma, st));
    case CGP_F_CD_SGP:
        return hip_rc(wave ? launch_filter<CdSgpPredict<SM, true>, Meas>(io, ma, st) : launch_filter<CdSgpPredict<SM, false>, Meas>(io, ma, st));
    default: return CGP_E_UNSUPPORTED;
    }
}
template <class SM>
static int smoother_sde(int method, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (method) {
    case CGP_S_CD_EKS:
        return hip_rc(wave ? launch_smoother<CdEksStep<SM, true>>(io, ma, st) : launch_smoother<CdEksStep<SM, false>>(io, ma, st));
    case CGP_S_CD_SGP:
        return hip_rc(wave ? launch_smoother<CdSgpsStep<SM, true>>(io, ma, st) : launch_smoother<CdSgpsStep<SM, false>>(io, ma, st));
    default: return CGP_E_UNSUPPORTED;
    }
}

}  // namespace cgp
