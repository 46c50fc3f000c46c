// Lane-cooperative kernels in the 8 x 8 tile layout (cgp_coop8.hpp): harmonic chirp models with two or three harmonics.
// The sigma-point filter and the smoothers take their polynomial steps as the compiler's own fma (C5's filter 10.68 -> 10.58 ms, its
// smoother 4.69 -> 4.57 ms; profiles/r04_ab_series.txt); the EKF of the same header measured 1.7 % slower with it and is instantiated
// in cgp_inst_coop8_ekf.hip with the inline-asm step.
#define CGP_COOP4_HELPERS_ONLY
#define CGP_HORNER_PLAIN
#include "cgp_coop8.hpp"
namespace cgp {
// the kernel stores through raw buffer windows: one trial's Pf must fit one (cgp_coop4.hpp:kOobMaxBytes)
bool coop8_filter_sgp_ok(int n_harm, int64_t T, const ModelArgs& ma) {
    const int64_t d = 2 * n_harm + 2;
    return (n_harm == 2 || n_harm == 3) && coop8_sigma_ok(ma) && T * d * d * 8 <= kOobMaxBytes;
}
int dispatch_filter_coop8_sgp(int n_harm, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (n_harm) {
    case 2: return launch_sgp8_coop<2>(io, ma, st);
    case 3: return launch_sgp8_coop<3>(io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
// The cooperative smoother keeps 16 step records and the tile's 64 filtering rows in 36.9 KB of static LDS (the split kernel:
// 32 records, 27.9 KB); with the staged sigma-point set beside it a workgroup must stay within 40 KB so that four of them
// (one per SIMD) share a CU's 160 KB (every cubature rule fits; larger sets take the lane-scan kernel).
bool coop8_smoother_ok(int d, int64_t T, const ModelArgs& ma) {
    return d >= 5 && d <= 8 && T * d * d * 8 <= kOobMaxBytes &&
           sigma_lds_bytes(ma, d) + sizeof(double) * (16 * kElemDoubles + 64 * kRowDoubles) + 64 <= 40 * 1024;
}
int dispatch_smoother_coop8_linear(int method, int d, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    if (method != CGP_S_EKS && method != CGP_S_SGP) return CGP_E_UNSUPPORTED;
    const bool sg = method == CGP_S_SGP;
    switch (d) {
    case 5: return hip_rc(sg ? launch_coop8_smoother<SgpsElement<LinearDisc<5>>>(io, ma, st) : launch_coop8_smoother<EksElement<LinearDisc<5>>>(io, ma, st));
    case 6: return hip_rc(sg ? launch_coop8_smoother<SgpsElement<LinearDisc<6>>>(io, ma, st) : launch_coop8_smoother<EksElement<LinearDisc<6>>>(io, ma, st));
    case 7: return hip_rc(sg ? launch_coop8_smoother<SgpsElement<LinearDisc<7>>>(io, ma, st) : launch_coop8_smoother<EksElement<LinearDisc<7>>>(io, ma, st));
    case 8: return hip_rc(sg ? launch_coop8_smoother<SgpsElement<LinearDisc<8>>>(io, ma, st) : launch_coop8_smoother<EksElement<LinearDisc<8>>>(io, ma, st));
    default: return CGP_E_UNSUPPORTED;
    }
}
// the sigma-point elements of the harmonic models are compiled in their collapsed form only: other sets take the lane-scan kernel
bool coop8_smoother_harm_ok(int method, const ModelArgs& ma) {
    return method == CGP_S_EKS || (method == CGP_S_SGP && (ma.sg.flags & CGP_SIGMA_STANDARD) && ma.sg.group_start);
}
int dispatch_smoother_coop8_harm(int method, int n_harm, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    if (method != CGP_S_EKS && method != CGP_S_SGP) return CGP_E_UNSUPPORTED;
    const bool sg = method == CGP_S_SGP;
    switch (n_harm) {
    case 2: return hip_rc(sg ? launch_coop8_smoother<SgpsElement<HarmonicLCD<2>, true>>(io, ma, st) : launch_coop8_smoother<EksElement<HarmonicLCD<2>>>(io, ma, st));
    case 3: return hip_rc(sg ? launch_coop8_smoother<SgpsElement<HarmonicLCD<3>, true>>(io, ma, st) : launch_coop8_smoother<EksElement<HarmonicLCD<3>>>(io, ma, st));
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
