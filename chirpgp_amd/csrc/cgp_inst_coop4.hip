// Lane-cooperative d = 4 kernels (chirp / La Scala models): EKF (cgp_coop4.hpp) and the sigma-point filters and
// continuous-discrete smoother (cgp_coop4_sigma.hpp), the continuous-discrete EKF / EKS (cgp_coop4_cd.hpp).
#include "cgp_coop4_cd.hpp"
namespace cgp {
int dispatch_filter_coop4(const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    return ((io.flags & CGP_DPP_KERNEL) || !ekf4_mfma_fits(io)) ? launch_ekf4_coop(io, ma, st) : dispatch_filter_mfma4(io, ma, st);
}
// the matrix-core sigma-point kernels take collapsible sets of at most 32 groups whose output windows fit a raw buffer
// (cgp_mfma4_sigma.hpp:sgp4_mfma_fits); CGP_DPP_KERNEL keeps the LDS-reduced cooperative kernels
static bool sigma_mfma(uint32_t flags, int64_t T, const ModelArgs& ma) { return !(flags & CGP_DPP_KERNEL) && collapsed_ok(ma) && T * 128 <= kOobMaxBytes; }
int dispatch_filter_coop4_sgp(const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    return sigma_mfma(io.flags, io.T, ma) ? dispatch_filter_mfma4_sgp(io, ma, st) : launch_sgp4_coop<HarmonicLCD<1>>(io, ma, st);
}
int dispatch_filter_coop4_cdsgp(const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    return sigma_mfma(io.flags, io.T, ma) ? dispatch_filter_mfma4_cdsgp(io, ma, st) : launch_cdsgp4_coop<HarmonicSDE<1>>(io, ma, st);
}
int dispatch_smoother_coop4_cdsgp(const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    return sigma_mfma(io.flags, io.T, ma) ? dispatch_smoother_mfma4_cdsgp(io, ma, st) : launch_cdsgps4_coop<HarmonicSDE<1>>(io, ma, st);
}
int dispatch_filter_coop4_cdekf(const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    return (!(io.flags & CGP_DPP_KERNEL) && io.T * 128 <= kOobMaxBytes) ? dispatch_filter_mfma4_cdekf(io, ma, st) : launch_cdekf4_coop(io, ma, st);
}
int dispatch_smoother_coop4_cdeks(const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    return (!(io.flags & CGP_DPP_KERNEL) && io.T * 128 <= kOobMaxBytes) ? dispatch_smoother_mfma4_cdeks(io, ma, st) : launch_cdeks4_coop(io, ma, st);
}
}  // namespace cgp
