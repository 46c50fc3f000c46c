// The two matrix-core sigma-point FILTERS at d = 4 (sgp_filter: cgp_mfma4_sigma.hpp; cd_sgp_filter: cgp_mfma4_cd.hpp) in a translation
// unit of their own, because they want their polynomial steps as the compiler's own fma (every coefficient is pinned in a register:
// FanRegs / SoftplusRegs) -- C3's filter 6.02 -> 5.88 ms, C4's 93.6 -> 91.9 ms -- while the continuous-discrete smoother and the
// cd_ekf / cd_eks kernels of cgp_inst_mfma4.hip measured 3 - 4 % SLOWER with it (profiles/r04_ab_series.txt) and keep the inline-asm step.
#define CGP_COOP4_HELPERS_ONLY
#define CGP_HORNER_PLAIN
#define CGP_NO_CD_EKF_KERNELS
#include "cgp_mfma4.hpp"
#include "cgp_mfma4_sigma.hpp"
#include "cgp_mfma4_cd.hpp"
namespace cgp {
int dispatch_filter_mfma4_sgp(const FilterIO& io, const ModelArgs& ma, hipStream_t st) { return launch_sgp4_mfma<HarmonicLCD<1>>(io, ma, st); }
int dispatch_filter_mfma4_cdsgp(const FilterIO& io, const ModelArgs& ma, hipStream_t st) { return launch_cdsgp4_mfma<HarmonicSDE<1>>(io, ma, st); }
}  // namespace cgp
