// The matrix-core d = 4 kernels -- sigma-point filter (cgp_mfma4_sigma.hpp), the continuous-discrete filters and smoothers
// (cgp_mfma4_cd.hpp); the EKF of cgp_mfma4.hpp is instantiated in cgp_inst_ekf4.hip -- in their own translation unit: it is compiled with the max-ILP scheduling strategy and VGPR-form MFMA results (Makefile), which suit
// their single long dependent chain and not the other kernels.
#define CGP_COOP4_HELPERS_ONLY
#include "cgp_mfma4.hpp"
#include "cgp_mfma4_sigma.hpp"
#include "cgp_mfma4_cd.hpp"
namespace cgp {
// (the two sigma-point FILTERS are instantiated in cgp_inst_mfma4_sgpf.hip)
int dispatch_smoother_mfma4_cdsgp(const SmootherIO& io, const ModelArgs& ma, hipStream_t st) { return launch_cdsgps4_mfma<HarmonicSDE<1>>(io, ma, st); }
int dispatch_smoother_split_fixup(const SmootherIO& io, double* junction_err, hipStream_t st) { return launch_smoother_split_fixup(io, junction_err, st); }
int dispatch_filter_mfma4_cdekf(const FilterIO& io, const ModelArgs& ma, hipStream_t st) { return launch_cdekf4_mfma(io, ma, st); }
int dispatch_smoother_mfma4_cdeks(const SmootherIO& io, const ModelArgs& ma, hipStream_t st) { return launch_cdeks4_mfma(io, ma, st); }
}  // namespace cgp
