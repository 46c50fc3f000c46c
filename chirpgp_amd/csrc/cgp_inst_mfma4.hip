// The matrix-core d = 4 EKF (cgp_mfma4.hpp) in its own translation unit: it is compiled with the max-ILP scheduling
// strategy and VGPR-form MFMA results (Makefile), which suit its single long dependent chain and not the other kernels.
#define CGP_COOP4_HELPERS_ONLY
#include "cgp_mfma4.hpp"
namespace cgp {
int dispatch_filter_mfma4(const FilterIO& io, const ModelArgs& ma, hipStream_t st) { return launch_ekf4_mfma(io, ma, st); }
}  // namespace cgp
