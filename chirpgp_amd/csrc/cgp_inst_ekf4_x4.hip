// The d = 4 matrix-core EKF with FOUR trials per wavefront (cgp_mfma4.hpp: ekf4_mfma_x4_kernel, 1024 < B < 9216) in its own translation
// unit: the flags of the matrix-core kernels (VGPR-form MFMA results, default scheduling strategy) WITH machine-level loop-invariant code
// motion, which the one-trial kernel's unit (cgp_inst_ekf4.hip) switches off -- without it this kernel runs 0.338 instead of 0.296 ms per
// 4096 x 500 (round 5, same box).
#define CGP_COOP4_HELPERS_ONLY
#define CGP_EKF4_X4_KERNELS
#define CGP_HORNER_PLAIN
#include "cgp_mfma4.hpp"
namespace cgp {
int launch_ekf4_mfma_x4(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) { return launch_ekf4_mfma_x4_impl(io, ma, stream); }
}  // namespace cgp
