// The d = 4 matrix-core EKF (cgp_mfma4.hpp: BASELINE C2's filter, the bench kernel) in its own translation unit: VGPR-form MFMA
// results like the other matrix-core kernels, but the DEFAULT scheduling strategy -- with four steps per loop iteration the
// max-ILP strategy measured slower on it (2.72 against 2.65 ms; Makefile).
#define CGP_COOP4_HELPERS_ONLY
#define CGP_EKF4_KERNELS
// Polynomial steps as plain fma() here: every coefficient of these kernels is pinned in a register pair (SpecRegs), so the compiler
// has nothing to rematerialise, and an inline-asm v_fma_f64 (cgp_fastmath.hpp: horner) is opaque to its hazard recogniser, which then
// pads each one with an s_nop: 74 -> 42 s_nop per eight steps of the bench kernel (round 4).
#define CGP_HORNER_PLAIN
#include "cgp_mfma4.hpp"
namespace cgp {
int dispatch_filter_mfma4(const FilterIO& io, const ModelArgs& ma, hipStream_t st) { return launch_ekf4_mfma(io, ma, st); }
int dispatch_filter_kf4_mfma(const FilterIO& io, const ModelArgs& ma, hipStream_t st) { return launch_kf4_mfma(io, ma, st); }
}  // namespace cgp
