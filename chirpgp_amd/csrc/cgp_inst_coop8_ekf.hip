// ekf on the 2- / 3-harmonic model in the 8 x 8 tile layout (cgp_coop8.hpp: ekf8_coop_kernel) -- apart from the other kernels of that
// header (cgp_inst_coop8.hip), which are compiled with plain-fma polynomial steps; this one keeps the inline-asm step.
#define CGP_COOP4_HELPERS_ONLY
#include "cgp_coop8.hpp"
namespace cgp {
int dispatch_filter_coop8_ekf(int n_harm, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (n_harm) {
    case 2: return launch_ekf8_coop<2>(io, ma, st);
    case 3: return launch_ekf8_coop<3>(io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
