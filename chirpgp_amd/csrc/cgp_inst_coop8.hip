// Lane-cooperative kernels in the 8 x 8 tile layout (cgp_coop8.hpp): harmonic chirp models with two or three harmonics.
#define CGP_COOP4_HELPERS_ONLY
#include "cgp_coop8.hpp"
namespace cgp {
bool coop8_filter_sgp_ok(int n_harm, const ModelArgs& ma) { return (n_harm == 2 || n_harm == 3) && coop8_sigma_ok(ma); }
int dispatch_filter_coop8_sgp(int n_harm, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (n_harm) {
    case 2: return launch_sgp8_coop<2>(io, ma, st);
    case 3: return launch_sgp8_coop<3>(io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
