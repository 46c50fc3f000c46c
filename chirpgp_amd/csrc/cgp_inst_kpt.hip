// ekf_for_kpt: linear dynamics of dimension n_harm + 2 with the harmonic measurement h, n_harm = 1..3.
#define CGP_COOP4_HELPERS_ONLY
#include "cgp_dispatch.hpp"
#include "cgp_kpt8.hpp"
namespace cgp {
template <int NH>
static int kpt(bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    using DM = KptLinear<NH + 2>;           // F = I + e_{d-1} e_0^T as d + 1 additions; any other F densely
    // one wavefront per trial: the tile-layout kernel (cgp_kpt8.hpp) unless the caller asks for the generic one or the record is too
    // long for its output windows
    if (wave && !(io.flags & CGP_GENERIC_KERNEL)) {
        const int rc = launch_kpt8_coop<NH>(io, ma, st);
        if (rc != CGP_E_UNSUPPORTED) return rc;
    }
    // one wavefront per trial: the measurement's softplus / sincos in their wave-uniform forms
    return hip_rc(wave ? launch_filter<EkfPredict<DM, true>, KptUpdate<NH, true>>(io, ma, st)
                       : launch_filter<EkfPredict<DM, false>, KptUpdate<NH, false>>(io, ma, st));
}
int dispatch_filter_kpt(int key, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return kpt<1>(wave, io, ma, st);
    case 2: return kpt<2>(wave, io, ma, st);
    case 3: return kpt<3>(wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
