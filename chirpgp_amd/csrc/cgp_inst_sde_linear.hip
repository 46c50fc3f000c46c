// Linear SDE (drift A u, constant dispersion): cd_ekf / cd_eks / cd_sgp_* on the linear test models, d = 1..8.
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_filter_sde_linear(int method, int key, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return filter_sde<LinearSDE<1>>(method, wave, io, ma, st);
    case 2: return filter_sde<LinearSDE<2>>(method, wave, io, ma, st);
    case 3: return filter_sde<LinearSDE<3>>(method, wave, io, ma, st);
    case 4: return filter_sde<LinearSDE<4>>(method, wave, io, ma, st);
    case 5: return filter_sde<LinearSDE<5>>(method, wave, io, ma, st);
    case 6: return filter_sde<LinearSDE<6>>(method, wave, io, ma, st);
    case 7: return filter_sde<LinearSDE<7>>(method, wave, io, ma, st);
    case 8: return filter_sde<LinearSDE<8>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
int dispatch_smoother_sde_linear(int method, int key, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return smoother_sde<LinearSDE<1>>(method, wave, io, ma, st);
    case 2: return smoother_sde<LinearSDE<2>>(method, wave, io, ma, st);
    case 3: return smoother_sde<LinearSDE<3>>(method, wave, io, ma, st);
    case 4: return smoother_sde<LinearSDE<4>>(method, wave, io, ma, st);
    case 5: return smoother_sde<LinearSDE<5>>(method, wave, io, ma, st);
    case 6: return smoother_sde<LinearSDE<6>>(method, wave, io, ma, st);
    case 7: return smoother_sde<LinearSDE<7>>(method, wave, io, ma, st);
    case 8: return smoother_sde<LinearSDE<8>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
