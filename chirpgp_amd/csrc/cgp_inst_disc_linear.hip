// Discrete linear model (kf / rts, and ekf / sgp on a linear cond_m_cov): d = 1..8.
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_filter_disc_linear(int method, int key, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return filter_disc<LinearDisc<1>>(method, wave, io, ma, st);
    case 2: return filter_disc<LinearDisc<2>>(method, wave, io, ma, st);
    case 3: return filter_disc<LinearDisc<3>>(method, wave, io, ma, st);
    case 4: return filter_disc<LinearDisc<4>>(method, wave, io, ma, st);
    case 5: return filter_disc<LinearDisc<5>>(method, wave, io, ma, st);
    case 6: return filter_disc<LinearDisc<6>>(method, wave, io, ma, st);
    case 7: return filter_disc<LinearDisc<7>>(method, wave, io, ma, st);
    case 8: return filter_disc<LinearDisc<8>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
int dispatch_smoother_disc_linear(int method, int key, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return smoother_disc<LinearDisc<1>>(method, wave, io, ma, st);
    case 2: return smoother_disc<LinearDisc<2>>(method, wave, io, ma, st);
    case 3: return smoother_disc<LinearDisc<3>>(method, wave, io, ma, st);
    case 4: return smoother_disc<LinearDisc<4>>(method, wave, io, ma, st);
    case 5: return smoother_disc<LinearDisc<5>>(method, wave, io, ma, st);
    case 6: return smoother_disc<LinearDisc<6>>(method, wave, io, ma, st);
    case 7: return smoother_disc<LinearDisc<7>>(method, wave, io, ma, st);
    case 8: return smoother_disc<LinearDisc<8>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
