// Harmonic chirp SDE drift (model_chirp, model_harmonic_chirp, model_lascala): n_harm = 1..3.
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_filter_sde_harm(int method, int key, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return filter_sde<HarmonicSDE<1>>(method, wave, io, ma, st);
    case 2: return filter_sde<HarmonicSDE<2>>(method, wave, io, ma, st);
    case 3: return filter_sde<HarmonicSDE<3>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
int dispatch_smoother_sde_harm(int method, int key, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return smoother_sde<HarmonicSDE<1>>(method, wave, io, ma, st);
    case 2: return smoother_sde<HarmonicSDE<2>>(method, wave, io, ma, st);
    case 3: return smoother_sde<HarmonicSDE<3>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
