// Harmonic chirp LCD models (disc_chirp_lcd, disc_harmonic_chirp_lcd, disc_model_lascala_lcd): n_harm = 1..5 (d = 4, 6, 8, 10, 12;
// 4 and 5 harmonics are what the reference's bat-call analyses run: real_applications/bats/myotis_myotis_analysis.py:50,
// eptesicus_nilssonii_analysis.py:49 -- on the generic kernels, the smoothers step by step beyond CGP_TP_MAX_D).
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_filter_disc_harm4(int method, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st);
int dispatch_filter_disc_harm5(int method, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st);
int dispatch_smoother_disc_harm4(int method, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st);
int dispatch_smoother_disc_harm5(int method, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st);
int dispatch_filter_disc_harm(int method, int key, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return filter_disc<HarmonicLCD<1>>(method, wave, io, ma, st);
    case 2: return filter_disc<HarmonicLCD<2>>(method, wave, io, ma, st);
    case 3: return filter_disc<HarmonicLCD<3>>(method, wave, io, ma, st);
    case 4: return dispatch_filter_disc_harm4(method, wave, io, ma, st);           // own translation units (build time)
    case 5: return dispatch_filter_disc_harm5(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
int dispatch_smoother_disc_harm(int method, int key, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return smoother_disc<HarmonicLCD<1>>(method, wave, io, ma, st);
    case 2: return smoother_disc<HarmonicLCD<2>>(method, wave, io, ma, st);
    case 3: return smoother_disc<HarmonicLCD<3>>(method, wave, io, ma, st);
    case 4: return dispatch_smoother_disc_harm4(method, wave, io, ma, st);
    case 5: return dispatch_smoother_disc_harm5(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
