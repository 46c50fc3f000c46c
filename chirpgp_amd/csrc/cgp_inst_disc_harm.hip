// Harmonic chirp LCD models (disc_chirp_lcd, disc_harmonic_chirp_lcd, disc_model_lascala_lcd): n_harm = 1..3 (d = 4, 6, 8).
#include "cgp_dispatch.hpp"
namespace cgp {
int dispatch_filter_disc_harm(int method, int key, bool wave, const FilterIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return filter_disc<HarmonicLCD<1>>(method, wave, io, ma, st);
    case 2: return filter_disc<HarmonicLCD<2>>(method, wave, io, ma, st);
    case 3: return filter_disc<HarmonicLCD<3>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
int dispatch_smoother_disc_harm(int method, int key, bool wave, const SmootherIO& io, const ModelArgs& ma, hipStream_t st) {
    switch (key) {
    case 1: return smoother_disc<HarmonicLCD<1>>(method, wave, io, ma, st);
    case 2: return smoother_disc<HarmonicLCD<2>>(method, wave, io, ma, st);
    case 3: return smoother_disc<HarmonicLCD<3>>(method, wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp
