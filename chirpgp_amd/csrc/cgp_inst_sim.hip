// Input side of the path: cgp_simulate, cgp_add_noise and the Philox test hook.
#include "cgp_simulate.hpp"
#include "cgp_ctx.hpp"

namespace cgp {
int dispatch_simulate(int model_id, int key, bool wave, const SimIO& io, const ModelArgs& ma, hipStream_t st) {
    if (model_id == CGP_M_LINEAR) {
        switch (key) {
        case 1: return launch_simulate<LinearDisc<1>>(wave, io, ma, st);
        case 2: return launch_simulate<LinearDisc<2>>(wave, io, ma, st);
        case 3: return launch_simulate<LinearDisc<3>>(wave, io, ma, st);
        case 4: return launch_simulate<LinearDisc<4>>(wave, io, ma, st);
        case 5: return launch_simulate<LinearDisc<5>>(wave, io, ma, st);
        case 6: return launch_simulate<LinearDisc<6>>(wave, io, ma, st);
        case 8: return launch_simulate<LinearDisc<8>>(wave, io, ma, st);
        default: return CGP_E_UNSUPPORTED;
        }
    }
    switch (key) {
    case 1: return launch_simulate<HarmonicLCD<1>>(wave, io, ma, st);
    case 2: return launch_simulate<HarmonicLCD<2>>(wave, io, ma, st);
    case 3: return launch_simulate<HarmonicLCD<3>>(wave, io, ma, st);
    default: return CGP_E_UNSUPPORTED;
    }
}
}  // namespace cgp

using namespace cgp;

extern "C" {

int cgp_simulate(cgp_ctx* ctx, const cgp_model* model, const cgp_init* init, double dt, uint64_t seed, int64_t trial0,
                 int64_t B, int64_t T, double* xs, double* ys, uint32_t flags, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (B < 0 || T < 0 || trial0 < 0) return fail(ctx, CGP_E_ARG, "negative B, T or trial0");
    if (B == 0 || T == 0) return CGP_OK;
    if (!xs && !ys) return fail(ctx, CGP_E_ARG, "xs and ys are both NULL");
    if (!model || !model->params) return fail(ctx, CGP_E_ARG, "model or model.params is NULL");
    if (!init || !init->m0 || (!init->P0 && !(flags & CGP_SIM_FIXED_X0))) return fail(ctx, CGP_E_ARG, "init.m0 / P0 must be set");
    if (ys && (!init->H || !init->Xi)) return fail(ctx, CGP_E_ARG, "measurements need init.H and init.Xi");
    int want_params, want_d = model->d, key = model->d;
    switch (model->model_id) {
    case CGP_M_LINEAR: want_params = 2 * model->d * model->d; break;
    case CGP_M_HARMONIC_LCD: want_params = 5; want_d = 2 * model->n_harm + 2; key = model->n_harm; break;
    case CGP_M_LASCALA_LCD: want_params = 2; want_d = 4; key = 1; break;
    default: return fail(ctx, CGP_E_ARG, "cgp_simulate takes a discrete (cond_m_cov) model");
    }
    if (model->d < 1 || model->d > CGP_MAX_D || model->d != want_d) return fail(ctx, CGP_E_ARG, "model.d does not match model_id / n_harm");
    if (model->n_params != want_params) return fail(ctx, CGP_E_ARG, "model.n_params does not match model_id / d");
    if (model->param_stride != 0 && model->param_stride < model->n_params) return fail(ctx, CGP_E_ARG, "model.param_stride < n_params");
    // counter word 2 holds step * ceil(d / 2) + pair
    if ((uint64_t)T * (uint64_t)((model->d + 1) / 2) > 0xFFFFFFFFull) return fail(ctx, CGP_E_ARG, "T too long for the 32-bit step counter");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");

    SimIO io;
    io.H = init->H; io.H_stride = init->H_stride;
    io.Xi = init->Xi; io.Xi_stride = init->Xi_stride;
    io.m0 = init->m0; io.m0_stride = init->m0_stride;
    io.P0 = init->P0; io.P0_stride = init->P0_stride;
    io.seed = seed; io.trial0 = trial0; io.B = B; io.T = T; io.xs = xs; io.ys = ys;
    io.vec_ok = 0; io.flags = flags;
    if (xs && ((uintptr_t)xs & 15) == 0 && ((T * model->d) & 1) == 0) io.vec_ok |= 1;
    if (ys && ((uintptr_t)ys & 15) == 0 && (T & 1) == 0) io.vec_ok |= 2;
    ModelArgs ma;
    ma.params = model->params; ma.param_stride = model->param_stride;
    ma.gamma = nullptr; ma.gamma_stride = 0;
    ma.model_id = model->model_id;
    ma.sg.xi = nullptr; ma.sg.w = nullptr; ma.sg.s = 0; ma.sg.group_start = nullptr; ma.sg.n_groups = 0;
    ma.sg.lds_xi = 0; ma.sg.lds_w = 0; ma.sg.lds_gs = 0; ma.sg.flags = 0;
    ma.dt = dt;
    // One wavefront per trial draws the noise of 64 steps in parallel and pays ~160 replicated instructions per step;
    // one lane per trial pays the ~750 instructions of a step once per 64 trials: crossover near 8 waves per SIMD.
    bool wave = B < (int64_t)ctx->num_cus * 4 * 8;
    if (flags & CGP_WAVE_PER_TRIAL) wave = true;
    if (flags & CGP_THREAD_PER_TRIAL) wave = false;
    const int rc = dispatch_simulate(model->model_id, key, wave, io, ma, (hipStream_t)stream);
    if (rc == CGP_E_UNSUPPORTED) return fail(ctx, rc, "this (model, dimension) combination is not compiled in");
    if (rc == CGP_E_HIP) return fail(ctx, rc, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return rc;
}

int cgp_add_noise(cgp_ctx* ctx, const double* clean, int64_t clean_stride, const double* Xi, int64_t Xi_stride,
                  uint64_t seed, int64_t trial0, int64_t B, int64_t T, double* ys, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (B < 0 || T < 0 || trial0 < 0) return fail(ctx, CGP_E_ARG, "negative B, T or trial0");
    if (B == 0 || T == 0) return CGP_OK;
    if (!clean || !Xi || !ys) return fail(ctx, CGP_E_ARG, "NULL pointer");
    if (clean_stride != 0 && clean_stride < T) return fail(ctx, CGP_E_ARG, "clean_stride < T");
    if ((uint64_t)T > 0x1FFFFFFFEull) return fail(ctx, CGP_E_ARG, "T too long for the 32-bit step counter");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    const int64_t total = B * ((T + 1) / 2), blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, clean, clean_stride, Xi, Xi_stride, seed, trial0, B, T, ys);
    return hipGetLastError() == hipSuccess ? CGP_OK : fail(ctx, CGP_E_HIP, "kernel launch failed");
}

int cgp_debug_philox(cgp_ctx* ctx, const uint32_t* ctr, const uint32_t* key, int64_t n, uint32_t* out, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (n < 0) return fail(ctx, CGP_E_ARG, "negative n");
    if (n == 0) return CGP_OK;
    if (!ctr || !key || !out) return fail(ctx, CGP_E_ARG, "NULL pointer");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(debug_philox_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, ctr, key, n, out);
    return hipGetLastError() == hipSuccess ? CGP_OK : fail(ctx, CGP_E_HIP, "kernel launch failed");
}

}  // extern "C"
