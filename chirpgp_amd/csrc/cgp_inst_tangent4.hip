// cgp_ekf_nll_grad: the EKF's NLL and its exact gradient by forward tangents through the scan (cgp_tangent4.hpp).
#include "cgp_tangent4.hpp"
#include "cgp_ctx.hpp"
using namespace cgp;

extern "C" int cgp_ekf_nll_grad(cgp_ctx* ctx, const cgp_model* model, const cgp_init* init, double dt,
                                const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index, int64_t B, int64_t T,
                                const double* dirs, int32_t n_dir, double* nll, double* grad, uint32_t flags, void* stream) {
    (void)flags;
    if (!ctx) return CGP_E_ARG;
    if (B < 0 || T < 0 || n_dir < 0) return fail(ctx, CGP_E_ARG, "negative B, T or n_dir");
    if (B == 0 || n_dir == 0) return CGP_OK;
    if (!model || !model->params) return fail(ctx, CGP_E_ARG, "model or model.params is NULL");
    const bool chirp = model->model_id == CGP_M_HARMONIC_LCD && model->n_harm == 1 && model->n_params == 5;
    const bool lascala = model->model_id == CGP_M_LASCALA_LCD && model->n_params == 2;
    if ((!chirp && !lascala) || model->d != 4)
        return fail(ctx, CGP_E_UNSUPPORTED, "cgp_ekf_nll_grad is built for the d = 4 chirp and La Scala LCD models");
    if (model->param_stride != 0 && model->param_stride < model->n_params) return fail(ctx, CGP_E_ARG, "model.param_stride < n_params");
    if (!init || !init->H || !init->Xi || !init->m0 || !init->P0) return fail(ctx, CGP_E_ARG, "init.H / Xi / m0 / P0 must be set");
    if (T > 0 && !ys) return fail(ctx, CGP_E_ARG, "ys is NULL");
    if (ys_stride < 0 || ys_repeat < 1) return fail(ctx, CGP_E_ARG, "ys_stride must be >= 0 and ys_repeat >= 1");
    if (!dirs || !nll || !grad) return fail(ctx, CGP_E_ARG, "dirs / nll / grad must be set");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    std::lock_guard<std::recursive_mutex> launches(ctx->launch_mutex);
    TangentIO io;
    io.H = init->H; io.H_stride = init->H_stride; io.Xi = init->Xi; io.Xi_stride = init->Xi_stride;
    io.m0 = init->m0; io.m0_stride = init->m0_stride; io.P0 = init->P0; io.P0_stride = init->P0_stride;
    io.ys = ys; io.ys_stride = ys_stride; io.ys_repeat = ys_repeat; io.ys_index = ys_index;
    io.dirs = dirs; io.B = B; io.T = T; io.n_dir = n_dir; io.nll = nll; io.grad = grad;
    ModelArgs ma;
    ma.params = model->params; ma.param_stride = model->param_stride; ma.gamma = nullptr; ma.gamma_stride = 0;
    ma.model_id = model->model_id; ma.dt = dt;
    ma.sg.xi = nullptr; ma.sg.w = nullptr; ma.sg.s = 0; ma.sg.group_start = nullptr; ma.sg.n_groups = 0;
    ma.sg.lds_xi = 0; ma.sg.lds_w = 0; ma.sg.lds_gs = 0; ma.sg.lds_tab = 0; ma.sg.flags = 0u;
    if (launch_ekf4_tangent(io, ma, (hipStream_t)stream) != hipSuccess)
        return fail(ctx, CGP_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return CGP_OK;
}
