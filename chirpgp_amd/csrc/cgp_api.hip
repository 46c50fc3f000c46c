// cgp_api.hip -- the C-ABI of include/chirpgp_hip.h: argument checks, launch-shape choice, dispatch.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "cgp_kernels.hpp"
#include "cgp_ctx.hpp"

namespace cgp {

// Launch-shape choice.  One wavefront per trial is the latency-optimal shape while the batch is about the number of
// SIMDs (1024): a step then costs its dependent-instruction chain once.  One lane per trial costs more per step but
// carries 64 trials per wave, so it wins as soon as enough wavefronts would have to share a SIMD -- how many depends on how
// much faster the lane-cooperative (matrix-core) kernel's step is than the one-lane step.  Crossovers measured on MI355X in
// round 3 (bench.py --batch B --flags 2 | 4; wave-per-trial time grows linearly with B beyond 1024, the lane-per-trial time is
// flat until B ~ 64 K), in trials per SIMD:
//     EKF, chirp / La Scala d = 4 (MFMA, four trials per wave above 1024)   1.25 ms vs 2.58 at 8 K, 2.29 vs 2.65 at 16 K, 3.33 vs 2.89 at 24 K   -> 20
//     EKF, 2 / 3 harmonics d = 6 / 8 (tile layout)                          0.93 vs 1.81 at 4 K, 1.73 vs 1.82 at 8 K, 3.29 vs 1.88 at 16 K        -> 8
//     sigma-point filter, d = 4 (MFMA sums)                                 1.38 vs 8.17 at 4 K, 5.12 vs 8.18 at 16 K, 6.2 vs 5.1 at 32 K         -> 24
//     sigma-point filter, d = 6 / 8 (tile layout)                           2.52 vs 6.95 at 4 K, 9.58 vs 6.99 at 16 K                             -> 11
//     cd_ekf / cd_eks, d = 4 (MFMA)                                         1.41 vs 1.38 / 1.38 vs 1.90 at 4 K, 4.95 vs 1.40 / 4.89 vs 2.07 at 16 K -> 4 / 5
//     cd_sgp filter / smoother, d = 4 (MFMA)                                2.3 vs 28.6 at 4 K, 9.2 vs 28.7 at 16 K                               -> 48
//     discrete smoothers on the cooperative walks (d = 4 .. 8)              never slower than one lane per trial: 0.36 vs 9.98 ms (sigma-point d = 4,
//                                                                           4 K), 2.77 vs 18.4 (EKS d = 8, 16 K), 9.4 vs 10.6 (EKS d = 4, 256 K)   -> always
//     everything on the generic kernels                                     as measured in round 1: 2.5 (EKF-type), 8 (sigma-point), 16 (affine scan)
struct ShapeLimit { int num, den; };                      // one wavefront per trial while B * den < num * SIMDs; num < 0: always
static bool choose_wave(const cgp_ctx* ctx, int64_t B, uint32_t flags, ShapeLimit limit, const cgp_sigma* sg = nullptr) {
    // the wave-per-trial shapes stage the sigma-point set in LDS; a set that does not fit runs one lane per trial
    if (sg && SigmaSet::stage_bytes(sg->s, sg->d, sg->n_groups, sg->group_start != nullptr) > (size_t)kSigLdsMaxBytes) return false;
    if (flags & CGP_WAVE_PER_TRIAL) return true;
    if (flags & CGP_THREAD_PER_TRIAL) return false;
    if (limit.num < 0) return true;
    const int64_t simds = (int64_t)(ctx ? ctx->num_cus : 256) * 4;
    return B * limit.den < (int64_t)limit.num * simds;
}

static int check_model(cgp_ctx* ctx, const cgp_model* m, bool sde, bool need_sigma, const cgp_sigma* sg) {
    if (!m || !m->params) return fail(ctx, CGP_E_ARG, "model or model.params is NULL");
    // dimensions above 8 are compiled in for the harmonic LCD model alone (4 and 5 harmonics: d = 10, 12)
    const int max_d = (m->model_id == CGP_M_HARMONIC_LCD) ? CGP_MAX_D : 8;
    if (m->d < 1 || m->d > max_d) return fail(ctx, CGP_E_UNSUPPORTED, "state dimension outside 1.." + std::to_string(max_d) + " for this model");
    int want_params = -1, want_d = m->d;
    switch (m->model_id) {
    case CGP_M_LINEAR: case CGP_M_KPT: want_params = 2 * m->d * m->d; if (m->model_id == CGP_M_KPT) want_d = m->n_harm + 2; break;
    case CGP_M_HARMONIC_LCD: want_params = 5; want_d = 2 * m->n_harm + 2; break;
    case CGP_M_LASCALA_LCD: want_params = 2; want_d = 4; break;
    case CGP_M_LINEAR_SDE: want_params = m->d * m->d; break;
    case CGP_M_HARMONIC_SDE: want_params = 3; want_d = 2 * m->n_harm + 2; break;
    default: return fail(ctx, CGP_E_ARG, "unknown model_id");
    }
    if (m->n_params != want_params) return fail(ctx, CGP_E_ARG, "model.n_params does not match model_id / d");
    if (m->d != want_d) return fail(ctx, CGP_E_ARG, "model.d does not match model_id / n_harm");
    if (m->param_stride != 0 && m->param_stride < m->n_params) return fail(ctx, CGP_E_ARG, "model.param_stride < n_params");
    const bool is_sde = m->model_id == CGP_M_LINEAR_SDE || m->model_id == CGP_M_HARMONIC_SDE;
    if (sde != is_sde) return fail(ctx, CGP_E_ARG, sde ? "continuous-discrete methods need an SDE model" : "discrete methods need a discrete (cond_m_cov) model");
    if (sde && !m->gamma) return fail(ctx, CGP_E_ARG, "SDE methods need model.gamma = b b^T");
    if (need_sigma) {
        if (!sg || !sg->xi || !sg->w || sg->s < 1) return fail(ctx, CGP_E_ARG, "sigma-point methods need a cgp_sigma");
        if (sg->d != m->d) return fail(ctx, CGP_E_ARG, "cgp_sigma.d != model.d");
        if (sg->group_start && (sg->n_groups < 1 || sg->n_groups > sg->s)) return fail(ctx, CGP_E_ARG, "cgp_sigma.n_groups outside 1..s");
    }
    return CGP_OK;
}

static ModelArgs model_args(const cgp_model* m, const cgp_sigma* sg, double dt, uint32_t flags) {
    ModelArgs a;
    a.params = m->params; a.param_stride = m->param_stride;
    a.gamma = m->gamma; a.gamma_stride = m->gamma_stride;
    a.model_id = m->model_id;
    a.sg.xi = sg ? sg->xi : nullptr; a.sg.w = sg ? sg->w : nullptr; a.sg.s = sg ? sg->s : 0;
    a.sg.group_start = sg ? sg->group_start : nullptr; a.sg.n_groups = (sg && sg->group_start) ? sg->n_groups : 0;
    a.sg.lds_xi = 0; a.sg.lds_w = 0; a.sg.lds_gs = 0; a.sg.lds_tab = 0;
    a.sg.flags = (sg && !(flags & CGP_LITERAL_SIGMA_SUM)) ? sg->flags : 0u;
    a.dt = dt;
    return a;
}

template <int FN>
__global__ void __launch_bounds__(256) gaussian_expectation_kernel(const double* __restrict__ ms, const double* __restrict__ sd,
                                                                   int64_t n, int64_t stride, const double* __restrict__ xi,
                                                                   const double* __restrict__ w, int order, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double m = ms[i * stride], s = sd[i * stride];
        double acc = 0.0;
        for (int p = 0; p < order; p++) {
            const double x = fma(s, xi[p], m);
            double f;
            if constexpr (FN == CGP_FN_SOFTPLUS) f = log(exp(x) + 1.0);        // models.py:50, the naive form as is
            else if constexpr (FN == CGP_FN_EXP) f = exp(x);
            else if constexpr (FN == CGP_FN_SQUARE) f = x * x;
            else f = x;
            acc = fma(w[p], f, acc);
        }
        out[i] = acc;
    }
}

// cgp_squared_error_sums: lane = time step (64 consecutive steps of a trial are one contiguous 64 d doubles), the four wavefronts of a
// workgroup take every fourth trial of its slab of 512; per-thread sums, then LDS across the wavefronts and one float64 atomic per
// (component, moment, step) and slab.  Reads every byte of a and r once: HBM-bound.
struct ErrComps { int c[8]; int n; };
__global__ void __launch_bounds__(256) squared_error_sums_kernel(const double* __restrict__ a, const double* __restrict__ r, int64_t B, int64_t T, int d,
                                                                 ErrComps comps, double* __restrict__ sums) {
    __shared__ double part[4][16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 64 + lane;
    const int64_t b0 = (int64_t)blockIdx.y * 512, b1 = (b0 + 512 < B) ? b0 + 512 : B;
    double acc[16];
    CGP_UNROLL for (int k = 0; k < 16; k++) acc[k] = 0.0;
    if (t < T) {
        for (int64_t b = b0 + w; b < b1; b += 4) {
            const double* __restrict__ pa = a + (b * T + t) * d;
            const double* __restrict__ pr = r + (b * T + t) * d;
            CGP_UNROLL for (int c = 0; c < 8; c++) {
                if (c < comps.n) {
                    const double e = pa[comps.c[c]] - pr[comps.c[c]], e2 = e * e;
                    acc[2 * c] += e2;
                    acc[2 * c + 1] = fma(e2, e2, acc[2 * c + 1]);
                }
            }
        }
    }
    CGP_UNROLL for (int k = 0; k < 16; k++) part[w][k][lane] = acc[k];
    __syncthreads();
    if (w == 0 && t < T) {
        CGP_UNROLL for (int k = 0; k < 16; k++) {
            if (k < 2 * comps.n) {
                const double v = (part[0][k][lane] + part[1][k][lane]) + (part[2][k][lane] + part[3][k][lane]);
                atomicAdd(sums + (int64_t)k * T + t, v);          // sums[c][m][t], k = 2 c + m
            }
        }
    }
}

// cgp_smoother_select behind a kernel that writes full rows only: the selected marginal read back out of mss / Pss
__global__ void __launch_bounds__(256) smooth_select_kernel(const double* __restrict__ mss, const double* __restrict__ Pss, int64_t n, int d, SmoothSel sel) {
    __shared__ double ghrule[2 * kGhMaxOrder];
    sel_stage_rule(sel, ghrule, threadIdx.x);
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        sel_write(sel, ghrule, i, mss[i * d + sel.comp], Pss[i * d * d + sel.comp * (d + 1)]);
}

__global__ void __launch_bounds__(256) debug_math_kernel(int op, const double* __restrict__ x, int64_t n,
                                                         double* __restrict__ o0, double* __restrict__ o1) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        double a = 0.0, b = 0.0;
        switch (op) {
        case 0: a = fast_exp(v); break;
        case 1: a = fast_log_ge1(v); break;
        case 2: fast_sincos(v, a, b); break;
        case 3: a = rcp_nr(v); break;
        case 7: softplus_pair_wide(v, a, b); break;
        case 9: a = rcp_nr1(v); break;
        case 10: { bool ok; softplus_pair_any(v, a, b, ok); if (!ok) { a = __builtin_nan(""); b = a; } break; }      // (not ok -> NaN: the caller's fallback)
        case 11: softplus_pair_mid(v, a, b); break;                                                                   // (valid for |v| <= 2)
        default: softplus_pair(v, a, b); break;
        }
        o0[i] = a;
        if (o1) o1[i] = b;
    }
}

// The wave-uniform variants: one input element per wavefront (all 64 lanes evaluate the same argument).
__global__ void __launch_bounds__(64) debug_math_uniform_kernel(int op, const double* __restrict__ x, int64_t n,
                                                                double* __restrict__ o0, double* __restrict__ o1) {
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const double v = x[i];
        double a, b;
        if (op == 5) softplus_pair_uniform(v, a, b);
        else if (op == 8) { SpecRegs R; R.init(); double q; const double t = exp_neg_lean(R, v); softplus_tail_lean(R, t, q, b); a = fma(q, t, v); }
        else fast_sincos_uniform(v, a, b);
        if (threadIdx.x == 0) { o0[i] = a; if (o1) o1[i] = b; }
    }
}

// Whether a d = 4 sigma-point launch takes the matrix-core kernel (cgp_inst_coop4.hip: sigma_mfma) -- the one that knows segments.
static bool sigma_mfma_taken(uint32_t flags, int64_t T, const ModelArgs& ma) {
    return !(flags & CGP_DPP_KERNEL) && (ma.sg.flags & CGP_SIGMA_STANDARD) && ma.sg.group_start && ma.sg.n_groups >= 1 && ma.sg.n_groups <= 32 && T * 128 <= 0x7FFFFF00LL;
}

// Fix-up pass of a time-split filter launch (one workgroup per trial):
//   * junction_err[b] = the largest relative mismatch, over the junctions of trial b, between the state a segment's burn-in arrived
//     at and the state the segment before it ended with (max |difference| / max |reference|, mean and covariance separately) --
//     everything the segment wrote afterwards is at most this far from the sequential filter's rows (the filter forgets its
//     initial condition; the caller decides what mismatch it accepts);
//   * the cumulative NLL of segment s > 0 starts at 0: add the totals of the segments before it (or, NLL-final mode, sum the totals).
__global__ void __launch_bounds__(64) filter_split_fixup_kernel(FilterIO io, int d, double* __restrict__ junction_err) {
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const int n = d + d * d;
    const double* __restrict__ rec = io.seg_state + b * io.segs * io.seg_stride;
    double worst = 0.0;
    for (int s = 1; s < io.segs; s++) {
        const double* start = rec + s * io.seg_stride;                   // (m, P) after the burn-in of segment s
        const double* end = rec + (s - 1) * io.seg_stride + n;           // (m, P) after the last step of segment s - 1
        double dm = 0.0, rm = 0.0, dp = 0.0, rp = 0.0;
        bool nan = false;
        for (int i = 0; i < n; i++) {
            const double a = start[i], r = end[i], e = fabs(a - r);
            nan = nan || !(e == e);
            if (i < d) { dm = fmax(dm, e); rm = fmax(rm, fabs(r)); } else { dp = fmax(dp, e); rp = fmax(rp, fabs(r)); }
        }
        double err = fmax(rm > 0.0 ? dm / rm : dm, rp > 0.0 ? dp / rp : dp);
        if (nan) err = __builtin_inf();                                  // a NaN on either side of a junction: the split run is not the sequential one
        worst = fmax(worst, err);
    }
    if (lane == 0) junction_err[b] = worst;
    if (!io.nll) return;
    const int tot = 2 * n;
    if (io.flags & CGP_NLL_FINAL_ONLY) {
        if (lane == 0) { double sum = 0.0; for (int s = 0; s < io.segs; s++) sum += rec[s * io.seg_stride + tot]; io.nll[b] = sum; }
        return;
    }
    double offset = 0.0;
    for (int s = 1; s < io.segs; s++) {
        offset += rec[(s - 1) * io.seg_stride + tot];
        const int64_t t0 = (int64_t)s * io.seg_len, t1 = (t0 + io.seg_len < io.T) ? t0 + io.seg_len : io.T;
        for (int64_t t = t0 + lane; t < t1; t += 64) io.nll[b * io.T + t] += offset;
    }
}

}  // namespace cgp

using namespace cgp;

extern "C" {

int cgp_version(void) { return CGP_VERSION; }

int cgp_create(cgp_ctx** out, int device) {
    if (!out) return CGP_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return CGP_E_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CGP_E_HIP;
    cgp_ctx* c = new cgp_ctx;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    return CGP_OK;
}

void cgp_destroy(cgp_ctx* ctx) {
    if (!ctx) return;
    {
        DeviceScope on_device(ctx->device);
        if (ctx->counters_mem) (void)hipFree(ctx->counters_mem);
        for (auto& kv : ctx->ws) if (kv.second.p) (void)hipFree(kv.second.p);      // hipFree waits for the device: no launch still uses them
    }
    delete ctx;
}

int cgp_reserve_workspace(cgp_ctx* ctx, size_t bytes, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (bytes == 0) return CGP_OK;
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    std::lock_guard<std::recursive_mutex> launches(ctx->launch_mutex);
    if (!ctx_workspace(ctx, (hipStream_t)stream, bytes, /*reserve=*/true)) return fail(ctx, CGP_E_HIP, "workspace allocation failed");
    return CGP_OK;
}

int cgp_release_workspace(cgp_ctx* ctx, void* stream) {
    if (!ctx) return CGP_E_ARG;
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    std::lock_guard<std::recursive_mutex> launches(ctx->launch_mutex);
    std::lock_guard<std::mutex> lock(ctx->ws_mutex);
    auto it = ctx->ws.find((hipStream_t)stream);
    if (it == ctx->ws.end()) return CGP_OK;
    if (it->second.p) {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, CGP_E_HIP, "hipStreamSynchronize failed (is the stream being captured?)"); }
        (void)hipFree(it->second.p);
    }
    ctx->ws.erase(it);
    return CGP_OK;
}

const char* cgp_source_hash(void) {
    return
#include "cgp_source_hash.inc"
    ;
}

int cgp_debug_set(cgp_ctx* ctx, int key, int64_t value) {
    if (!ctx) return CGP_E_ARG;
    switch (key) {
    case CGP_DBG_WALK_SEGMENTS:
        if (value < 0 || value > 4096) return fail(ctx, CGP_E_ARG, "CGP_DBG_WALK_SEGMENTS outside 0..4096");
        ctx->walk_segments = (int)value;
        return CGP_OK;
    case CGP_DBG_LANE_BUFFERS:
        if (value != 0 && value != 2 && value != 3) return fail(ctx, CGP_E_ARG, "CGP_DBG_LANE_BUFFERS: 0 (default), 2 or 3");
        ctx->lane_buffers = (int)value;
        return CGP_OK;
    case CGP_DBG_COUNT_REGIMES: {
        if (value != 0 && !ctx->counters_mem) {
            DeviceScope on_device(ctx->device);
            if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
            if (hipMalloc(&ctx->counters_mem, 8 * sizeof(unsigned long long)) != hipSuccess) return fail(ctx, CGP_E_HIP, "hipMalloc of the counters failed");
            if (hipMemset(ctx->counters_mem, 0, 8 * sizeof(unsigned long long)) != hipSuccess) return fail(ctx, CGP_E_HIP, "hipMemset of the counters failed");
        }
        ctx->counters = value != 0 ? ctx->counters_mem : nullptr;
        return CGP_OK;
    }
    default: return fail(ctx, CGP_E_ARG, "unknown cgp_debug_set key");
    }
}

int cgp_debug_counters(cgp_ctx* ctx, uint64_t* out, int reset, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (!out) return fail(ctx, CGP_E_ARG, "out is NULL");
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!ctx->counters_mem) return CGP_OK;
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(out, ctx->counters_mem, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess) return fail(ctx, CGP_E_HIP, "copy of the counters failed");
    if (reset && hipMemsetAsync(ctx->counters_mem, 0, 8 * sizeof(uint64_t), st) != hipSuccess) return fail(ctx, CGP_E_HIP, "reset of the counters failed");
    if (hipStreamSynchronize(st) != hipSuccess) return fail(ctx, CGP_E_HIP, "hipStreamSynchronize failed");
    return CGP_OK;
}

const char* cgp_last_error(const cgp_ctx* ctx) {
    if (!ctx) return "null context";
    const ThreadError& e = thread_error();
    return e.ctx == ctx ? e.msg.c_str() : "";
}

static int filter_impl(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, const cgp_init* init,
                       double dt, const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index,
                       int64_t B, int64_t T, double* mfs, double* Pfs, double* nll, uint32_t flags, void* stream,
                       int64_t segments, int64_t burn_in, double* junction_err) {
    if (!ctx) return CGP_E_ARG;
    if (B < 0 || T < 0) return fail(ctx, CGP_E_ARG, "negative B or T");
    if (B == 0 || T == 0) return CGP_OK;
    if (!ys) return fail(ctx, CGP_E_ARG, "ys is NULL");
    if (ys_stride < 0) return fail(ctx, CGP_E_ARG, "ys_stride must be >= 0 (T for dense [B][T] records, 0 for one shared record)");
    if (ys_repeat < 1) return fail(ctx, CGP_E_ARG, "ys_repeat must be >= 1");
    if (!init || !init->Xi || !init->m0 || !init->P0) return fail(ctx, CGP_E_ARG, "init.Xi / m0 / P0 must be set");
    if (method != CGP_F_EKF_KPT && !init->H) return fail(ctx, CGP_E_ARG, "init.H must be set");
    const bool sde = method == CGP_F_CD_EKF || method == CGP_F_CD_SGP;
    const bool sig = method == CGP_F_SGP || method == CGP_F_CD_SGP;
    if (method < CGP_F_EKF || method > CGP_F_EKF_KPT) return fail(ctx, CGP_E_ARG, "unknown filter method");
    int rc = check_model(ctx, model, sde, sig, sigma);
    if (rc != CGP_OK) return rc;
    if ((method == CGP_F_EKF_KPT) != (model->model_id == CGP_M_KPT)) return fail(ctx, CGP_E_ARG, "CGP_F_EKF_KPT goes with CGP_M_KPT only");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    // One call's launches are enqueued as a unit: the per-stream scratch of the time-split forms (memset, kernel, fix-up) is protected by
    // stream order only if two host threads that share this context AND a stream cannot interleave their launches.
    std::lock_guard<std::recursive_mutex> launches(ctx->launch_mutex);

    FilterIO io;
    io.H = init->H; io.H_stride = init->H_stride;
    io.Xi = init->Xi; io.Xi_stride = init->Xi_stride;
    io.m0 = init->m0; io.m0_stride = init->m0_stride;
    io.P0 = init->P0; io.P0_stride = init->P0_stride;
    io.ys = ys; io.ys_stride = ys_stride; io.ys_repeat = ys_repeat; io.ys_index = ys_index; io.B = B; io.T = T; io.mfs = mfs; io.Pfs = Pfs; io.nll = nll; io.flags = flags;
    io.counters = ctx->counters;
    const ModelArgs ma = model_args(model, sigma, dt, flags);
    // ---- time-split with burn-in (cgp_filter_time_split): segments of whole 64-step chunks, one wavefront each
    const bool split = segments > 1;
    double* seg_ws = nullptr;
    if (split) {
        if (burn_in < 0) return fail(ctx, CGP_E_ARG, "burn_in must be >= 0");
        if (!junction_err) return fail(ctx, CGP_E_ARG, "junction_err must be set: a time-split filter is only as good as its junctions");
        // the kernels that know segments, under the conditions the dispatch below sends a launch to them
        const bool lcd4 = model->n_harm == 1 && (model->model_id == CGP_M_HARMONIC_LCD || model->model_id == CGP_M_LASCALA_LCD);
        bool known = false;
        if (!(flags & (CGP_GENERIC_KERNEL | CGP_DPP_KERNEL | CGP_THREAD_PER_TRIAL | CGP_FOUR_TRIALS_PER_WAVE))) {
            if (method == CGP_F_SGP && lcd4) known = sigma_mfma_taken(flags, T, ma);
            else if (method == CGP_F_SGP && model->model_id == CGP_M_HARMONIC_LCD) known = coop8_filter_sgp_ok(model->n_harm, T, ma);
            else if (method == CGP_F_CD_SGP && model->model_id == CGP_M_HARMONIC_SDE && model->n_harm == 1) known = sigma_mfma_taken(flags, T, ma);
            else if (method == CGP_F_EKF && lcd4) known = T * 128 <= 0x7FFFFF00LL;
        }
        if (!known)
            return fail(ctx, CGP_E_UNSUPPORTED, "time-split filters are built for ekf / sgp_filter / cd_sgp_filter on the d = 4 chirp and La Scala models "
                                                "and sgp_filter on the 2- / 3-harmonic model (matrix-core and tile-layout kernels, standard sigma sets)");
        int64_t seg_len = ((T + segments - 1) / segments + 63) / 64 * 64;
        const int64_t segs = (T + seg_len - 1) / seg_len;                  // without the empty ones
        io.segs = (int)segs; io.seg_len = seg_len; io.burn_in = (burn_in + 63) / 64 * 64;
        io.seg_stride = 2 * (model->d + model->d * model->d) + 1;
        if (segs > 1) {
            // (a set too large for the LDS stage runs one lane per trial whatever the flags say -- choose_wave -- and those kernels
            // know no segments: refuse instead of reading records nobody wrote)
            if (sig && SigmaSet::stage_bytes(sigma->s, sigma->d, sigma->n_groups, sigma->group_start != nullptr) > (size_t)kSigLdsMaxBytes)
                return fail(ctx, CGP_E_UNSUPPORTED, "time-split filters need a sigma-point set that fits the LDS stage of the one-wavefront-per-trial kernels");
            const size_t seg_doubles = (size_t)io.seg_stride * (size_t)B * (size_t)segs;
            seg_ws = (double*)ctx_workspace(ctx, (hipStream_t)stream, sizeof(double) * seg_doubles);
            if (!seg_ws) return fail(ctx, CGP_E_HIP, "no workspace for the segment records (allocation failed, or the buffer would grow inside a graph capture: cgp_reserve_workspace first)");
            // NaN-filled: a kernel that ignored the segments would show as junction_err = inf, not as garbage
            if (hipMemsetAsync(seg_ws, 0xFF, sizeof(double) * seg_doubles, (hipStream_t)stream) != hipSuccess) return fail(ctx, CGP_E_HIP, "hipMemsetAsync of the segment records failed");
            io.seg_state = seg_ws;
            flags |= CGP_WAVE_PER_TRIAL;
            io.flags = flags;
        } else io.segs = 1;
    }
    // which lane-cooperative kernel (if any) a wave-per-trial launch of this call would take decides the crossover
    const bool spec = !(flags & CGP_GENERIC_KERNEL);
    const bool chirp4 = spec && model->n_harm == 1 && (model->model_id == CGP_M_HARMONIC_LCD || model->model_id == CGP_M_LASCALA_LCD);
    const bool harm8 = spec && model->model_id == CGP_M_HARMONIC_LCD && (model->n_harm == 2 || model->n_harm == 3);
    const bool sde4 = spec && model->model_id == CGP_M_HARMONIC_SDE && model->n_harm == 1;
    const bool mfma = !(flags & CGP_DPP_KERNEL);
    ShapeLimit limit = sig ? ShapeLimit{8, 1} : ShapeLimit{5, 2};
    // (round 5: where the large-batch lane kernel of cgp_lane4.hpp takes the launch, one lane per trial wins from 9 / 9 trials per
    // SIMD on -- tools/lane_crossover.sh, profiles/r05_lane_crossover.txt: EKF 0.49 against 0.59 ms at 8192 x 500 and 0.71 against 0.59 at
    // 10 240 (CRLB records, i.e. with the four-trials-per-wavefront kernel on its branch-free wide step), GH-3 4.3 against 5.0 ms at
    // 8192 x 500 and 6.5 against 5.0 at 12 288 (after the lane kernel's fan stopped running twice on records outside the lean regime); the generic lane kernel it replaces there kept the round-3 limits of 20 / 24)
    const bool lane4 = spec && lane4_filter_fits(io);
    if (method == CGP_F_EKF && chirp4 && mfma && !(flags & CGP_ONE_TRIAL_PER_WAVE)) limit = lane4 ? ShapeLimit{9, 1} : ShapeLimit{20, 1};
    else if (method == CGP_F_EKF && harm8) limit = {8, 1};
    else if (method == CGP_F_SGP && chirp4 && mfma) limit = (lane4 && sigma_lds_bytes(ma, 4) <= (size_t)kLane4SigLdsMaxBytes) ? ShapeLimit{9, 1} : ShapeLimit{24, 1};
    else if (method == CGP_F_SGP && harm8) limit = {11, 1};
    else if (method == CGP_F_CD_EKF && sde4 && mfma) limit = {4, 1};
    else if (method == CGP_F_CD_SGP && sde4 && mfma) limit = {48, 1};
    const bool wave = choose_wave(ctx, B, flags, limit, sig ? sigma : nullptr);
    if (split && io.segs > 1 && !wave) return fail(ctx, CGP_E_UNSUPPORTED, "time-split filters run one wavefront per trial only");
    hipStream_t st = (hipStream_t)stream;
    switch (model->model_id) {
    case CGP_M_LINEAR:
        // kf at d = 4, one wavefront per trial: the matrix-core step of the EKF with the constant Jacobian F
        if (method == CGP_F_EKF && model->d == 4 && wave && !(flags & (CGP_GENERIC_KERNEL | CGP_DPP_KERNEL)) && T * 128 <= 0x7FFFFF00LL) rc = dispatch_filter_kf4_mfma(io, ma, st);
        else rc = dispatch_filter_disc_linear(method, model->d, wave, io, ma, st);
        break;
    case CGP_M_HARMONIC_LCD:
    case CGP_M_LASCALA_LCD:
        // d = 4 EKF, one wavefront per trial: the lane-cooperative kernel (covariance spread over a 16-lane DPP row)
        if (method == CGP_F_EKF && model->n_harm == 1 && wave && !(flags & CGP_GENERIC_KERNEL)) rc = dispatch_filter_coop4(io, ma, st);
        else if (method == CGP_F_SGP && model->n_harm == 1 && wave && !(flags & CGP_GENERIC_KERNEL)) rc = dispatch_filter_coop4_sgp(io, ma, st);
        else if (method == CGP_F_EKF && wave && !(flags & CGP_GENERIC_KERNEL) && model->model_id == CGP_M_HARMONIC_LCD && (model->n_harm == 2 || model->n_harm == 3)
                 && io.T * model->d * model->d * 8 <= 0x7FFFFF00LL)                              // the kernel's output windows (cgp_coop4.hpp:kOobMaxBytes)
            rc = dispatch_filter_coop8_ekf(model->n_harm, io, ma, st);
        else if (method == CGP_F_SGP && wave && !(flags & CGP_GENERIC_KERNEL) && model->model_id == CGP_M_HARMONIC_LCD && coop8_filter_sgp_ok(model->n_harm, io.T, ma))
            rc = dispatch_filter_coop8_sgp(model->n_harm, io, ma, st);
        else if ((method == CGP_F_EKF || (method == CGP_F_SGP && sigma_lds_bytes(ma, 4) <= (size_t)kLane4SigLdsMaxBytes)) && model->n_harm == 1 && !wave &&
                 !(flags & CGP_GENERIC_KERNEL) && lane4_filter_fits(io))
            rc = dispatch_filter_lane4(method, io, ma, st);                                      // large batches: cgp_lane4.hpp
        else rc = dispatch_filter_disc_harm(method, model->n_harm, wave, io, ma, st);
        break;
    case CGP_M_LINEAR_SDE:   rc = dispatch_filter_sde_linear(method, model->d, wave, io, ma, st); break;
    case CGP_M_HARMONIC_SDE:
        if (method == CGP_F_CD_SGP && model->n_harm == 1 && wave && !(flags & CGP_GENERIC_KERNEL)) rc = dispatch_filter_coop4_cdsgp(io, ma, st);
        else if (method == CGP_F_CD_EKF && model->n_harm == 1 && wave && !(flags & CGP_GENERIC_KERNEL)) rc = dispatch_filter_coop4_cdekf(io, ma, st);
        else rc = dispatch_filter_sde_harm(method, model->n_harm, wave, io, ma, st);
        break;
    case CGP_M_KPT:          rc = dispatch_filter_kpt(model->n_harm, wave, io, ma, st); break;
    default: rc = CGP_E_ARG;
    }
    if (split) {
        if (rc == CGP_OK) {
            if (io.segs > 1) hipLaunchKernelGGL(filter_split_fixup_kernel, dim3((unsigned)B), dim3(64), 0, st, io, model->d, junction_err);
            else if (hipMemsetAsync(junction_err, 0, sizeof(double) * (size_t)B, st) != hipSuccess) rc = CGP_E_HIP;
        }
    }
    if (rc == CGP_E_UNSUPPORTED) return fail(ctx, rc, "this (method, model, dimension) combination is not compiled in");
    if (rc == CGP_E_HIP) return fail(ctx, rc, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return rc;
}

int cgp_filter(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, const cgp_init* init,
               double dt, const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index,
               int64_t B, int64_t T, double* mfs, double* Pfs, double* nll, uint32_t flags, void* stream) {
    return filter_impl(ctx, method, model, sigma, init, dt, ys, ys_stride, ys_repeat, ys_index, B, T, mfs, Pfs, nll, flags, stream, 1, 0, nullptr);
}

int cgp_filter_time_split(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, const cgp_init* init,
                          double dt, const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index,
                          int64_t B, int64_t T, double* mfs, double* Pfs, double* nll, uint32_t flags,
                          int64_t segments, int64_t burn_in, double* junction_err, void* stream) {
    if (ctx && segments < 1) return fail(ctx, CGP_E_ARG, "segments must be >= 1");
    if (ctx && segments == 1 && junction_err && B > 0) {                 // nothing is split: no junction, no mismatch
        DeviceScope on_device(ctx->device);
        if (!on_device.ok || hipMemsetAsync(junction_err, 0, sizeof(double) * (size_t)B, (hipStream_t)stream) != hipSuccess) return fail(ctx, CGP_E_HIP, "hipMemsetAsync failed");
    }
    return filter_impl(ctx, method, model, sigma, init, dt, ys, ys_stride, ys_repeat, ys_index, B, T, mfs, Pfs, nll, flags, stream, segments, burn_in, junction_err);
}

static int smoother_impl(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, double dt,
                         const double* mfs, const double* Pfs, int64_t B, int64_t T, const cgp_smooth_out* out,
                         uint32_t flags, void* stream, int64_t segments = 1, int64_t burn_in = 0, double* junction_err = nullptr) {
    if (!ctx) return CGP_E_ARG;
    if (B < 0 || T < 0) return fail(ctx, CGP_E_ARG, "negative B or T");
    if (B == 0 || T == 0) return CGP_OK;
    if (!out) return fail(ctx, CGP_E_ARG, "out is NULL");
    double* const mss = out->mss;
    double* const Pss = out->Pss;
    const bool want_sel = out->comp_mean || out->comp_var || out->expect;
    if (!mfs || !Pfs) return fail(ctx, CGP_E_ARG, "mfs / Pfs must be set");
    if (!want_sel && (!mss || !Pss)) return fail(ctx, CGP_E_ARG, "mfs / Pfs / mss / Pss must be set");
    if (want_sel) {
        if (!model || out->comp < 0 || out->comp >= model->d) return fail(ctx, CGP_E_ARG, "cgp_smooth_out.comp outside 0..d-1");
        if (out->expect && (out->func < CGP_FN_SOFTPLUS || out->func > CGP_FN_SQUARE || out->order < 1 || out->order > kGhMaxOrder || !out->xi || !out->w))
            return fail(ctx, CGP_E_ARG, "cgp_smooth_out.expect needs func, 1 <= order <= 32, xi and w");
    }
    if (method < CGP_S_EKS || method > CGP_S_CD_SGP) return fail(ctx, CGP_E_ARG, "unknown smoother method");
    const bool sde = method == CGP_S_CD_EKS || method == CGP_S_CD_SGP;
    const bool sig = method == CGP_S_SGP || method == CGP_S_CD_SGP;
    int rc = check_model(ctx, model, sde, sig, sigma);
    if (rc != CGP_OK) return rc;
    if (model->model_id == CGP_M_KPT) return fail(ctx, CGP_E_ARG, "the KPT model has no smoother in the reference");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    std::lock_guard<std::recursive_mutex> launches(ctx->launch_mutex);        // (see filter_impl: compose + apply of one call stay together)

    SmootherIO io;
    io.mfs = mfs; io.Pfs = Pfs; io.B = B; io.T = T; io.mss = mss; io.Pss = Pss; io.flags = flags;
    io.host_ctx = ctx;
    io.lane_buffers = ctx->lane_buffers;
    // Time-split form of the discrete wave-per-trial smoothers: when the batch leaves two thirds of the SIMDs idle (or on
    // request).  cgp_debug_set(CGP_DBG_WALK_SEGMENTS) caps the number of segments of this context (tuning aid).
    io.num_cus = ctx->num_cus;
    {
        const long env_segs = ctx->walk_segments;
        const bool forced = (flags & CGP_TIME_SPLIT) != 0;
        // Not by default for the LINEAR model (rts): its F is the caller's, and composing the affine maps of hundreds of steps
        // (C = Pf - G Pp G^T, A <- G A) is only as accurate as those products are well conditioned -- 1e-11 of the whole-record
        // walk for the chirp family's gains (d = 4 ... 8, every parameter set of the tests), but 1e-7 for a random F at d = 6,
        // where partial products of the gains grow in directions the true carry never excites.  The split is exact in exact
        // arithmetic either way; for a linear model it is the caller's choice (CGP_TIME_SPLIT).
        const bool own_model = model->model_id != CGP_M_LINEAR;
        if ((flags & CGP_NO_TIME_SPLIT) || (!forced && (!own_model || 3 * B > (int64_t)ctx->num_cus * 4))) io.segs = 1;      // measured: x3 at B = 125, x1.6 at 250, x0.9 at 500
        else io.segs = env_segs > 1 ? (int)env_segs : (env_segs == 1 ? 1 : 0);
        io.min_tiles = forced ? 1 : 4;
    }
    const ModelArgs ma = model_args(model, sigma, dt, flags);
    const bool affine = (method == CGP_S_EKS || method == CGP_S_SGP) && !(flags & CGP_SEQUENTIAL_SCAN);
    // launch shape (see choose_wave): the cooperative walks of the discrete smoothers are never slower than one lane per trial
    const bool spec = !(flags & CGP_GENERIC_KERNEL);
    const bool sde4 = spec && model->model_id == CGP_M_HARMONIC_SDE && model->n_harm == 1 && !(flags & CGP_DPP_KERNEL);
    bool coop_walk = false;
    if (affine && spec) {
        if (model->model_id == CGP_M_LINEAR) coop_walk = coop8_smoother_ok(model->d, T, ma) || (model->d == 4 && walk4_smoother_fits(T, ma));
        else if (model->model_id == CGP_M_HARMONIC_LCD || model->model_id == CGP_M_LASCALA_LCD)
            coop_walk = (model->n_harm >= 2 && coop8_smoother_ok(model->d, T, ma) && coop8_smoother_harm_ok(method, ma)) || (model->n_harm == 1 && walk4_smoother_fits(T, ma));
    }
    ShapeLimit limit = affine ? ShapeLimit{16, 1} : (sig ? ShapeLimit{8, 1} : ShapeLimit{5, 2});
    if (coop_walk) limit = {-1, 1};
    else if (method == CGP_S_CD_EKS && sde4) limit = lane4_smoother_fits(io) ? ShapeLimit{3, 1} : ShapeLimit{5, 1};
    // (round 5: eks on the d = 4 chirp models beyond 24 trials per SIMD runs one lane per trial in the kernel of cgp_lane4.hpp -- 1.21 against
    // 1.48 ms at 32 768 x 500, 8.0 against 10.6 ms at 262 144 x 500, the walk ahead below: 0.70 against 0.93 ms at 16 384;
    // profiles/r05_lane_smoothers.txt.  cd_eks: 1.74 against 2.48 ms at 4096 x 500)
    if (coop_walk && method == CGP_S_EKS && spec && model->n_harm == 1 && (model->model_id == CGP_M_HARMONIC_LCD || model->model_id == CGP_M_LASCALA_LCD) &&
        lane4_smoother_fits(io) && !(flags & CGP_SEQUENTIAL_SCAN)) limit = {24, 1};
    else if (method == CGP_S_CD_SGP && sde4) limit = {48, 1};
    const bool wave = choose_wave(ctx, B, flags, limit, sig ? sigma : nullptr);
    hipStream_t st = (hipStream_t)stream;
    // which kernel takes the launch -- decided before anything is enqueued, because only some of them write the selected outputs themselves
    enum Route { kCoop8Linear, kWalk4Linear, kDiscLinear, kCoop8Harm, kWalk4Harm, kLane4, kDiscHarm, kSdeLinear, kCoop4CdSgp, kCoop4CdEks, kSdeHarm, kNone };
    Route route = kNone;
    switch (model->model_id) {
    case CGP_M_LINEAR:
        // 5 <= d <= 8, one wavefront per trial: maps built per lane, applied cooperatively in the tile layout (cgp_coop8.hpp)
        if (affine && wave && !(flags & CGP_GENERIC_KERNEL) && coop8_smoother_ok(model->d, T, ma)) route = kCoop8Linear;
        // d = 4: gains per lane, the recursion walked on the matrix cores (cgp_walk4.hpp)
        else if (affine && wave && !(flags & CGP_GENERIC_KERNEL) && model->d == 4 && walk4_smoother_fits(T, ma)) route = kWalk4Linear;
        else route = kDiscLinear;
        break;
    case CGP_M_HARMONIC_LCD:
    case CGP_M_LASCALA_LCD:
        if (affine && wave && !(flags & CGP_GENERIC_KERNEL) && model->n_harm >= 2 && coop8_smoother_ok(model->d, T, ma) && coop8_smoother_harm_ok(method, ma)) route = kCoop8Harm;
        else if (affine && wave && !(flags & CGP_GENERIC_KERNEL) && model->n_harm == 1 && walk4_smoother_fits(T, ma)) route = kWalk4Harm;
        else if (method == CGP_S_EKS && model->n_harm == 1 && !wave && !(flags & CGP_GENERIC_KERNEL) && lane4_smoother_fits(io)) route = kLane4;      // one lane per trial: cgp_lane4.hpp
        else route = kDiscHarm;
        break;
    case CGP_M_LINEAR_SDE:   route = kSdeLinear; break;
    case CGP_M_HARMONIC_SDE:
        if (method == CGP_S_CD_SGP && model->n_harm == 1 && wave && !(flags & CGP_GENERIC_KERNEL)) route = kCoop4CdSgp;
        else if (method == CGP_S_CD_EKS && model->n_harm == 1 && wave && !(flags & CGP_GENERIC_KERNEL)) route = kCoop4CdEks;
        else if (method == CGP_S_CD_EKS && model->n_harm == 1 && !wave && !(flags & CGP_GENERIC_KERNEL) && lane4_smoother_fits(io)) route = kLane4;   // one lane per trial: cgp_lane4.hpp
        else route = kSdeHarm;
        break;
    default: break;
    }
    // selected outputs (cgp_smoother_select): written by the kernel itself where it is one of the d = 4 walks / lane kernels or the tile-layout
    // kernels (mss / Pss then optional); any other kernel writes the full rows and a gather launch reads the marginal back out of them
    const bool sel_native = route == kWalk4Linear || route == kWalk4Harm || route == kLane4 || route == kCoop8Linear || route == kCoop8Harm;
    if (want_sel) {
        if (sel_native) {
            io.sel.comp = out->comp; io.sel.func = out->func; io.sel.order = out->order;
            io.sel.mean = out->comp_mean; io.sel.var = out->comp_var; io.sel.expect = out->expect;
            io.sel.xi = out->xi; io.sel.w = out->w;
        } else if (!mss || !Pss)
            return fail(ctx, CGP_E_UNSUPPORTED, "this (method, model, launch shape) writes selected outputs from its full rows only: pass mss and Pss as well");
    }
    // ---- time-split with burn-in (cgp_smoother_time_split): the continuous-discrete sigma-point smoother on the d = 4 matrix-core kernel
    if (segments > 1) {
        if (burn_in < 0) return fail(ctx, CGP_E_ARG, "burn_in must be >= 0");
        if (!junction_err) return fail(ctx, CGP_E_ARG, "junction_err must be set: a time-split smoother is only as good as its junctions");
        if (want_sel) return fail(ctx, CGP_E_UNSUPPORTED, "cgp_smoother_time_split writes full rows only");
        const bool split_sgp = route == kCoop4CdSgp && !(flags & CGP_DPP_KERNEL) && sigma_mfma_taken(flags, T, ma);
        const bool split_eks = route == kCoop4CdEks && !(flags & CGP_DPP_KERNEL) && T * 128 <= 0x7FFFFF00LL;        // cgp_inst_coop4.hip: the matrix-core kernel
        if (!split_sgp && !split_eks)
            return fail(ctx, CGP_E_UNSUPPORTED, "time-split smoothers with burn-in are built for cd_eks and cd_sgp_smoother on the d = 4 chirp / La Scala SDE (matrix-core "
                                                "kernels; standard sigma set); the discrete smoothers split exactly (CGP_TIME_SPLIT)");
        const int64_t chunks = (T - 1 + 63) / 64;
        if (chunks >= 2) {
            const int64_t cps = (chunks + segments - 1) / segments;
            const int64_t segs = (chunks + cps - 1) / cps;
            if (segs > 1) {
                io.bsegs = (int)segs; io.chunks_per_bseg = (int)cps; io.burn_chunks = (int)((burn_in + 63) / 64);
                io.junction = (double*)ctx_workspace(ctx, st, sizeof(double) * 20 * (size_t)B * (size_t)segs);
                if (!io.junction) return fail(ctx, CGP_E_HIP, "no workspace for the junction states (allocation failed, or the buffer would grow inside a graph capture: cgp_reserve_workspace first)");
            }
        }
    }
    switch (route) {
    case kCoop8Linear: rc = dispatch_smoother_coop8_linear(method, model->d, io, ma, st); break;
    case kWalk4Linear: rc = dispatch_smoother_walk4_linear(method, io, ma, st); break;
    case kDiscLinear:  rc = dispatch_smoother_disc_linear(method, model->d, wave, io, ma, st); break;
    case kCoop8Harm:   rc = dispatch_smoother_coop8_harm(method, model->n_harm, io, ma, st); break;
    case kWalk4Harm:   rc = dispatch_smoother_walk4_harm(method, io, ma, st); break;
    case kLane4:       rc = dispatch_smoother_lane4(method, model->model_id, io, ma, st); break;
    case kDiscHarm:    rc = dispatch_smoother_disc_harm(method, model->n_harm, wave, io, ma, st); break;
    case kSdeLinear:   rc = dispatch_smoother_sde_linear(method, model->d, wave, io, ma, st); break;
    case kCoop4CdSgp:  rc = dispatch_smoother_coop4_cdsgp(io, ma, st); break;
    case kCoop4CdEks:  rc = dispatch_smoother_coop4_cdeks(io, ma, st); break;
    case kSdeHarm:     rc = dispatch_smoother_sde_harm(method, model->n_harm, wave, io, ma, st); break;
    default: rc = CGP_E_ARG;
    }
    if (rc == CGP_OK && junction_err) {
        if (io.bsegs > 1) rc = dispatch_smoother_split_fixup(io, junction_err, st);
        else if (hipMemsetAsync(junction_err, 0, sizeof(double) * (size_t)B, st) != hipSuccess) rc = CGP_E_HIP;      // nothing was split: no junction, no mismatch
    }
    if (rc == CGP_OK && want_sel && !sel_native) {
        SmoothSel sel;
        sel.comp = out->comp; sel.func = out->func; sel.order = out->order;
        sel.mean = out->comp_mean; sel.var = out->comp_var; sel.expect = out->expect; sel.xi = out->xi; sel.w = out->w;
        const int64_t n = B * T, blocks = (n + 255) / 256;
        hipLaunchKernelGGL(smooth_select_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, mss, Pss, n, model->d, sel);
        if (hipGetLastError() != hipSuccess) rc = CGP_E_HIP;
    }
    if (rc == CGP_E_UNSUPPORTED) return fail(ctx, rc, "this (method, model, dimension) combination is not compiled in");
    if (rc == CGP_E_HIP) return fail(ctx, rc, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return rc;
}

int cgp_smoother(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, double dt,
                 const double* mfs, const double* Pfs, int64_t B, int64_t T, double* mss, double* Pss,
                 uint32_t flags, void* stream) {
    cgp_smooth_out out;
    memset(&out, 0, sizeof(out));
    out.mss = mss; out.Pss = Pss; out.comp = -1;
    return smoother_impl(ctx, method, model, sigma, dt, mfs, Pfs, B, T, &out, flags, stream);
}

int cgp_smoother_time_split(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, double dt,
                            const double* mfs, const double* Pfs, int64_t B, int64_t T, double* mss, double* Pss,
                            uint32_t flags, int64_t segments, int64_t burn_in, double* junction_err, void* stream) {
    if (ctx && segments < 1) return fail(ctx, CGP_E_ARG, "segments must be >= 1");
    if (ctx && !junction_err) return fail(ctx, CGP_E_ARG, "junction_err must be set");
    cgp_smooth_out out;
    memset(&out, 0, sizeof(out));
    out.mss = mss; out.Pss = Pss; out.comp = -1;
    return smoother_impl(ctx, method, model, sigma, dt, mfs, Pfs, B, T, &out, flags, stream, segments, burn_in, junction_err);
}

int cgp_smoother_select(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, double dt,
                        const double* mfs, const double* Pfs, int64_t B, int64_t T, const cgp_smooth_out* out,
                        uint32_t flags, void* stream) {
    return smoother_impl(ctx, method, model, sigma, dt, mfs, Pfs, B, T, out, flags, stream);
}

int cgp_gaussian_expectation(cgp_ctx* ctx, const double* ms, const double* sd, int64_t n, int64_t in_stride,
                             const double* xi, const double* w, int32_t order, double* out, void* stream) {
    return cgp_gaussian_expectation_fn(ctx, CGP_FN_SOFTPLUS, ms, sd, n, in_stride, xi, w, order, out, stream);
}

int cgp_gaussian_expectation_fn(cgp_ctx* ctx, int func, const double* ms, const double* sd, int64_t n, int64_t in_stride,
                                const double* xi, const double* w, int32_t order, double* out, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (func < CGP_FN_SOFTPLUS || func > CGP_FN_SQUARE) return fail(ctx, CGP_E_ARG, "unknown integrand");
    if (n < 0 || order < 1) return fail(ctx, CGP_E_ARG, "bad n or order");
    if (n == 0) return CGP_OK;
    if (!ms || !sd || !xi || !w || !out) return fail(ctx, CGP_E_ARG, "NULL pointer");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    const int64_t blocks = (n + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    hipStream_t st = (hipStream_t)stream;
    switch (func) {
    case CGP_FN_EXP:      hipLaunchKernelGGL(gaussian_expectation_kernel<CGP_FN_EXP>, dim3(grid), dim3(256), 0, st, ms, sd, n, in_stride, xi, w, order, out); break;
    case CGP_FN_IDENTITY: hipLaunchKernelGGL(gaussian_expectation_kernel<CGP_FN_IDENTITY>, dim3(grid), dim3(256), 0, st, ms, sd, n, in_stride, xi, w, order, out); break;
    case CGP_FN_SQUARE:   hipLaunchKernelGGL(gaussian_expectation_kernel<CGP_FN_SQUARE>, dim3(grid), dim3(256), 0, st, ms, sd, n, in_stride, xi, w, order, out); break;
    default:              hipLaunchKernelGGL(gaussian_expectation_kernel<CGP_FN_SOFTPLUS>, dim3(grid), dim3(256), 0, st, ms, sd, n, in_stride, xi, w, order, out); break;
    }
    return hipGetLastError() == hipSuccess ? CGP_OK : fail(ctx, CGP_E_HIP, "kernel launch failed");
}

int cgp_squared_error_sums(cgp_ctx* ctx, const double* a, const double* r, int64_t B, int64_t T, int32_t d,
                           const int32_t* comps, int32_t n, double* sums, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (B < 0 || T < 0 || d < 1 || n < 1 || n > 8) return fail(ctx, CGP_E_ARG, "bad B, T, d or n (1 <= n <= 8)");
    if (B == 0 || T == 0) return CGP_OK;
    if (!a || !r || !comps || !sums) return fail(ctx, CGP_E_ARG, "NULL pointer");
    ErrComps ec;
    ec.n = n;
    for (int i = 0; i < 8; i++) {
        ec.c[i] = i < n ? comps[i] : 0;
        if (ec.c[i] < 0 || ec.c[i] >= d) return fail(ctx, CGP_E_ARG, "component outside 0..d-1");
    }
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    const int64_t slabs = (B + 511) / 512;
    if (slabs > 65535) return fail(ctx, CGP_E_ARG, "more than 65535 x 512 trials in one call: reduce chunk by chunk");
    hipLaunchKernelGGL(squared_error_sums_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)slabs), dim3(256), 0, (hipStream_t)stream, a, r, B, T, (int)d, ec, sums);
    return hipGetLastError() == hipSuccess ? CGP_OK : fail(ctx, CGP_E_HIP, "kernel launch failed");
}

int cgp_debug_math(cgp_ctx* ctx, int op, const double* x, int64_t n, double* out0, double* out1, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (n < 0 || op < 0 || op > 11) return fail(ctx, CGP_E_ARG, "bad op or n");
    if (n == 0) return CGP_OK;
    if (!x || !out0) return fail(ctx, CGP_E_ARG, "NULL pointer");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    const int64_t blocks = (n + 255) / 256;
    if (op == 5 || op == 6 || op == 8) hipLaunchKernelGGL(debug_math_uniform_kernel, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(64), 0, (hipStream_t)stream, op, x, n, out0, out1);
    else hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, op, x, n, out0, out1);
    return hipGetLastError() == hipSuccess ? CGP_OK : fail(ctx, CGP_E_HIP, "kernel launch failed");
}

}  // extern "C"
