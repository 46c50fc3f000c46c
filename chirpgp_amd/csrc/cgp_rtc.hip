// cgp_rtc.hip -- models compiled at run time (hiprtc) into the generic lane-per-trial kernels: cgp_model_from_source, cgp_filter_custom,
// cgp_smoother_custom.  See cgp_custom.hpp for the model interface.  libhiprtc is opened on first use (dlopen), so a process that never
// builds a custom model never loads it.
#include <dlfcn.h>
#include <cstring>
#include <string>
#include <vector>
#include "cgp_kernels.hpp"
#include "cgp_ctx.hpp"

struct cgp_custom_model {
    int kind = 0, d = 0;
    int device = 0;
    hipModule_t module = nullptr;
    hipFunction_t filter = nullptr, smoother = nullptr, sgp_filter = nullptr, sgp_smoother = nullptr;
};

namespace {
using namespace cgp;

struct Rtc {
    void* lib = nullptr;
    int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*AddNameExpression)(void*, const char*) = nullptr;
    int (*CompileProgram)(void*, int, const char**) = nullptr;
    int (*GetProgramLogSize)(void*, size_t*) = nullptr;
    int (*GetProgramLog)(void*, char*) = nullptr;
    int (*GetLoweredName)(void*, const char*, const char**) = nullptr;
    int (*GetCodeSize)(void*, size_t*) = nullptr;
    int (*GetCode)(void*, char*) = nullptr;
    int (*DestroyProgram)(void**) = nullptr;
    bool ok = false;
};
Rtc& rtc() {
    static Rtc r = [] {
        Rtc x;
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        auto sym = [&](const char* n) { return dlsym(x.lib, n); };
        x.CreateProgram = (decltype(x.CreateProgram))sym("hiprtcCreateProgram");
        x.AddNameExpression = (decltype(x.AddNameExpression))sym("hiprtcAddNameExpression");
        x.CompileProgram = (decltype(x.CompileProgram))sym("hiprtcCompileProgram");
        x.GetProgramLogSize = (decltype(x.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
        x.GetProgramLog = (decltype(x.GetProgramLog))sym("hiprtcGetProgramLog");
        x.GetLoweredName = (decltype(x.GetLoweredName))sym("hiprtcGetLoweredName");
        x.GetCodeSize = (decltype(x.GetCodeSize))sym("hiprtcGetCodeSize");
        x.GetCode = (decltype(x.GetCode))sym("hiprtcGetCode");
        x.DestroyProgram = (decltype(x.DestroyProgram))sym("hiprtcDestroyProgram");
        x.ok = x.CreateProgram && x.AddNameExpression && x.CompileProgram && x.GetProgramLogSize && x.GetProgramLog && x.GetLoweredName &&
               x.GetCodeSize && x.GetCode && x.DestroyProgram;
        return x;
    }();
    return r;
}

// (the sigma-point set as a plain point list read from global memory: no groups, no LDS stage, no closed forms)
ModelArgs custom_args(const double* params, int64_t param_stride, const double* gamma, int64_t gamma_stride, double dt, const cgp_sigma* sg) {
    ModelArgs ma;
    ma.params = params; ma.param_stride = param_stride; ma.gamma = gamma; ma.gamma_stride = gamma_stride;
    ma.model_id = -1; ma.dt = dt;
    ma.sg.xi = sg ? sg->xi : nullptr; ma.sg.w = sg ? sg->w : nullptr; ma.sg.s = sg ? sg->s : 0; ma.sg.group_start = nullptr; ma.sg.n_groups = 0;
    ma.sg.lds_xi = 0; ma.sg.lds_w = 0; ma.sg.lds_gs = 0; ma.sg.lds_tab = 0; ma.sg.flags = 0u;
    return ma;
}
}  // namespace

extern "C" {

int cgp_model_from_source(cgp_ctx* ctx, int kind, int32_t d, const char* body, const char* include_dir, cgp_custom_model** out) {
    if (!ctx) return CGP_E_ARG;
    if (!out) return fail(ctx, CGP_E_ARG, "out is NULL");
    *out = nullptr;
    if (kind != CGP_CUSTOM_DISCRETE && kind != CGP_CUSTOM_SDE && kind != CGP_CUSTOM_MEASUREMENT)
        return fail(ctx, CGP_E_ARG, "kind must be CGP_CUSTOM_DISCRETE, CGP_CUSTOM_SDE or CGP_CUSTOM_MEASUREMENT");
    if (d < 1 || d > 8) return fail(ctx, CGP_E_UNSUPPORTED, "custom models: state dimension 1..8");
    if (!body || !include_dir) return fail(ctx, CGP_E_ARG, "body / include_dir is NULL");
    Rtc& R = rtc();
    if (!R.ok) return fail(ctx, CGP_E_UNSUPPORTED, "libhiprtc.so could not be loaded: custom models need the runtime compiler of ROCm");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return fail(ctx, CGP_E_HIP, "hipGetDeviceProperties failed");

    const std::string D = std::to_string(d);
    std::string src = "#include \"cgp_custom.hpp\"\nnamespace cgp_user {\nusing namespace cgp::ad;\n#line 1 \"model\"\n";
    src += body;
    // (the kernels take FilterIO / SmootherIO / ModelArgs BY VALUE: the headers under include_dir must be the ones this library was built from --
    // their sizes travel with the code object and are compared after loading)
    src += "\n}\nextern \"C\" __device__ const unsigned cgp_rtc_abi[3] = {(unsigned)sizeof(cgp::FilterIO), (unsigned)sizeof(cgp::SmootherIO), (unsigned)sizeof(cgp::ModelArgs)};\n"
           "struct CgpUserModel {\n";
    std::string f_name, s_name, sf_name, ss_name;
    if (kind == CGP_CUSTOM_DISCRETE) {
        src += "    template <class T> __device__ static void mean(const T* u, const double* p, double dt, T* m) { cgp_user::cond_mean(u, p, dt, m); }\n"
               "    __device__ static void cov(const double* u, const double* p, double dt, double* c) { cgp_user::cond_cov(u, p, dt, c); }\n};\n"
               "using CgpUserM = cgp::CustomDisc<" + D + ", CgpUserModel>;\n";
        f_name = "cgp::filter_kernel<cgp::EkfPredict<CgpUserM, false>, cgp::LinearMeasurement<" + D + ">>";
        s_name = "cgp::smoother_kernel<cgp::EksStep<CgpUserM, false>>";
        sf_name = "cgp::filter_kernel<cgp::SgpPredictCustom<CgpUserM>, cgp::LinearMeasurement<" + D + ">>";
        ss_name = "cgp::smoother_kernel<cgp::SgpsStepCustom<CgpUserM>>";
    } else if (kind == CGP_CUSTOM_MEASUREMENT) {
        // ekf_for_kpt: linear dynamics (F, Sigma as the parameter vector), the caller's h -- a filter, nothing else
        src += "    template <class T> __device__ static T measure(const T* u, const double* q) { return cgp_user::measure(u, q); }\n};\n";
        f_name = "cgp::filter_kernel<cgp::EkfPredict<cgp::LinearDisc<" + D + ">, false>, cgp::CustomMeasurement<" + D + ", CgpUserModel>>";
    } else {
        src += "    template <class T> __device__ static void drift(const T* u, const double* p, T* a) { cgp_user::drift(u, p, a); }\n};\n"
               "using CgpUserM = cgp::CustomSDE<" + D + ", CgpUserModel>;\n";
        f_name = "cgp::filter_kernel<cgp::CdEkfPredict<CgpUserM, false>, cgp::LinearMeasurement<" + D + ">>";
        s_name = "cgp::smoother_kernel<cgp::CdEksStep<CgpUserM, false>>";
        sf_name = "cgp::filter_kernel<cgp::CdSgpPredict<CgpUserM, false>, cgp::LinearMeasurement<" + D + ">>";
        ss_name = "cgp::smoother_kernel<cgp::CdSgpsStep<CgpUserM, false>>";
    }
    void* prog = nullptr;
    if (R.CreateProgram(&prog, src.c_str(), "cgp_custom_model.hip", 0, nullptr, nullptr) != 0) return fail(ctx, CGP_E_HIP, "hiprtcCreateProgram failed");
    auto done = [&](int code, const std::string& msg) { R.DestroyProgram(&prog); return fail(ctx, code, msg); };
    const std::string* names[4] = {&f_name, &s_name, &sf_name, &ss_name};
    for (const std::string* n : names)
        if (!n->empty() && R.AddNameExpression(prog, n->c_str()) != 0) return done(CGP_E_HIP, "hiprtcAddNameExpression failed");
    const std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
    const std::string inc1 = std::string("-I") + include_dir, inc2 = std::string("-I") + include_dir + "/../../include";
    const char* opts[] = {arch.c_str(), "-std=c++17", "-O3", "-fno-fast-math", inc1.c_str(), inc2.c_str()};
    const int rc = R.CompileProgram(prog, 6, opts);
    if (rc != 0) {
        size_t n = 0;
        std::string log;
        if (R.GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); R.GetProgramLog(prog, &log[0]); }
        if (log.size() > 6000) log.resize(6000);
        return done(CGP_E_ARG, "the model source does not compile:\n" + log);
    }
    const char* low[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t size = 0;
    for (int i = 0; i < 4; i++)
        if (!names[i]->empty() && R.GetLoweredName(prog, names[i]->c_str(), &low[i]) != 0) return done(CGP_E_HIP, "hiprtc: no lowered name");
    if (R.GetCodeSize(prog, &size) != 0) return done(CGP_E_HIP, "hiprtc: no code");
    std::vector<char> code(size);
    if (R.GetCode(prog, code.data()) != 0) return done(CGP_E_HIP, "hiprtcGetCode failed");
    cgp_custom_model* m = new cgp_custom_model;
    m->kind = kind; m->d = d; m->device = ctx->device;
    hipFunction_t* fns[4] = {&m->filter, &m->smoother, &m->sgp_filter, &m->sgp_smoother};
    bool loaded = hipModuleLoadData(&m->module, code.data()) == hipSuccess;
    for (int i = 0; loaded && i < 4; i++)
        if (low[i]) loaded = hipModuleGetFunction(fns[i], m->module, low[i]) == hipSuccess;
    if (!loaded) {
        const std::string why = hipGetErrorString(hipGetLastError());
        if (m->module) (void)hipModuleUnload(m->module);
        delete m;
        return done(CGP_E_HIP, "loading the compiled model failed: " + why);
    }
    {
        hipDeviceptr_t abi_ptr = nullptr;
        size_t abi_bytes = 0;
        unsigned abi[3] = {0, 0, 0};
        const bool read = hipModuleGetGlobal(&abi_ptr, &abi_bytes, m->module, "cgp_rtc_abi") == hipSuccess && abi_bytes == sizeof(abi) &&
                          hipMemcpy(abi, abi_ptr, sizeof(abi), hipMemcpyDeviceToHost) == hipSuccess;
        if (!read || abi[0] != sizeof(FilterIO) || abi[1] != sizeof(SmootherIO) || abi[2] != sizeof(ModelArgs)) {
            (void)hipModuleUnload(m->module);
            delete m;
            return done(CGP_E_ARG, "the headers under include_dir are not the ones this library was built from (argument structs differ in size): "
                                   "point include_dir at the csrc/ of this build");
        }
    }
    R.DestroyProgram(&prog);
    *out = m;
    return CGP_OK;
}

void cgp_custom_model_destroy(cgp_custom_model* m) {
    if (!m) return;
    {
        DeviceScope on_device(m->device);
        if (m->module) (void)hipModuleUnload(m->module);
    }
    delete m;
}

int cgp_filter_custom(cgp_ctx* ctx, const cgp_custom_model* m, const cgp_sigma* sigma, const double* params, int64_t param_stride, const double* gamma, int64_t gamma_stride,
                      const cgp_init* init, double dt, const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index,
                      int64_t B, int64_t T, double* mfs, double* Pfs, double* nll, uint32_t flags, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (!m) return fail(ctx, CGP_E_ARG, "model is NULL");
    if (B < 0 || T < 0) return fail(ctx, CGP_E_ARG, "negative B or T");
    if (B == 0 || T == 0) return CGP_OK;
    if (m->device != ctx->device) return fail(ctx, CGP_E_ARG, "the model was compiled for another device's context");
    if (!ys || !params) return fail(ctx, CGP_E_ARG, "ys / params is NULL");
    if (ys_stride < 0 || ys_repeat < 1) return fail(ctx, CGP_E_ARG, "ys_stride must be >= 0 and ys_repeat >= 1");
    if (!init || !init->H || !init->Xi || !init->m0 || !init->P0) return fail(ctx, CGP_E_ARG, "init.H / Xi / m0 / P0 must be set");
    if (m->kind == CGP_CUSTOM_SDE && !gamma) return fail(ctx, CGP_E_ARG, "SDE models need gamma = b b^T");
    if (!(sigma ? m->sgp_filter : m->filter)) return fail(ctx, CGP_E_UNSUPPORTED, "a measurement function (ekf_for_kpt) has no sigma-point form");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    std::lock_guard<std::recursive_mutex> launches(ctx->launch_mutex);
    FilterIO io;
    io.H = init->H; io.H_stride = init->H_stride; io.Xi = init->Xi; io.Xi_stride = init->Xi_stride;
    io.m0 = init->m0; io.m0_stride = init->m0_stride; io.P0 = init->P0; io.P0_stride = init->P0_stride;
    io.ys = ys; io.ys_stride = ys_stride; io.ys_repeat = ys_repeat; io.ys_index = ys_index; io.B = B; io.T = T;
    io.mfs = mfs; io.Pfs = Pfs; io.nll = nll; io.flags = flags & CGP_NLL_FINAL_ONLY;
    if (sigma && (!sigma->xi || !sigma->w || sigma->s < 1 || sigma->d != m->d)) return fail(ctx, CGP_E_ARG, "cgp_sigma needs xi, w, s >= 1 and d = the model's");
    ModelArgs ma = custom_args(params, param_stride, gamma, gamma_stride, dt, sigma);
    void* args[] = {&io, &ma};
    if (hipModuleLaunchKernel(sigma ? m->sgp_filter : m->filter, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess)
        return fail(ctx, CGP_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return CGP_OK;
}

int cgp_smoother_custom(cgp_ctx* ctx, const cgp_custom_model* m, const cgp_sigma* sigma, const double* params, int64_t param_stride, const double* gamma, int64_t gamma_stride,
                        double dt, const double* mfs, const double* Pfs, int64_t B, int64_t T, double* mss, double* Pss, uint32_t flags, void* stream) {
    if (!ctx) return CGP_E_ARG;
    if (!m) return fail(ctx, CGP_E_ARG, "model is NULL");
    if (B < 0 || T < 0) return fail(ctx, CGP_E_ARG, "negative B or T");
    if (B == 0 || T == 0) return CGP_OK;
    if (m->device != ctx->device) return fail(ctx, CGP_E_ARG, "the model was compiled for another device's context");
    if (!mfs || !Pfs || !mss || !Pss || !params) return fail(ctx, CGP_E_ARG, "mfs / Pfs / mss / Pss / params must be set");
    if (m->kind == CGP_CUSTOM_SDE && !gamma) return fail(ctx, CGP_E_ARG, "SDE models need gamma = b b^T");
    if (!(sigma ? m->sgp_smoother : m->smoother))
        return fail(ctx, CGP_E_UNSUPPORTED, "a measurement function (ekf_for_kpt) has no smoother of its own: its dynamics are linear -- cgp_smoother with CGP_S_EKS on (F, Sigma)");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(ctx, CGP_E_HIP, "hipSetDevice failed");
    std::lock_guard<std::recursive_mutex> launches(ctx->launch_mutex);
    SmootherIO io;
    io.mfs = mfs; io.Pfs = Pfs; io.B = B; io.T = T; io.mss = mss; io.Pss = Pss; io.flags = flags;
    if (sigma && (!sigma->xi || !sigma->w || sigma->s < 1 || sigma->d != m->d)) return fail(ctx, CGP_E_ARG, "cgp_sigma needs xi, w, s >= 1 and d = the model's");
    ModelArgs ma = custom_args(params, param_stride, gamma, gamma_stride, dt, sigma);
    void* args[] = {&io, &ma};
    if (hipModuleLaunchKernel(sigma ? m->sgp_smoother : m->smoother, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess)
        return fail(ctx, CGP_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return CGP_OK;
}

}  // extern "C"
