// cgp_ctx.hpp -- the opaque context of include/chirpgp_hip.h, shared by the translation units that implement the C-ABI.
#pragma once
#include <string>
#include <hip/hip_runtime.h>

struct cgp_ctx {
    int device;
    int num_cus;
    int walk_segments = 0;                       // cgp_debug_set(CGP_DBG_WALK_SEGMENTS): 0 = choose, 1 = off, n = cap
    unsigned long long* counters = nullptr;      // cgp_debug_set(CGP_DBG_COUNT_REGIMES): eight device counters, NULL = off
    unsigned long long* counters_mem = nullptr;  // the allocation (kept while counting is switched off)
};

namespace cgp {
// The error message is kept per THREAD (like errno), tagged with the context it belongs to: several host threads may
// share one context (the Python layer hands one per device to all of them) and a failing call in one must not race with
// cgp_last_error() in another.  The string lives until the calling thread's next failing call.
struct ThreadError { const cgp_ctx* ctx = nullptr; std::string msg; };
inline ThreadError& thread_error() {
    static thread_local ThreadError e;
    return e;
}
inline int fail(cgp_ctx* ctx, int code, const std::string& msg) {
    ThreadError& e = thread_error();
    e.ctx = ctx;
    e.msg = msg;
    return code;
}

// Every entry point runs on the context's device and leaves the calling thread's current device as it found it
// (a framework such as PyTorch tracks the current device itself and must not see it move behind its back).
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess; else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
}  // namespace cgp
