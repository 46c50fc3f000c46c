// cgp_ctx.hpp -- the opaque context of include/chirpgp_hip.h, shared by the translation units that implement the C-ABI.
#pragma once
#include <mutex>
#include <string>
#include <unordered_map>
#include <hip/hip_runtime.h>

struct cgp_ctx {
    int device;
    int num_cus;
    int walk_segments = 0;                       // cgp_debug_set(CGP_DBG_WALK_SEGMENTS): 0 = choose, 1 = off, n = cap
    int lane_buffers = 0;                        // cgp_debug_set(CGP_DBG_LANE_BUFFERS): 0 / 2 = default (rows requested one step ahead), 3 = two where the LDS allows
    unsigned long long* counters = nullptr;      // cgp_debug_set(CGP_DBG_COUNT_REGIMES): eight device counters, NULL = off
    unsigned long long* counters_mem = nullptr;  // the allocation (kept while counting is switched off)
    // Scratch of the time-split launches (segment records of cgp_filter_time_split, composed maps of the time-split smoothers):
    // ONE buffer per stream, kept by the context and grown on demand (cgp::ctx_workspace).  Work on one stream is ordered, and the
    // launches of one C-ABI call are enqueued under `launch_mutex`, so two calls never hold the same buffer at once -- whatever the
    // threads that share the context do (round 6; ADVICE r5: the lock used to be dropped before the kernels were enqueued).
    // `pinned`: sized by cgp_reserve_workspace -- a captured graph may hold the pointer, so it is never freed or regrown by a launch
    // (a launch that needs more gets NULL); only cgp_reserve_workspace / cgp_release_workspace / cgp_destroy touch it.
    struct Workspace { void* p = nullptr; size_t bytes = 0; bool pinned = false; };
    std::recursive_mutex launch_mutex;
    std::mutex ws_mutex;
    std::unordered_map<hipStream_t, Workspace> ws;
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

// The stream's scratch buffer of at least `bytes` bytes, or NULL (allocation failed, or the buffer would have to grow while the
// stream is being captured into a graph: size it first with cgp_reserve_workspace).  Growing waits for the stream's queued work --
// which may still use the old buffer -- and frees it; a buffer that is large enough is returned without any HIP call.
// The caller holds ctx->launch_mutex until its last kernel is enqueued (cgp_filter / cgp_smoother do), so no other call on this
// context can be between "fetched the pointer" and "enqueued its kernels" while the buffer is replaced here.
inline void* ctx_workspace(cgp_ctx* ctx, hipStream_t st, size_t bytes, bool reserve = false) {
    if (!ctx) return nullptr;
    std::lock_guard<std::mutex> lock(ctx->ws_mutex);
    cgp_ctx::Workspace& w = ctx->ws[st];
    if (reserve) w.pinned = true;
    if (w.bytes >= bytes && w.p) return w.p;
    if (w.pinned && !reserve && w.p) return nullptr;      // a reserved buffer is only regrown by cgp_reserve_workspace itself
    if (w.p) {
        if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        (void)hipFree(w.p);
        w.p = nullptr; w.bytes = 0;
    }
    const size_t want = (bytes + (size_t)0xFFFFF) & ~(size_t)0xFFFFF;      // whole MiB: small changes of (B, segments) do not reallocate
    if (hipMalloc(&w.p, want) != hipSuccess) { (void)hipGetLastError(); w.p = nullptr; return nullptr; }
    w.bytes = want;
    return w.p;
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
