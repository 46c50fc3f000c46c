// What bounds the matrix part of the d = 4 filter step (cgp_mfma4.hpp)?  One wavefront, one SIMD; ticks per iteration of
//   0  the dependent chain alone:        P -> Pa = mfma(P, a) -> S = mfma(a, Pa, c) -> 1/S (rcp + Newton) -> k = PH rS -> P = fma(-k, PH, Pp)
//   1  chain + the step's six other matrix instructions (independent of the chain within a step)
//   2  chain with the general-H ordering: P -> Q -> Pp -> PH -> S -> 1/S -> P   (four matrix instructions on the chain)
//   3  four dependent v_mfma alone       4  four dependent v_fma_f64 alone      5  v_rcp_f64 + Newton alone (dependent)
//   hipcc --offload-arch=gfx950 -O3 -o kf_chain kf_chain.hip && ./kf_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
__device__ __forceinline__ double rcp1(double s) { double r = __builtin_amdgcn_rcp(s); return fma(r, fma(-s, r, 1.0), r); }
template <int MODE> __global__ void k(double* out, long long* cyc, int n) {
    double P = 1.0 + threadIdx.x * 1e-9, a = 0.25, c = 1.5, J = 0.3, ur = 0.1, Sig = 1e-3, H = 0.5;
    asm volatile("" : "+v"(a), "+v"(c), "+v"(J), "+v"(Sig), "+v"(H));
    double acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < n; i++) {
        if (MODE == 0 || MODE == 1) {
            const double Pa = MF(P, a, 0.0);
            const double S = MF(a, Pa, c);
            double Pp = Sig, PHr = Pa, PHq = Pa;
            if (MODE == 1) {
                const double fr = MF(J, ur, 0.0), fq = MF(ur, J, 0.0);
                const double Q = MF(P, J, 0.0);
                PHr = MF(J, Pa, Sig); PHq = MF(Pa, J, Sig);
                Pp = MF(J, Q, Sig);
                ur = fma(PHr, 1e-3, fr); acc += fq;
            }
            const double rS = rcp1(S);
            P = fma(-(PHr * rS), PHq, Pp);
        } else if (MODE == 2) {
            const double Q = MF(P, J, 0.0);
            const double Pp = MF(J, Q, Sig);
            const double PHr = MF(Pp, H, 0.0), PHq = MF(H, Pp, 0.0);
            const double S = MF(H, PHr, c);
            const double rS = rcp1(S);
            P = fma(-(PHr * rS), PHq, Pp);
        } else if (MODE == 3) {
            P = MF(P, a, 0.0); P = MF(P, a, 0.0); P = MF(P, a, 0.0); P = MF(P, a, 0.0);
        } else if (MODE == 4) {
            P = fma(P, a, c); P = fma(P, a, c); P = fma(P, a, c); P = fma(P, a, c);
        } else if (MODE == 5) {
            P = rcp1(P);
        } else if (MODE == 6 || MODE == 7) {
            // the whole E1 step of kf4_mfma_trial (no stores): 6 = the compiler's order, 7 = the order pinned with sched_barrier
            const double y = 0.3;
            if (MODE == 6) {
                const double fr = MF(J, ur, 0.0), fq = MF(ur, J, 0.0);
                const double Q = MF(P, J, 0.0);
                const double Pa = MF(P, a, 0.0);
                const double S = MF(a, Pa, c);
                const double PHr = MF(J, Pa, Sig), PHq = MF(Pa, J, Sig);
                const double Pp = MF(J, Q, Sig);
                const double innov = y - fq;
                const double rS = rcp1(S);
                P = fma(-(PHr * rS), PHq, Pp);
                const double g = rS * innov;
                ur = fma(PHr, g, fr);
                acc = fma(PHq, g, fq);
            } else {
                const double Pa = MF(P, a, 0.0);
                __builtin_amdgcn_sched_barrier(0);
                const double Q = MF(P, J, 0.0);
                __builtin_amdgcn_sched_barrier(0);
                const double S = MF(a, Pa, c);
                __builtin_amdgcn_sched_barrier(0);
                const double Pp = MF(J, Q, Sig);
                __builtin_amdgcn_sched_barrier(0);
                const double r0 = __builtin_amdgcn_rcp(S);
                __builtin_amdgcn_sched_barrier(0);
                const double fq = MF(ur, J, 0.0);
                __builtin_amdgcn_sched_barrier(0);
                const double PHq = MF(Pa, J, Sig);
                __builtin_amdgcn_sched_barrier(0);
                const double e = fma(-S, r0, 1.0);
                const double innov = y - fq;
                const double rS = fma(r0, e, r0);
                __builtin_amdgcn_sched_barrier(0);
                const double PHr = MF(J, Pa, Sig);
                __builtin_amdgcn_sched_barrier(0);
                const double g = rS * innov;
                __builtin_amdgcn_sched_barrier(0);
                const double fr = MF(J, ur, 0.0);
                __builtin_amdgcn_sched_barrier(0);
                acc = fma(PHq, g, fq);
                P = fma(-(PHr * rS), PHq, Pp);
                ur = fma(PHr, g, fr);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 8) {
            // eight INDEPENDENT matrix instructions and nothing else: the pipe's issue rate
            double x0 = MF(a, J, 0.0), x1 = MF(J, a, 0.0), x2 = MF(a, c, 0.0), x3 = MF(c, a, 0.0);
            double x4 = MF(J, c, 0.0), x5 = MF(c, J, 0.0), x6 = MF(H, a, 0.0), x7 = MF(a, H, 0.0);
            asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = P + acc + ur;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h;
    (void)hipMalloc(&out, 64 * sizeof(double)); (void)hipMalloc(&cyc, 8);
    const int n = 1 << 15;
    const char* names[9] = {"E1 chain alone (2 mfma + rcp/Newton + mul + fma)", "E1 chain + 6 independent mfma", "general chain (4 mfma + rcp/Newton + mul + fma)",
                            "4 dependent mfma", "4 dependent v_fma_f64", "rcp + Newton (dependent)", "whole kf step, compiler order", "whole kf step, pinned order",
                            "8 independent mfma"};
    for (int m = 0; m < 9; m++) {
        for (int rep = 0; rep < 2; rep++) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 7) hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 8) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-52s %7.1f ticks per iteration\n", names[m], (double)h / n);
    }
    // wall-clock calibration of the tick: 2^15 iterations of mode 3 timed with events
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc, n * 16); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("tick = %.3f ns (%.0f ticks in %.3f ms)\n", ms * 1e6 / (double)h, (double)h, ms);
    return 0;
}
