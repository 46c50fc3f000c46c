// Latency of the links of the d = 4 EKF step's dependent chain (cgp_mfma4.hpp), one wavefront on one SIMD: ticks per
// iteration of   x -> LINK -> v_fma_f64 -> x   minus the v_fma_f64 alone.  The value chain stays finite (the link's result is
// scaled back into range by the fma).
//   hipcc --offload-arch=gfx950 -O3 -o chain_links chain_links.hip && ./chain_links
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
__device__ __forceinline__ double dpp_bcast1(double x) {
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x55, 0xf, 0xf, false), hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x55, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane17(double x) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 17), hi = __builtin_amdgcn_readlane(__double2hiint(x), 17);
    return __hiloint2double(hi, lo);
}
template <int LINK> __global__ void k(double* out, long long* cyc, int n) {
    double x = 1.0 + threadIdx.x * 1e-12, a = 0.25, c = 0.75, one = 1.0;
    asm volatile("" : "+v"(a), "+v"(c), "+v"(one));
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        double y = x;
        if (LINK == 1) y = MF(x, one, 0.0);                   // x as the A operand
        if (LINK == 2) y = MF(one, x, 0.0);                   // x as the B operand
        if (LINK == 3) y = MF(a, one, x);                     // x as the C operand
        if (LINK == 4) y = dpp_bcast1(x);
        if (LINK == 5) y = readlane17(x) + one;               // v_readlane pair -> SGPR operand of a v_add_f64
        if (LINK == 6) y = __builtin_amdgcn_rcp(x);
        if (LINK == 7) y = __builtin_amdgcn_ldexp(x, 1);
        if (LINK == 8) y = __builtin_rint(x);
        if (LINK == 9) y = MF(MF(x, one, 0.0), one, 0.0);     // two dependent matrix instructions
        if (LINK == 10) y = dpp_bcast1(MF(x, one, 0.0));      // matrix -> quad broadcast
        if (LINK == 11) y = readlane17(MF(x, one, 0.0)) + one;// matrix -> readlane -> add
        if (LINK == 12) { const double r = __builtin_amdgcn_rcp(x); y = fma(r, fma(-x, r, 1.0), r); }   // rcp + Newton
        x = fma(y, a, c);
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 64 * sizeof(double)); (void)hipMalloc(&cyc, 8);
    const int n = 1 << 15;
    const char* names[13] = {"v_fma_f64 alone", "mfma (A operand)", "mfma (B operand)", "mfma (C operand)", "quad broadcast (2 v_mov_dpp)", "readlane pair + v_add_f64 (SGPR)",
                             "v_rcp_f64", "v_ldexp_f64", "v_rndne_f64", "mfma -> mfma", "mfma -> quad broadcast", "mfma -> readlane pair + add", "v_rcp_f64 + Newton step"};
    double base = 0.0;
    for (int m = 0; m < 13; m++) {
        for (int rep = 0; rep < 2; rep++) {
#define L(M) if (m == M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12)
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        const double t = (double)h / n;
        if (m == 0) base = t;
        printf("%-36s %7.1f ticks (link alone: %6.1f)\n", names[m], t, t - base);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, n * 64); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("tick = %.3f ns\n", ms * 1e6 / (double)h);
    return 0;
}
