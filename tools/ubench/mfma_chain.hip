// Dependent-chain latencies of v_mfma_f64_4x4x4_4b_f64 on gfx950, one wavefront per SIMD (the regime of the wave-per-trial
// kernels):  hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
//   chain A:  x = mfma(x, b, 0)                     result feeds the next instruction's A operand
//   chain C:  x = mfma(a, b, x)                     result feeds the next accumulator
//   chain V:  x = mfma(x - p, b, 0)                 one v_add_f64 between two matrix instructions (the smoother walk's X = Ps - Pp)
//   chain W:  w = mfma(x - p, b, 0); x = mfma(b, w, c)   the walk step of cgp_walk4.hpp / cgp_coop8.hpp
//   fma:      x = fma(x, b, c)                      for scale
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
template <int MODE> __global__ void chain(double* out, long long* cyc, int n) {
    double x = 1.0 + threadIdx.x * 1e-9, b = 0.25, c = 1e-3, p = 1e-6;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(p));
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < n; i++) {
        if (MODE == 0) x = MF(x, b, 0.0);
        if (MODE == 1) x = MF(b, c, x);
        if (MODE == 2) x = MF(x - p, b, 0.0);
        if (MODE == 3) { const double w = MF(x - p, b, 0.0); x = MF(b, w, c); }
        if (MODE == 4) x = fma(x, b, c);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, 8);
    const int n = 1 << 16;
    const char* names[5] = {"mfma -> A operand", "mfma -> accumulator", "mfma -> v_add -> mfma", "walk step (v_add, mfma, mfma)", "v_fma_f64 chain"};
    for (int m = 0; m < 5; m++) {
        long long h = 0;
        for (int rep = 0; rep < 2; rep++) {
            if (m == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 1) hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 2) hipLaunchKernelGGL(chain<2>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 3) hipLaunchKernelGGL(chain<3>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            if (m == 4) hipLaunchKernelGGL(chain<4>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-32s %7.1f clock ticks per iteration\n", names[m], (double)h / n);
    }
    return 0;
}
