// Do LDS writes of a wavefront issue in the shadow of its own v_mfma_f64_4x4x4?  (Vector instructions do not:
// mfma_valu_overlap.hip.)  One wavefront on one SIMD; ticks per iteration of 8 independent matrix instructions, each followed
// by N ds_write_b64, against the same writes alone.  MI355X: 8 matrix instructions 140; 8 / 16 / 32 writes alone 180 / 276 /
// 472 (12 a write), with the matrix instructions 273 / 372 / 589 -- all but ~ 40 of the 140 are added: no.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds_overlap mfma_lds_overlap.hip && ./mfma_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
template <int KIND, int N, bool WITH_MFMA> __global__ void k(double* out, long long* cyc, int n) {
    __shared__ double lds[64 * 8];
    double a = 0.25 + threadIdx.x * 1e-9, b = 0.5;
    asm volatile("" : "+v"(a), "+v"(b));
    double m[8];
    for (int j = 0; j < 8; j++) m[j] = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (WITH_MFMA) m[j] = MF(a, b, m[j]);
#pragma unroll
            for (int r = 0; r < N; r++) {
                if (KIND == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"((int)(threadIdx.x * 8)), "v"(a), "i"(512 * ((j * N + r) % 7)) : "memory");
            }
        }
        if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    double acc = lds[threadIdx.x];
    for (int j = 0; j < 8; j++) acc += m[j];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int N, bool W> double run(double* out, long long* cyc) {
    const int n = 1 << 13; long long h = 0;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<KIND, N, W>), dim3(1), dim3(64), 0, 0, out, cyc, n); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    return (double)h / n;
}
int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * sizeof(double)); (void)hipMalloc(&cyc, 8);
    printf("8 matrix instructions alone: %.1f ticks\n", run<0, 0, true>(out, cyc));
    printf("ds_write_b64 N per matrix instr = 1 / 2 / 4:  alone %6.1f %6.1f %6.1f   with the 8 matrix instr %6.1f %6.1f %6.1f\n",
           run<1, 1, false>(out, cyc), run<1, 2, false>(out, cyc), run<1, 4, false>(out, cyc), run<1, 1, true>(out, cyc), run<1, 2, true>(out, cyc), run<1, 4, true>(out, cyc));
    return 0;
}
