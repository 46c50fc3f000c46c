// Can a wavefront issue vector instructions in the shadow of v_mfma_f64_4x4x4_4b_f64?  One wavefront on one SIMD; ticks per
// loop iteration of M independent matrix instructions + N independent vector instructions of one kind
//   kind 0: v_fma_f64    kind 1: v_add_u32 (32-bit integer)    kind 2: v_fma_f32    kind 3: v_mov_b32 with DPP quad_perm
// If the matrix pipe were separate, time(M, N) = max(M x 20, N x issue); if the instruction occupies the issue port /
// the double-precision units, time(M, N) = M x 20 + N x issue.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
template <int M, int N, int KIND> __global__ void k(double* out, long long* cyc, int n) {
    double a = 0.25 + threadIdx.x * 1e-9, b = 0.5, c = 1.5;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    double xd[N > 0 ? N : 1]; float xf[N > 0 ? N : 1]; unsigned xi[N > 0 ? N : 1]; double m[M > 0 ? M : 1];
    for (int j = 0; j < N; j++) { xd[j] = 1e-3 * j; xf[j] = 1e-3f * j; xi[j] = j + threadIdx.x; }
    for (int j = 0; j < M; j++) m[j] = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < (M > N ? M : N); j++) {
            if (j < M) { m[j] = MF(a, b, m[j]); }
            if (j < N) {
                if (KIND == 0) { xd[j] = fma(xd[j], a, c); }
                if (KIND == 1) { xi[j] = xi[j] + (unsigned)i; }
                if (KIND == 2) { xf[j] = fmaf(xf[j], 0.5f, 1.0f); }
                if (KIND == 3) { xi[j] = __builtin_amdgcn_mov_dpp(xi[j], 0xB1, 0xf, 0xf, false); }
            }
        }
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (KIND == 0) asm volatile("" : "+v"(xd[j]));
            if (KIND == 1 || KIND == 3) asm volatile("" : "+v"(xi[j]));
            if (KIND == 2) asm volatile("" : "+v"(xf[j]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double acc = 0.0;
    for (int j = 0; j < N; j++) acc += xd[j] + xf[j] + xi[j];
    for (int j = 0; j < M; j++) acc += m[j];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int M, int N, int KIND> double run(double* out, long long* cyc) {
    const int n = 1 << 14; long long h = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<M, N, KIND>), dim3(1), dim3(64), 0, 0, out, cyc, n);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    }
    return (double)h / n;
}
template <int KIND> void table(const char* name, double* out, long long* cyc) {
    printf("%-22s  N = 0      8     16     32     64   (ticks per iteration)\n", name);
    printf("  no matrix instr.   %6.1f %6.1f %6.1f %6.1f %6.1f\n", 0.0, run<0, 8, KIND>(out, cyc), run<0, 16, KIND>(out, cyc), run<0, 32, KIND>(out, cyc), run<0, 64, KIND>(out, cyc));
    printf("  4 v_mfma_f64_4x4x4 %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<4, 0, KIND>(out, cyc), run<4, 8, KIND>(out, cyc), run<4, 16, KIND>(out, cyc), run<4, 32, KIND>(out, cyc), run<4, 64, KIND>(out, cyc));
    printf("  8 v_mfma_f64_4x4x4 %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<8, 0, KIND>(out, cyc), run<8, 8, KIND>(out, cyc), run<8, 16, KIND>(out, cyc), run<8, 32, KIND>(out, cyc), run<8, 64, KIND>(out, cyc));
}
int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * sizeof(double)); (void)hipMalloc(&cyc, 8);
    table<0>("v_fma_f64", out, cyc);
    table<1>("v_add_u32", out, cyc);
    table<2>("v_fma_f32", out, cyc);
    table<3>("v_mov_b32 dpp", out, cyc);
    return 0;
}
