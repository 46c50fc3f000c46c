// Issue cost (ticks per instruction, one wavefront on one SIMD) of the instruction kinds of the d = 4 EKF step: N independent
// dependency chains of one kind in a loop; the slope between N = 8 and N = 24 is the cost of one instruction.
//   hipcc --offload-arch=gfx950 -O3 -o issue_costs issue_costs.hip && ./issue_costs
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, int N> __global__ void k(double* out, long long* cyc, int n) {
    double x[N]; int xi[N];
    double a = 0.999 + threadIdx.x * 1e-9, c = 1e-3;
    asm volatile("" : "+v"(a), "+v"(c));
    for (int j = 0; j < N; j++) { x[j] = 1.0 + 1e-3 * j; xi[j] = j + threadIdx.x; }
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (KIND == 0) x[j] = fma(x[j], a, c);
            if (KIND == 1) x[j] = x[j] * a;
            if (KIND == 2) x[j] = x[j] + c;
            if (KIND == 3) x[j] = __builtin_amdgcn_rcp(x[j]);
            if (KIND == 4) x[j] = __builtin_rint(x[j]);
            if (KIND == 5) x[j] = __builtin_amdgcn_ldexp(x[j], 0);
            if (KIND == 6) xi[j] = (int)x[j] + xi[j];                                   // v_cvt_i32_f64 + v_add_u32
            if (KIND == 7) xi[j] = __builtin_amdgcn_readlane(xi[j], 17) + xi[j];        // v_readlane_b32 + v_add_u32 (SGPR operand)
            if (KIND == 8) x[j] = fmax(x[j], a);
            if (KIND == 9) xi[j] = __builtin_amdgcn_mov_dpp(xi[j], 0x55, 0xf, 0xf, false);
            if (KIND == 10) xi[j] = xi[j] + i;
            if (KIND == 11) x[j] = fma(x[j], -0.5, 1.0);
            if (KIND == 12) { long long v = __builtin_bit_cast(long long, x[j]); v = __builtin_amdgcn_update_dpp(v, v, 0x151, 0xf, 0xf, false); x[j] = __builtin_bit_cast(double, v); }
            if (KIND == 13) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(x[j]) : "v"(a), "v"(c));
        }
#pragma unroll
        for (int j = 0; j < N; j++) { asm volatile("" : "+v"(x[j])); asm volatile("" : "+v"(xi[j])); }
    }
    const long long t1 = __builtin_readcyclecounter();
    double acc = 0.0;
    for (int j = 0; j < N; j++) acc += x[j] + xi[j];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int N> double run(double* out, long long* cyc) {
    const int n = 1 << 13; long long h = 0;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<KIND, N>), dim3(1), dim3(64), 0, 0, out, cyc, n); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    return (double)h / n;
}
int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 64 * sizeof(double)); (void)hipMalloc(&cyc, 8);
    const char* names[14] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rndne_f64", "v_ldexp_f64", "v_cvt_i32_f64 + v_add_u32", "v_readlane_b32 + v_add_u32",
                             "v_max_f64", "v_mov_b32 dpp", "v_add_u32 (SGPR operand)", "v_fma_f64 inline constants", "v_mov_b64 dpp row_newbcast", "v_fmac_f64 dpp row_newbcast"};
#define ROW(K) printf("%-28s %6.2f ticks per instruction (N = 8: %6.1f, N = 24: %6.1f)\n", names[K], (run<K, 24>(out, cyc) - run<K, 8>(out, cyc)) / 16.0, run<K, 8>(out, cyc), run<K, 24>(out, cyc));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13)
    return 0;
}
