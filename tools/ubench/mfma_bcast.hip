// Does v_mfma_f64_4x4x4_4b_f64 honour CBSZ / ABID (broadcast of ONE block's A operand to all four blocks) on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bcast mfma_bcast.hip && ./mfma_bcast
// Lane l = 16 r + 4 b + q.  A operand: lane (r, b, q) supplies A_b[q][r]; B operand: B_b[r][q]; result D_b[r][q].
// Test: A differs per block (A_b = (b + 1) * A0), B = identity.  Without broadcast D_b = A_b; with cbsz = 2, abid = 0 every
// block must return A_0; abid = 3 must return A_3.  Also times the dependent chain with and without the modifier.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID> __global__ void probe(double* out) {
    const int l = threadIdx.x, r = l >> 4, b = (l >> 2) & 3, q = l & 3;
    const double a = (b + 1) * (10.0 * q + r + 1);          // A_b[q][r]
    const double bm = (r == q) ? 1.0 : 0.0;                 // identity
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bm, 0.0, CBSZ, ABID, 0);
}
template <int CBSZ> __global__ void chain(double* out, long long* cyc, int n) {
    double x = 1.0 + threadIdx.x * 1e-9, bq = 0.25;
    asm volatile("" : "+v"(bq));
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < n; i++) x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, bq, 0.0, CBSZ, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; double h[64];
    hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, 8);
    for (int mode = 0; mode < 3; mode++) {
        if (mode == 0) hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, out);
        if (mode == 1) hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, out);
        if (mode == 2) hipLaunchKernelGGL((probe<2, 3>), dim3(1), dim3(64), 0, 0, out);
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        int ok = 1;
        const int src = mode == 0 ? -1 : (mode == 1 ? 0 : 3);
        for (int l = 0; l < 64; l++) {
            const int r = l >> 4, b = (l >> 2) & 3, q = l & 3;
            const int blk = src < 0 ? b : src;
            const double want = (blk + 1) * (10.0 * r + q + 1);      // D_b[r][q] = A_blk[r][q]
            if (h[l] != want) ok = 0;
        }
        printf("cbsz=%d abid=%d : %s   (lane 4 = block 1 holds %.0f, lane 0 = block 0 holds %.0f)\n", mode ? 2 : 0, src < 0 ? 0 : src,
               ok ? "as expected" : "NOT the broadcast semantics", h[4], h[0]);
    }
    const int n = 1 << 16;
    long long hc;
    hipLaunchKernelGGL((chain<0>), dim3(1), dim3(64), 0, 0, out, cyc, n); hipLaunchKernelGGL((chain<0>), dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); printf("dependent chain, plain      %6.1f ticks\n", (double)hc / n);
    hipLaunchKernelGGL((chain<2>), dim3(1), dim3(64), 0, 0, out, cyc, n); hipLaunchKernelGGL((chain<2>), dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); printf("dependent chain, cbsz = 2   %6.1f ticks\n", (double)hc / n);
    return 0;
}
