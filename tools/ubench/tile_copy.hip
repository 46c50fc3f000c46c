// How much of the time-parallel smoother's memory time is its access SHAPE?  1000 single-wave workgroups each stream a
// record of T rows (20 doubles in, 20 doubles out per row, like (mf, Pf) -> (ms, Ps) at d = 4) tile by tile (256 rows):
//   strided   : lane l moves rows 4 l .. 4 l + 3 of the tile in 16-byte pieces (640 B per lane: every instruction touches 64 lines)
//   coalesced : instruction n moves the tile's bytes [1024 n, 1024 n + 1024) (16 B per lane, 8 full lines per instruction)
// Same bytes, no arithmetic, one wave per SIMD (launch bounds force 512 registers worth of occupancy via a big LDS carve-out).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <bool STRIDED>
__global__ void __launch_bounds__(64) copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, int T) {
    __shared__ double pad[5000];                 // 40 KB: at most 4 workgroups per CU, one per SIMD
    if (threadIdx.x == 1000) pad[0] = 1.0;
    const int lane = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * T * 10;      // double2 units: 10 per row
    for (int t0 = 0; t0 + 256 <= T; t0 += 256) {
        double2 v[40];
        if (STRIDED) {
#pragma unroll
            for (int k = 0; k < 40; k++) v[k] = in[base + (size_t)(t0 + 4 * lane) * 10 + k];
#pragma unroll
            for (int k = 0; k < 40; k++) { v[k].x += 1.0; out[base + (size_t)(t0 + 4 * lane) * 10 + k] = v[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 40; k++) v[k] = in[base + (size_t)t0 * 10 + k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 40; k++) { v[k].x += 1.0; out[base + (size_t)t0 * 10 + k * 64 + lane] = v[k]; }
        }
    }
}
int main() {
    const int B = 1000, T = 9984;
    const size_t n = (size_t)B * T * 10;
    double2 *in, *out;
    hipMalloc(&in, n * sizeof(double2)); hipMalloc(&out, n * sizeof(double2));
    hipMemset(in, 0, n * sizeof(double2));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; variant++) {
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(copy_kernel<true>, dim3(B), dim3(64), 0, 0, in, out, T);
            else hipLaunchKernelGGL(copy_kernel<false>, dim3(B), dim3(64), 0, 0, in, out, T);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
        }
        printf("%-10s %.3f ms  %.2f TB/s (in + out)\n", variant == 0 ? "strided" : "coalesced", best, 2.0 * n * 16 / (best * 1e-3) / 1e12);
    }
    return 0;
}
