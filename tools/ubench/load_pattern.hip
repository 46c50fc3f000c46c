// What does the memory system give the INPUT PATTERN of the lane-per-trial smoothers with selected outputs (cgp_lane4.hpp:
// lane4_smoother_kernel<Step, SELN>; round 6)?  Every trial is a stream of its own, walked BACKWARDS: per step 128 B of filtering
// covariance and 32 B of filtering mean in, and -- with the marginal alone -- 8 B of smoothed mean + 8 B of smoothed variance out
// (176 B a step), T * 128 / T * 32 / T * 8 bytes apart from the next trial's.  This kernel issues exactly those accesses and nothing else
// (plus NF dependent f64 FMAs per step to stand for the arithmetic), the rows of step t - AHEAD requested while step t is consumed:
//   a wave instruction of the covariance read covers 8 trials x one 128-byte row (lane -> (trial, 16-byte piece), as the kernel's LDS-DMA does),
//   the means leave / arrive as whole lines of 4 steps, the selected outputs as whole lines of 16 steps.
// One wavefront per SIMD (LDS carve-out), like the kernel (256 + ~120 registers).
// usage: load_pattern [B] [T] [NF] [AHEAD 1|3] [full 0|1: also write the 160 B rows, i.e. the full smoother's pattern]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int AHEAD, bool FULL>
__global__ void __launch_bounds__(64) pattern_kernel(const double* __restrict__ Pf, const double* __restrict__ mf, double* __restrict__ sm, double* __restrict__ sv,
                                                     double* __restrict__ Ps, double* __restrict__ ms, long B, int T, int nf, double seed) {
    extern __shared__ double carve[];
    const int lane = threadIdx.x;
    if (lane == 1000) carve[0] = 1.0;
    const long first = (long)blockIdx.x * 64;
    double acc = seed + lane;
    double2 row[AHEAD + 1][8];
    auto request = [&](int t, double2 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int g = i * 64 + lane, tr = g >> 3, pc = g & 7;
            r[i] = *reinterpret_cast<const double2*>(Pf + ((first + tr) * T + t) * 16 + 2 * pc);
        }
    };
    // T a multiple of 16; steps walked from T - 1 down
#pragma unroll
    for (int a = 0; a < AHEAD; a++) request(T - 1 - a, row[a]);
    for (int t0 = T - 1; t0 >= 0; t0 -= 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = t0 - k;
            if (t - AHEAD >= 0) request(t - AHEAD, row[(k + AHEAD) % (AHEAD + 1)]);
            if ((k & 3) == 0) {                                      // the means of steps t - 3 .. t: 64 trials x one 128-byte line, eight instructions
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int g = i * 64 + lane, tr = g >> 3, pc = g & 7;
                    const double2 v = *reinterpret_cast<const double2*>(mf + ((first + tr) * T + (t - 3)) * 4 + 2 * pc);
                    acc += v.x + v.y;
                }
            }
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++) s += row[k % (AHEAD + 1)][i].x + row[k % (AHEAD + 1)][i].y;
            acc += s;
            for (int f = 0; f < nf; f++) acc = __builtin_fma(acc, 0.999999, 1e-9);
            if (FULL) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int g = i * 64 + lane, tr = g >> 3, pc = g & 7;
                    *reinterpret_cast<double2*>(Ps + ((first + tr) * T + t) * 16 + 2 * pc) = make_double2(acc, acc + i);
                }
                if ((k & 3) == 3) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int gg = (i * 64 + lane), trial = gg >> 3, piece = gg & 7;
                        *reinterpret_cast<double2*>(ms + ((first + trial) * T + t) * 4 + 2 * piece) = make_double2(acc, acc - i);
                    }
                }
            }
        }
        // selected outputs of the 16 steps t0 - 15 .. t0: one line per trial and array, 8 instructions each
        if (!FULL) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int gg = (i * 64 + lane), trial = gg >> 3, piece = gg & 7;
                *reinterpret_cast<double2*>(sm + (first + trial) * T + (t0 - 15) + 2 * piece) = make_double2(acc, acc + i);
                *reinterpret_cast<double2*>(sv + (first + trial) * T + (t0 - 15) + 2 * piece) = make_double2(acc, acc - i);
            }
        }
    }
    if (acc == 12345.678) sm[0] = acc;
}

int main(int argc, char** argv) {
    const long B = argc > 1 ? atol(argv[1]) : 262144;
    const int T = argc > 2 ? atoi(argv[2]) : 512;
    const int nf = argc > 3 ? atoi(argv[3]) : 0;
    const int ahead = argc > 4 ? atoi(argv[4]) : 1;
    const int full = argc > 5 ? atoi(argv[5]) : 0;
    if (T % 16 || B % 64) { printf("T must be a multiple of 16, B of 64\n"); return 1; }
    double *Pf, *mf, *sm, *sv, *Ps = nullptr, *ms = nullptr;
    const size_t n = (size_t)B * T;
    if (hipMalloc(&Pf, n * 128) != hipSuccess || hipMalloc(&mf, n * 32) != hipSuccess || hipMalloc(&sm, n * 8) != hipSuccess || hipMalloc(&sv, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    if (full && (hipMalloc(&Ps, n * 128) != hipSuccess || hipMalloc(&ms, n * 32) != hipSuccess)) { printf("alloc failed\n"); return 1; }
    hipMemset(Pf, 0, n * 128); hipMemset(mf, 0, n * 32);
    const double bytes = (double)n * (full ? 320.0 : 176.0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t carve = 36 * 1024;                         // four workgroups per CU: one wavefront per SIMD
    auto launch = [&]() {
        const dim3 grid((unsigned)(B / 64)), block(64);
        if (ahead == 1 && !full) hipLaunchKernelGGL((pattern_kernel<1, false>), grid, block, carve, 0, Pf, mf, sm, sv, Ps, ms, B, T, nf, 1.0);
        else if (ahead == 3 && !full) hipLaunchKernelGGL((pattern_kernel<3, false>), grid, block, carve, 0, Pf, mf, sm, sv, Ps, ms, B, T, nf, 1.0);
        else if (ahead == 1) hipLaunchKernelGGL((pattern_kernel<1, true>), grid, block, carve, 0, Pf, mf, sm, sv, Ps, ms, B, T, nf, 1.0);
        else hipLaunchKernelGGL((pattern_kernel<3, true>), grid, block, carve, 0, Pf, mf, sm, sv, Ps, ms, B, T, nf, 1.0);
    };
    launch(); launch();
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < 6; r++) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms_; hipEventElapsedTime(&ms_, e0, e1);
        best = ms_ < best ? ms_ : best; sum += ms_;
    }
    printf("B %ld T %d nf %d ahead %d full %d: best %.3f ms mean %.3f ms = %.2f TB/s (best) for %.2f GB\n", B, T, nf, ahead, full, best, sum / 6, bytes / best / 1e9, bytes / 1e9);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
