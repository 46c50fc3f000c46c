// What does the memory system give the OUTPUT PATTERN of the lane-per-trial filters (cgp_kernels.hpp: filter_kernel, one lane
// per trial, 64 trials per wavefront, outputs batch-major [B][T][...])?  Every trial is a stream of its own: per step 128 B of
// covariance, 32 B of mean, 8 B of NLL out and 8 B of measurement in, T * 128 / T * 32 / T * 8 bytes apart from the next trial's.
// This kernel issues exactly those accesses and nothing else (plus NF dependent f64 FMAs per step to stand for the arithmetic):
//   K  = steps staged per covariance store: a trial's visit writes K * 128 contiguous bytes
//   KM = steps staged per mean store (KM * 32 B contiguous), KN = steps per NLL store / measurement load (KN * 8 B)
//   occupancy capped by a dynamic-LDS carve-out (the staging a real kernel would need is what limits it there, too)
// Lane -> (trial, piece) as in block_store_rows: consecutive lanes take consecutive 16-byte pieces of one trial's run, so every
// wave instruction covers whole 128-byte lines (K * 128 / 16 pieces per trial and visit).
// usage: store_pattern [B] [T] [NF]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int K, int KM, int KN, bool WANT_P, bool WANT_NLL, bool NT>
__global__ void __launch_bounds__(64) pattern_kernel(const double* __restrict__ ys, double* __restrict__ Ps, double* __restrict__ ms,
                                                     double* __restrict__ nll, long B, int T, int nf, double seed, int nshift) {
    nll += nshift; ys += nshift;          // rows that start nshift doubles into a line: every 128-byte chunk straddles two lines
    extern __shared__ double carve[];
    const int lane = threadIdx.x;
    if (lane == 1000) carve[0] = 1.0;
    const long first = (long)blockIdx.x * 64;
    double acc = seed + lane;
    constexpr int KB = KN;                      // outer block: the largest staging (KN >= KM >= K assumed, all powers of two)
    for (int t0 = 0; t0 + KB <= T; t0 += KB) {
        // measurements: 64 trials x KN steps, pieces of 16 B
        constexpr int YP = KN / 2;              // 16-byte pieces per trial
        double ysum = 0.0;
#pragma unroll
        for (int i = 0; i < YP; i++) {
            const int g = i * 64 + lane, tr = g / YP, pc = g % YP;
            const double2 v = *reinterpret_cast<const double2*>(ys + (first + tr) * T + t0 + 2 * pc);
            ysum += v.x + v.y;
        }
        acc += ysum;
#pragma unroll 1
        for (int k0 = 0; k0 < KB; k0 += K) {
            for (int k = 0; k < K; k++)
                for (int f = 0; f < nf; f++) acc = __builtin_fma(acc, 0.999999, 1e-9);
            if (WANT_P) {
                constexpr int PP = K * 8;       // 16-byte pieces per trial and visit
#pragma unroll
                for (int i = 0; i < PP; i++) {
                    const int g = i * 64 + lane, tr = g / PP, pc = g % PP;
                    double2* p = reinterpret_cast<double2*>(Ps + ((first + tr) * T + t0 + k0) * 16 + 2 * pc);
                    const double2 v = make_double2(acc, acc + i);
                    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
                    else *p = v;
                }
            }
            if ((k0 + K) % KM == 0) {
                constexpr int MP = KM * 2;
#pragma unroll
                for (int i = 0; i < MP; i++) {
                    const int g = i * 64 + lane, tr = g / MP, pc = g % MP;
                    double2* p = reinterpret_cast<double2*>(ms + ((first + tr) * T + t0 + k0 + K - KM) * 4 + 2 * pc);
                    const double2 v = make_double2(acc, acc - i);
                    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
                    else *p = v;
                }
            }
        }
        if (WANT_NLL) {
#pragma unroll
            for (int i = 0; i < YP; i++) {
                const int g = i * 64 + lane, tr = g / YP, pc = g % YP;
                *reinterpret_cast<double2*>(nll + (first + tr) * T + t0 + 2 * pc) = make_double2(acc, acc * i);
            }
        }
    }
}

static double *ys, *Ps, *ms, *nll;
static long B = 262144; static int T = 512, NF = 0;
static hipEvent_t e0, e1;

template <int K, int KM, int KN, bool WANT_P, bool WANT_NLL, bool NT>
void run(int waves_per_simd, int nshift = 0) {
    const size_t lds = 160 * 1024 / (4 * waves_per_simd) - 64;
    auto kern = pattern_kernel<K, KM, KN, WANT_P, WANT_NLL, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B / 64)), dim3(64), lds, 0, ys, Ps, ms, nll, B, T, NF, 1.0, nshift);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); if (rep > 0 && t < best) best = t;
    }
    const double bytes = (double)B * T * (8 + 32 + (WANT_P ? 128 : 0) + (WANT_NLL ? 8 : 0));
    printf("%-5s K=%-2d KM=%-2d KN=%-2d waves/SIMD=%d %s shift=%d: %7.3f ms  %5.2f TB/s\n", WANT_P ? "full" : "means", K, KM, KN, waves_per_simd,
           NT ? "nt" : "  ", nshift, best, bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) B = atol(argv[1]);
    if (argc > 2) T = atoi(argv[2]);
    if (argc > 3) NF = atoi(argv[3]);
    if (B % 64 || T % 64) { printf("B and T must be multiples of 64\n"); return 1; }
    const size_t n = (size_t)B * T;
    if (hipMalloc(&ys, n * 8 + 256) != hipSuccess || hipMalloc(&Ps, n * 128) != hipSuccess || hipMalloc(&ms, n * 32) != hipSuccess ||
        hipMalloc(&nll, n * 8 + 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(ys, 0, n * 8);
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("B=%ld T=%d NF=%d\n", B, T, NF);
    if (argc > 4) {          // granularity series: which piece sizes does the memory system take at full rate?
        for (int w : {2, 4}) {
            run<1, 4, 16, true, true, false>(w);
            run<1, 4, 16, true, true, false>(w, 4);          // NLL / measurement chunks straddle lines in 32-byte pieces (T = 500)
            run<1, 4, 16, true, true, false>(w, 8);          // ... in 64-byte pieces
            run<1, 4, 8, true, true, false>(w);              // NLL / measurements in aligned 64-byte halves of lines
            run<1, 2, 8, true, true, false>(w);              // ... and the means too
            run<1, 2, 16, true, true, false>(w);
            run<1, 1, 16, true, true, false>(w);             // means in 32-byte pieces
            run<1, 4, 16, true, false, false>(w);            // no NLL rows
        }
        return 0;
    }
    for (int w : {1, 2, 4, 8}) {
        run<1, 4, 16, true, true, false>(w);
        run<2, 4, 16, true, true, false>(w);
        run<4, 4, 16, true, true, false>(w);
        run<8, 8, 16, true, true, false>(w);
        run<16, 16, 16, true, true, false>(w);
        run<4, 16, 16, true, true, false>(w);
        run<4, 4, 64, true, true, false>(w);
        run<4, 4, 16, true, true, true>(w);
        run<1, 4, 16, false, false, false>(w);
        run<1, 8, 16, false, false, false>(w);
        run<1, 16, 16, false, false, false>(w);
        run<1, 16, 64, false, false, false>(w);
        run<1, 64, 64, false, false, false>(w);
        run<1, 16, 16, false, false, true>(w);
    }
    return 0;
}
