// Where do the cycles of the d = 4 EKF step go?  The step's operations (tools/sched/ekf4_sched.py) as straight-line blocks of
// four steps, one wavefront on one SIMD, no stores: the whole step, its head (softplus -> rotation, vector ALU only) and
// its tail (matrix instructions, lane moves, 1 / S) alone, each in the list scheduler's order and in source order.
//   for p in all head tail; do for o in "" --source-order; do python tools/sched/ekf4_sched.py --part $p --no-stores $o \
//       --emit tools/ubench/ekf4_parts_gen/$p${o:+_src}.inc; done; done
//   hipcc --offload-arch=gfx950 -O3 -fno-fast-math -I chirpgp_amd/csrc -I include -mllvm -amdgpu-mfma-vgpr-form tools/ubench/ekf4_parts.hip -o tools/ubench/ekf4_parts
#include <hip/hip_runtime.h>
#include <cstdio>
#define CGP_EKF4_PARTS_ONLY
#include "cgp_mfma4.hpp"
using namespace cgp;
#define SB __builtin_amdgcn_sched_barrier(0)
template <int WHICH> __global__ void __launch_bounds__(64) k(double* out, long long* cyc, int n) {
    const int lane = threadIdx.x;
    Ekf4MfmaConst K;
    K.ang = 6.283e-3; K.rho = 0.9999; K.Sig = (lane % 5 == 0) ? 1e-5 : 0.0; K.SigHq = ((lane & 3) == 1) ? 1e-5 : 0.0; K.Xi = 0.1;
    K.kc = (lane % 21 == 0) ? 1.0 : 0.0; K.ks = (lane == 16) ? 1.0 : (lane == 1 ? -1.0 : 0.0); K.kj = (lane == 32) ? -1.0 : (lane == 33 ? 1.0 : 0.0);
    K.kk = (lane == 42 || lane == 63) ? 0.999 : 0.0;
    K.fold();
    asm volatile("" : "+v"(K.ang), "+v"(K.Sig), "+v"(K.SigHq), "+v"(K.Xi), "+v"(K.kcr), "+v"(K.ksr), "+v"(K.kk), "+v"(K.kja));
    SpecRegs R;
    R.init(K.ang);
    double P = ((lane >> 4) == (lane & 3)) ? 1.0 : 0.0, ur = 0.5 + (lane >> 4), uq = 0.5 + (lane & 3) + 4.0 * ((lane & 3) == 2), th = 0.04, c = cos(0.04), s = sin(0.04);
    double ychunk = 0.3 + 1e-3 * lane, accD = 0.0;
    unsigned accU = 0u;
    const int slot = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        const double P0 = P, ur0 = ur, uq0 = uq, th0 = th, c0 = c, s0 = s, accD0 = accD;
        const unsigned accU0 = accU;
        if constexpr (WHICH == 0) {
#include "ekf4_parts_gen/all.inc"
            P = P4; ur = ur4; uq = uq4; th = th4; c = c4; s = s4; accU = accU4; accD = accD4;
        } else if constexpr (WHICH == 1) {
#include "ekf4_parts_gen/all_src.inc"
            P = P4; ur = ur4; uq = uq4; th = th4; c = c4; s = s4; accU = accU4; accD = accD4;
        } else if constexpr (WHICH == 2) {
#include "ekf4_parts_gen/head.inc"
            P = P4; ur = ur4; uq = uq4; th = th4; c = c4; s = s4; accU = accU4; accD = accD4;
        } else if constexpr (WHICH == 3) {
#include "ekf4_parts_gen/head_src.inc"
            P = P4; ur = ur4; uq = uq4; th = th4; c = c4; s = s4; accU = accU4; accD = accD4;
        } else if constexpr (WHICH == 4) {
#include "ekf4_parts_gen/tail.inc"
            P = P4; ur = ur4; uq = uq4; th = th4; c = c4; s = s4; accU = accU4; accD = accD4;
        } else {
#include "ekf4_parts_gen/tail_src.inc"
            P = P4; ur = ur4; uq = uq4; th = th4; c = c4; s = s4; accU = accU4; accD = accD4;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[lane] = P + ur + uq + th + c + s + accD + accU;
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 64 * sizeof(double)); (void)hipMalloc(&cyc, 8);
    const int n = 1 << 12;
    const char* names[6] = {"whole step, list schedule", "whole step, source order", "head, list schedule", "head, source order", "tail, list schedule", "tail, source order"};
    for (int m = 0; m < 6; m++) {
        for (int rep = 0; rep < 2; rep++) {
#define L(M) if (m == M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, n);
            L(0) L(1) L(2) L(3) L(4) L(5)
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-28s %7.1f ticks per step\n", names[m], (double)h / n / 4);
    }
    return 0;
}
