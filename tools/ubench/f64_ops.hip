// Dependent-chain latency (cycles per op, one wave per SIMD) of the single instructions and of the engine's own
// elementary functions that sit on the serial chain of a filter step.
//   hipcc --offload-arch=gfx950 -O3 -fno-fast-math -I chirpgp_amd/csrc -I include tools/ubench/f64_ops.hip -o /tmp/f64_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "cgp_models.hpp"

using namespace cgp;

template <int OP> __global__ void op_chain(double* out, long long* cyc, int iters) {
    double a = 7.0 + threadIdx.x * 0.0, b = 0.5;
    FastMathRegs fm;
    fm.init();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) a = fma(a, 1.0000001, 1e-9);
            if (OP == 1) a = __builtin_rint(a * 1.0000001) + 0.25;
            if (OP == 2) a = __builtin_amdgcn_ldexp(a, 1) * 0.5000001;
            if (OP == 3) a = (double)((int)a) + 0.25;
            if (OP == 4) { const int h = __builtin_amdgcn_readfirstlane(__double2hiint(a)); if (h >= 0x40180000 && h < 0x4085E000) a = a * 1.0000001; else a = a + 100.0; }
            if (OP == 5) a = __builtin_amdgcn_rcp(a) + 7.0;
            if (OP == 6) a = rcp_nr(a) + 7.0;
            if (OP == 7) a = fast_exp_core(-a) + 7.0;
            if (OP == 8) { double sp, dsp; softplus_pair_uniform(fm, a, sp, dsp); a = sp + dsp * 1e-9; }
            if (OP == 9) { double sp, dsp; softplus_pair_uniform(a, sp, dsp); a = sp + dsp * 1e-9; }
            if (OP == 10) { double s, c; fast_sincos_uniform(fm, a, s, c); a = 7.0 + s + c * 1e-3; }
            if (OP == 11) { double s, c; fast_sincos(a, s, c); a = 7.0 + s + c * 1e-3; }
            if (OP == 12) { double sp, dsp; softplus_pair(a, sp, dsp); a = sp + dsp * 1e-9; }
            if (OP == 13) a = __hiloint2double(__double2hiint(a) ^ 0, __double2loint(a)) * 1.0000001;
            if (OP == 14) a = (a > 3.0 ? a : b) * 1.0000001;
            if (OP == 15) a = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1e-9, a, 0, 0, 0);
            if (OP == 16) a = __builtin_amdgcn_frexp_mant(a) + 6.5;
            if (OP == 17) a = a * 1.0000001 + 1e-9;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K> void run(const char* name, K kern, int iters) {
    const int grid = 1024, block = 64;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * grid * block); hipMalloc(&cyc, sizeof(long long) * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("%-52s %8.2f ticks/op   %8.2f ns/op (wall)\n", name, avg / ((double)iters * 8), ms * 1e6 / ((double)iters * 8));
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 1000;
    run("v_fma_f64 dependent", op_chain<0>, it);
    run("v_mul + v_rint_f64 + v_add", op_chain<1>, it);
    run("v_ldexp_f64 + v_mul", op_chain<2>, it);
    run("cvt f64->i32->f64 + add", op_chain<3>, it);
    run("readfirstlane + s_cmp + branch + mul", op_chain<4>, it);
    run("v_rcp_f64 + add", op_chain<5>, it);
    run("rcp_nr + add", op_chain<6>, it);
    run("fast_exp_core + add", op_chain<7>, it);
    run("softplus_pair_uniform (pinned regs, Estrin)", op_chain<8>, it);
    run("softplus_pair_uniform (literals, Horner)", op_chain<9>, it);
    run("fast_sincos_uniform (pinned regs)", op_chain<10>, it);
    run("fast_sincos (per lane)", op_chain<11>, it);
    run("softplus_pair (per lane, naive form)", op_chain<12>, it);
    run("xor sign bits + mul", op_chain<13>, it);
    run("v_cmp + v_cndmask + mul", op_chain<14>, it);
    run("v_mfma_f64_4x4x4 dependent (C and A)", op_chain<15>, it);
    run("v_frexp_mant_f64 + add", op_chain<16>, it);
    run("mul + add (2 dependent)", op_chain<17>, it);
    return 0;
}
