// Microbenchmark: cycles per v_fma_f64 for one wave per SIMD, as a function of ILP (independent chains),
// plus v_rcp_f64 / v_sqrt_f64 / v_readlane and ocml exp/log/sincos cost.  s_memtime around a long unrolled loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP> __global__ void fma_chain(double* out, long long* cyc, int iters) {
    double a[ILP];
    for (int i = 0; i < ILP; i++) a[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const double b = 1.0000001, c = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) a[i] = fma(a[i], b, c);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < ILP; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> __global__ void op_chain(double* out, long long* cyc, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5 + threadIdx.x * 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) a = __builtin_amdgcn_rcp(a) + 1.0;
            if (OP == 1) a = __builtin_amdgcn_sqrt(a) + 1.0;
            if (OP == 2) a = exp(a * 1e-3) ;
            if (OP == 3) a = log(a + 1.5);
            if (OP == 4) { double s, c; sincos(a * 0.01, &s, &c); a = s + c; }
            if (OP == 5) a = 1.0 / (a + 1.0) + 1.0;
            if (OP == 6) a = sqrt(a + 1.0);
            if (OP == 7) { int lo = __builtin_amdgcn_readlane(__double2loint(a), r), hi = __builtin_amdgcn_readlane(__double2hiint(a), r); a = __hiloint2double(hi, lo) + b; }
            if (OP == 8) a = __shfl_xor(a, 1) + b;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K> void run(const char* name, K kern, int grid, int block, int iters, int ops_per_iter) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * grid * block); hipMalloc(&cyc, sizeof(long long) * grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("%-34s grid %5d block %4d : %8.2f cycles/op (memtime ticks)\n", name, grid, block, avg / ((double)iters * ops_per_iter));
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 2000;
    run("fma_f64 ILP1 1wave/SIMD", fma_chain<1>, 1024, 64, it, 16 * 1);
    run("fma_f64 ILP2 1wave/SIMD", fma_chain<2>, 1024, 64, it, 16 * 2);
    run("fma_f64 ILP4 1wave/SIMD", fma_chain<4>, 1024, 64, it, 16 * 4);
    run("fma_f64 ILP8 1wave/SIMD", fma_chain<8>, 1024, 64, it, 16 * 8);
    run("fma_f64 ILP1 2wave/SIMD (blk128x1024)", fma_chain<1>, 1024, 128, it, 16 * 1);
    run("fma_f64 ILP4 2wave/SIMD", fma_chain<4>, 1024, 128, it, 16 * 4);
    run("fma_f64 ILP4 4wave/SIMD", fma_chain<4>, 1024, 256, it, 16 * 4);
    run("fma_f64 ILP1 single wave on chip", fma_chain<1>, 1, 64, it, 16 * 1);
    run("fma_f64 ILP4 single wave on chip", fma_chain<4>, 1, 64, it, 16 * 4);
    run("v_rcp_f64 (+add) dep", op_chain<0>, 1024, 64, it, 8);
    run("v_sqrt_f64 (+add) dep", op_chain<1>, 1024, 64, it, 8);
    run("ocml exp dep", op_chain<2>, 1024, 64, 500, 8);
    run("ocml log dep", op_chain<3>, 1024, 64, 500, 8);
    run("ocml sincos dep", op_chain<4>, 1024, 64, 500, 8);
    run("f64 divide dep", op_chain<5>, 1024, 64, 500, 8);
    run("f64 sqrt() dep", op_chain<6>, 1024, 64, 500, 8);
    run("readlane f64 (+add) dep", op_chain<7>, 1024, 64, it, 8);
    run("shfl_xor f64 (+add) dep", op_chain<8>, 1024, 64, it, 8);
    return 0;
}
