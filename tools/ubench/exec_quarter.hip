// Does a VALU instruction get cheaper when only one 16-lane quarter of the wavefront is active?  (v_fma_f64, ILP 4.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ACTIVE> __global__ void k(double* out, long long* cyc, int iters) {
    double a[4];
    for (int i = 0; i < 4; i++) a[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const double b = 1.0000001, c = 1e-9;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < ACTIVE) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 16; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] = fma(a[i], b, c);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K> void run(const char* name, K kern) {
    const int grid = 1024, iters = 2000;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * grid * 64); hipMalloc(&cyc, sizeof(long long) * grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("%-28s %6.2f ticks/op  %6.3f ns/op\n", name, avg / (iters * 64.0), ms * 1e6 / (iters * 64.0));
}
int main() {
    run("64 lanes active", k<64>);
    run("32 lanes active", k<32>);
    run("16 lanes active", k<16>);
    run(" 1 lane  active", k<1>);
    return 0;
}
