// Operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found empirically:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
// For every pair (la, lb) of lanes in block 0 the kernel feeds A = delta(lane == la), B = delta(lane == lb), C = 0 and
// records which lanes receive a non-zero D.  A[i][k] * B[k'][j] lands in D[i][j] iff k == k'.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(int la, int lb, double* out) {
    const int lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

int main() {
    double* d;
    hipMalloc(&d, 64 * sizeof(double));
    double h[64];
    printf("rows: A lane la; columns: B lane lb; entry: D lane that is non-zero ('.' = none)\n      ");
    for (int lb = 0; lb < 16; lb++) printf("%3d", lb);
    printf("\n");
    for (int la = 0; la < 16; la++) {
        printf("la=%2d ", la);
        for (int lb = 0; lb < 16; lb++) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, lb, d);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            int hit = -1, n = 0;
            for (int l = 0; l < 64; l++) if (h[l] != 0.0) { hit = l; n++; }
            if (n == 0) printf("  ."); else if (n == 1) printf("%3d", hit); else printf("  *");
        }
        printf("\n");
    }
    // block structure: A in block 1 (lane 16 + la) against B in block 0 and block 1
    for (int bb = 0; bb < 2; bb++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 16, 16 * bb, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("A lane 16, B lane %d ->", 16 * bb);
        for (int l = 0; l < 64; l++) if (h[l] != 0.0) printf(" D lane %d", l);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
