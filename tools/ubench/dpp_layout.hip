// Pins the lane conventions the d = 8 tile-layout kernels rely on (run on MI355X): DPP row rotations / shifts with bank
// masks, quad permutes, and the 4x4x4 four-block f64 MFMA as "out = A^T B per block".
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    int v = l;
    out[0 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x124, 0xF, 0xF, false);   // row_ror:4
    out[1 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x128, 0xF, 0xF, false);   // row_ror:8
    out[2 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    out[3 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x104, 0xF, 0xF, false);   // row_shl:4
    out[4 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x124, 0xF, 0xA, false);   // row_ror:4, banks 1 and 3 written
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xF, 0x2, false);   // row_shr:4, bank 1 written
    out[6 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    out[7 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    out[8 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x140, 0xF, 0xF, false);   // row_mirror
    out[9 * 64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x108, 0xF, 0xC, true);    // row_shl:8 banks 2,3, bound_ctrl
    // mfma: A = lane id, B = one-hot probes
    double a = (double)l, bb = (l == 16 * 2 + 4 * 1 + 3) ? 1.0 : 0.0;                 // B_1[k=2][q=3] = 1
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, 0, 0, 0);
    out[10 * 64 + l] = (int)d;
}
int main() {
    int* d; hipMalloc(&d, 11 * 64 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[11 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"row_ror:4", "row_ror:8", "row_shr:4", "row_shl:4", "row_ror:4 bank 0xA", "row_shr:4 bank 0x2", "quad_perm 1032", "row_half_mirror", "row_mirror", "row_shl:8 bank 0xC bc", "mfma A=lane B=onehot(k2,b1,q3)"};
    for (int t = 0; t < 11; t++) { printf("%-32s:", names[t]); for (int l = 0; l < 32; l++) printf(" %d", h[t * 64 + l]); printf("\n"); }
    return 0;
}
