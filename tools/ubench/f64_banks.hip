// Does the issue cost of v_fma_f64 (5 ticks in tools/ubench/issue_costs.hip, 4 at the peak rate) depend on the VGPR banks of
// its operands?  Sixteen independent v_fma_f64 per iteration with explicit registers: 64-bit operands start at even
// registers, i.e. sit in banks (0,1) ["A": register index = 0 mod 4] or (2,3) ["B": = 2 mod 4].
//   hipcc --offload-arch=gfx950 -O3 -o f64_banks f64_banks.hip && ./f64_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
// dst/src0 v[2j : 2j+1] for j = 0..15 (alternating A, B), src1 = v[S1 : S1+1], src2 = v[S2 : S2+1]
#define FMA16(OP, S1, S2) \
    OP " v[0:1], v[0:1], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[2:3], v[2:3], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" \
    OP " v[4:5], v[4:5], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[6:7], v[6:7], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" \
    OP " v[8:9], v[8:9], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[10:11], v[10:11], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" \
    OP " v[12:13], v[12:13], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[14:15], v[14:15], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" \
    OP " v[16:17], v[16:17], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[18:19], v[18:19], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" \
    OP " v[20:21], v[20:21], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[22:23], v[22:23], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" \
    OP " v[24:25], v[24:25], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[26:27], v[26:27], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" \
    OP " v[28:29], v[28:29], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n" OP " v[30:31], v[30:31], v[" #S1 ":" #S1 "+1], v[" #S2 ":" #S2 "+1]\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v40","v41","v42","v43","v44","v45","v46","v47"
template <int V> __global__ void k(long long* cyc, int n) {
    asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3ff00000\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" ::: CLOB);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        if (V == 0) asm volatile(FMA16("v_fma_f64", 40, 42) ::: CLOB);      // src1 A, src2 B
        if (V == 1) asm volatile(FMA16("v_fma_f64", 40, 44) ::: CLOB);      // src1 A, src2 A
        if (V == 2) asm volatile(FMA16("v_fma_f64", 42, 46) ::: CLOB);      // src1 B, src2 B
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
#undef FMA16
int main() {
    long long* cyc; long long h = 0;
    (void)hipMalloc(&cyc, 8);
    const int n = 1 << 14;
    const char* names[3] = {"src1 in banks (0,1), src2 in (2,3)", "src1 and src2 both in (0,1)", "src1 and src2 both in (2,3)"};
    for (int m = 0; m < 3; m++) {
        for (int rep = 0; rep < 2; rep++) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, cyc, n);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, cyc, n);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, cyc, n);
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-40s %6.2f ticks per v_fma_f64 (16 per iteration, dst = src0 alternating between the bank pairs)\n", names[m], (double)h / n / 16);
    }
    return 0;
}
