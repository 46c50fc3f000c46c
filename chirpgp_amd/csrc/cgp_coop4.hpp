// cgp_coop4.hpp -- lane-cooperative (DPP) layout for the d = 4 chirp model: the helpers every cooperative kernel shares
// (DPP moves, the scalar-measurement update, the NLL flush) and the DPP EKF kernel.  That kernel was BASELINE config C2's
// filter until the matrix-core version (cgp_mfma4.hpp) replaced it; it remains for records too long for the MFMA
// kernel's 2 GiB output windows and behind CGP_DPP_KERNEL, and its update is what the sigma-point kernels call.
//
// Why: with one trial per wavefront the generic filter kernel executes every float64 instruction of a step on 64
// identical lanes: 303 VALU + 42 SALU instructions and 1917 cycles per step (profiles/r01_v4_ekf_eks_pmc.json), i.e. the
// SIMD retires one useful lane-operation per 4-cycle instruction slot.  Here the 4 x 4 covariance algebra is spread
// over a 16-lane DPP row instead -- lane (i, j) = (lane >> 2 & 3, lane & 3) owns P[i][j] -- and only the scalar chain
// that cannot be split (softplus -> sincos of the state's frequency component, the mean) stays replicated:
//
//     Q  = P J^T          Q[i][j]  = sum_l J[j][l] P[i][l]        P[i][l]: quad broadcast (DPP quad_perm)
//     Pp = J Q + Sigma    Pp[i][j] = sum_r J[i][l_r] Q[l_r][j]    Q[l_r][j]: row rotation by 4 r lanes (DPP row_ror)
//     PH[j] (column sums over rows: row_ror 4, 8), PH[i] (quad sums), S, 1/S uniform; Pf[i][j] = Pp - PH[i] PH[j] / S
//
// The four 16-lane rows of the wavefront carry identical copies (DPP row operations never cross a row).  J = blockdiag(
// rho Rot(theta), M32) + (d/du_v column) is assembled per lane as a few FMAs with per-lane 0/1/constant coefficients
// fixed at kernel start, so no per-step selects are needed.  Which row a rotated operand came from is discovered at
// start-up by rotating the row index itself, so the code does not depend on the rotation direction convention.
//
// Outputs: lanes 0..15 store Pf[t] as one coalesced 128-B row; lane 0 stores mf[t]; the NLL goes through the same
// 64-step latch / prefix-sum as the generic kernel.
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {

template <int CTRL> CGP_DEV int dpp_i32(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, false); }
template <int CTRL> CGP_DEV double dpp_f64(double x) {
    const int lo = dpp_i32<CTRL>(__double2loint(x));
    const int hi = dpp_i32<CTRL>(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
// Lane N of every 16-lane row to all lanes of the row, as ONE 64-bit move (v_mov_b64_dpp row_newbcast:N; the double-precision
// DPP of gfx90a+ takes no other control): 5 issue cycles where a quad_perm broadcast of a double is two v_mov_b32_dpp (8).
// In the matrix-core layout lane = 16 r + 4 b + q of a kernel whose four MFMA blocks b are replicas, row_bcast_f64<q0> IS the
// quad broadcast of lane q0 (it takes block 0's copy).
template <int N> CGP_DEV double row_bcast_f64(double x) {
    long long v = __builtin_bit_cast(long long, x);
    long long r = __builtin_amdgcn_mov_dpp(v, 0x150 + N, 0xF, 0xF, false);     // (update_dpp would tie the result to a copy of its `old` operand)
    return __builtin_bit_cast(double, r);
}
constexpr int kQuadBcast0 = 0x00, kQuadBcast1 = 0x55, kQuadBcast2 = 0xAA, kQuadBcast3 = 0xFF;
constexpr int kQuadSwap1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int kQuadSwap2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int kRowRor4 = 0x124, kRowRor8 = 0x128, kRowRor12 = 0x12C;

// J[i][l] of the chirp LCD model as  b * s + g0 * jv0 + g1 * jv1 + k  for l != i  (c only sits on the diagonal).
struct OffDiagCoef { double b, g0, g1, k; };
CGP_DEV OffDiagCoef offdiag_coef(int i, int l, const double (&M)[4]) {
    OffDiagCoef o{0.0, 0.0, 0.0, 0.0};
    if (i == 0 && l == 1) o.b = -1.0;
    if (i == 1 && l == 0) o.b = 1.0;
    if (i == 0 && l == 2) o.g0 = 1.0;
    if (i == 1 && l == 2) o.g1 = 1.0;
    if (i == 2 && l == 3) o.k = M[1];
    if (i == 3 && l == 2) o.k = M[2];
    return o;
}

// Measurement constants of one trial in the cooperative layout: H replicated, plus H[i] and H[j] of the lane.
struct Coop4Meas {
    double H0, H1, H2, H3, Hi, Hj, Xi;
    CGP_DEV void load(const FilterIO& io, int64_t trial, int li, int lj) {
        const double* __restrict__ Hp = io.H + trial * io.H_stride;
        H0 = Hp[0]; H1 = Hp[1]; H2 = Hp[2]; H3 = Hp[3];
        Hi = Hp[li]; Hj = Hp[lj];
        Xi = io.Xi[trial * io.Xi_stride];
    }
};

// Scalar-measurement update (filters_smoothers.py:55-68) in the cooperative layout: Pp is the lane's entry of the
// predicted covariance, f0..f3 the (replicated) predicted mean; returns the lane's entry of Pf in P and the updated
// mean in u0..u3, plus S and the innovation for the NLL.
CGP_DEV void coop4_update(const Coop4Meas& M, double Pp, double f0, double f1, double f2, double f3, double y,
                          double& P, double& u0, double& u1, double& u2, double& u3, double& S_out, double& innov_out) {
    double PHj = Pp * M.Hi;                                   // PH[j] = sum_i Pp[i][j] H[i]: sum over the rows
    PHj += dpp_f64<kRowRor4>(PHj);
    PHj += dpp_f64<kRowRor8>(PHj);
    double PHi = Pp * M.Hj;                                   // PH[i] = sum_j Pp[i][j] H[j]: sum over the quad
    PHi += dpp_f64<kQuadSwap1>(PHi);
    PHi += dpp_f64<kQuadSwap2>(PHi);
    double S = M.Hj * PHj;
    S += dpp_f64<kQuadSwap1>(S);
    S += dpp_f64<kQuadSwap2>(S);
    S += M.Xi;
    const double pred = fma(M.H3, f3, fma(M.H2, f2, fma(M.H1, f1, M.H0 * f0)));
    const double innov = y - pred;
    const double rS = rcp_nr(S);
    P = fma(-(PHi * rS), PHj, Pp);                            // Pf = Pp - K (Pp H)^T
    const double g = rS * innov;
    u0 = fma(dpp_f64<kQuadBcast0>(PHj), g, f0);               // mf = mp + K innov, K = PH / S
    u1 = fma(dpp_f64<kQuadBcast1>(PHj), g, f1);
    u2 = fma(dpp_f64<kQuadBcast2>(PHj), g, f2);
    u3 = fma(dpp_f64<kQuadBcast3>(PHj), g, f3);
    S_out = S;
    innov_out = innov;
}

// A trial's output array as a raw buffer: stores take a 32-bit byte offset per lane and the hardware drops the lanes
// whose offset is past the end (and all of them if the output is not wanted: zero records).  "Which lanes write" thus
// becomes data instead of control flow -- an exec-masked store costs a skip branch, and a branch in the middle of a
// step splits the basic block the scheduler works on (cgp_fastmath.hpp).  kOobOffset marks a lane that never writes;
// windows are limited to 2 GiB so that marker + a small immediate offset stays out of range.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;
struct OobWindow {
    __amdgpu_buffer_rsrc_t rsrc;
    CGP_DEV void init(const double* base, int64_t bytes) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, base ? (int)bytes : 0, 0x27000);
    }
    CGP_DEV double load(unsigned off) const {          // lanes past the end (and a missing array) read 0
        const u32x2_t d = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
        return __hiloint2double((int)d.y, (int)d.x);
    }
    CGP_DEV void store(double v, unsigned off) const {
        u32x2_t d; d.x = (unsigned)__double2loint(v); d.y = (unsigned)__double2hiint(v);
        __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, (int)off, 0, 0);
    }
    // the same with a wave-uniform part of the offset in the instruction's SCALAR offset (not part of the range check: a lane
    // is dropped by its vector offset alone, so `soff` must keep in-range lanes inside the window)
    CGP_DEV void store_s(double v, unsigned off, unsigned soff) const {
        u32x2_t d; d.x = (unsigned)__double2loint(v); d.y = (unsigned)__double2hiint(v);
        __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, (int)off, (int)soff, 0);
    }
    CGP_DEV void store2(double a, double b, unsigned off) const {
        u32x4_t d; d.x = (unsigned)__double2loint(a); d.y = (unsigned)__double2hiint(a);
        d.z = (unsigned)__double2loint(b); d.w = (unsigned)__double2hiint(b);
        __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)off, 0, 0);
    }
};
constexpr int64_t kOobMaxBytes = 0x7FFFFF00;

// NLL of up to 64 latched steps: every lane evaluates its increment, inclusive prefix sum across the wave, one
// coalesced store; returns the new running total (wave-uniform).
CGP_DEV double nll_flush_wave(double S_l, double innov_l, int lane, int nsteps, double cum, double* __restrict__ nll_chunk) {
    double v = (lane < nsteps) ? nll_increment(S_l, innov_l) : 0.0;
    v = wave_inclusive_scan(v);
    v += cum;
    if (nll_chunk && lane < nsteps) nll_chunk[lane] = v;
    return readlane_f64(v, nsteps - 1);
}

// Lane's entry of the (symmetric) initial covariance, read from the lower triangle like the generic kernels do.
CGP_DEV double coop4_load_sym_entry(const double* __restrict__ p, int li, int lj) {
    return (li >= lj) ? p[li * 4 + lj] : p[lj * 4 + li];
}

#ifndef CGP_COOP4_HELPERS_ONLY      // cgp_inst_mfma4.hip takes the helpers above and not a second copy of this kernel
__global__ void __launch_bounds__(64) ekf4_coop_kernel(FilterIO io, ModelArgs ma) {
    const int lane = threadIdx.x;
    const int li = (lane >> 2) & 3, lj = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    HarmonicLCD<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    const double rho = model.rho, dt = model.dt, fs = model.fs;
    const double M0 = model.M[0], M1 = model.M[1], M2 = model.M[2], M3 = model.M[3];

    Coop4Meas meas;
    meas.load(io, trial, li, lj);

    // Sigma[i][j] of this lane (models.py:302-308)
    double Sig = 0.0;
    if (li == lj) Sig = (li < 2) ? model.q : (li == 2 ? model.MS[0] : model.MS[2]);
    else if (li + lj == 5) Sig = model.MS[1];

    // J[j][l], l = 0..3 (row j, for Q = P J^T)
    const double ac0 = (lj == 0) ? 1.0 : 0.0, bc0 = (lj == 1) ? 1.0 : 0.0;       // J[j][0] = ac0 c + bc0 s
    const double ac1 = (lj == 1) ? 1.0 : 0.0, bc1 = (lj == 0) ? -1.0 : 0.0;      // J[j][1] = ac1 c + bc1 s
    const double wc0 = (lj == 0) ? 1.0 : 0.0, wc1 = (lj == 1) ? 1.0 : 0.0;       // J[j][2] = wc0 jv0 + wc1 jv1 + kc2
    const double kc2 = (lj == 2) ? M0 : (lj == 3 ? M2 : 0.0);
    const double kc3 = (lj == 2) ? M1 : (lj == 3 ? M3 : 0.0);                    // J[j][3]
    // J[i][l_r]: r = 0 is the diagonal, r = 1..3 whatever row the rotation by 4 r lanes delivers
    const double ar0 = (li < 2) ? 1.0 : 0.0, kr0 = (li == 2) ? M0 : (li == 3 ? M3 : 0.0);
    const OffDiagCoef r1 = offdiag_coef(li, dpp_i32<kRowRor4>(li), model.M);
    const OffDiagCoef r2 = offdiag_coef(li, dpp_i32<kRowRor8>(li), model.M);
    const OffDiagCoef r3 = offdiag_coef(li, dpp_i32<kRowRor12>(li), model.M);

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    double u0 = m0p[0], u1 = m0p[1], u2 = m0p[2], u3 = m0p[3];
    const double* __restrict__ P0p = io.P0 + trial * io.P0_stride;
    double P = coop4_load_sym_entry(P0p, li, lj);

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    double* __restrict__ mfs = io.mfs ? io.mfs + trial * T * 4 : nullptr;
    double* __restrict__ Pfs = io.Pfs ? io.Pfs + trial * T * 16 : nullptr;
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;

    FastMathRegs fm;
    fm.init();
    double cum = 0.0, S_l = 1.0, innov_l = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        for (int slot = 0; slot < nsteps; slot++) {
            const int64_t t = t0 + slot;
            const double y = readlane_f64(ychunk, slot);
            // ---- replicated scalar chain: rotation of the chirp block at frequency g(u2) (models.py:296-301) and N1
            double sp, dsp;
            softplus_pair_uniform(fm, u2, sp, dsp);
            const double w = (kTwoPi * sp) * fs, dw = (kTwoPi * dsp) * fs;
            double s1, c1;
            fast_sincos_uniform(fm, dt * w, s1, c1);
            const double c = c1 * rho, s = s1 * rho;
            const double f0 = fma(c, u0, -s * u1), f1 = fma(s, u0, c * u1);
            const double f2 = fma(M0, u2, M1 * u3), f3 = fma(M2, u2, M3 * u3);
            const double dth = dt * dw;
            const double jv0 = -dth * f1, jv1 = dth * f0;             // d f0 / d u2, d f1 / d u2
            // ---- per-lane Jacobian entries
            const double Jc0 = fma(ac0, c, bc0 * s);
            const double Jc1 = fma(ac1, c, bc1 * s);
            const double Jc2 = fma(wc0, jv0, fma(wc1, jv1, kc2));
            const double Jr0 = fma(ar0, c, kr0);
            const double Jr1 = fma(r1.b, s, fma(r1.g0, jv0, fma(r1.g1, jv1, r1.k)));
            const double Jr2 = fma(r2.b, s, fma(r2.g0, jv0, fma(r2.g1, jv1, r2.k)));
            const double Jr3 = fma(r3.b, s, fma(r3.g0, jv0, fma(r3.g1, jv1, r3.k)));
            // ---- Q = P J^T
            double Q = Jc0 * dpp_f64<kQuadBcast0>(P);
            Q = fma(Jc1, dpp_f64<kQuadBcast1>(P), Q);
            Q = fma(Jc2, dpp_f64<kQuadBcast2>(P), Q);
            Q = fma(kc3, dpp_f64<kQuadBcast3>(P), Q);
            // ---- Pp = J Q + Sigma
            double Pp = fma(Jr0, Q, Sig);
            Pp = fma(Jr1, dpp_f64<kRowRor4>(Q), Pp);
            Pp = fma(Jr2, dpp_f64<kRowRor8>(Q), Pp);
            Pp = fma(Jr3, dpp_f64<kRowRor12>(Q), Pp);
            // ---- update (filters_smoothers.py:55-68)
            double S, innov;
            coop4_update(meas, Pp, f0, f1, f2, f3, y, P, u0, u1, u2, u3, S, innov);
            if (lane == slot) { S_l = S; innov_l = innov; }
            if (Pfs && lane < 16) Pfs[t * 16 + lane] = P;
            if (mfs && lane == 0) {
                *reinterpret_cast<double2*>(mfs + t * 4) = make_double2(u0, u1);
                *reinterpret_cast<double2*>(mfs + t * 4 + 2) = make_double2(u2, u3);
            }
        }
        if (want_nll) cum = nll_flush_wave(S_l, innov_l, lane, nsteps, cum, nll ? nll + t0 : nullptr);
    }
    if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
}

inline int launch_ekf4_coop(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    hipLaunchKernelGGL(ekf4_coop_kernel, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}

#endif

}  // namespace cgp
