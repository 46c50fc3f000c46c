// cgp_coop4_cd.hpp -- lane-cooperative d = 4 continuous-discrete EKF / EKS (RK4 on the moment ODEs) for the chirp /
// La Scala SDE model:
//     cdekf4_coop_kernel   cd_ekf   dm = a(m),                     dP = J P + P J^T + gamma          (filters_smoothers.py:384-394)
//     cdeks4_coop_kernel   cd_eks   dm = a(m) + G^T (m - mf),      dP = A P + P A^T - gamma,  A = J + G^T, G = Pf^{-1} gamma
//                                                                                                     (filters_smoothers.py:427-438)
// Same distributed layout as cgp_coop4.hpp: lane (i, j) owns P[i][j];  (X P)[i][j] = sum_r X[i][l_r] P[l_r][j] by DPP row
// rotations and (P X^T)[i][j] = sum_l P[i][l] X[j][l] by quad broadcasts, with the lane's entries of the drift Jacobian
//     J_a = [[-lam, -w, -dw u1, 0], [w, -lam, dw u0, 0], [0, 0, 0, 1], [0, 0, -g^2, -2 g]]    
// assembled as FMAs with per-lane constant coefficients.  The RK4 bookkeeping of the covariance is 3 instructions.
#pragma once
#include "cgp_coop4_sigma.hpp"

namespace cgp {

// J_a[i][l] for l != i as  b * w + g0 * jv0 + g1 * jv1 + k.
CGP_DEV OffDiagCoef sde_offdiag_coef(int i, int l, double gam) {
    OffDiagCoef o{0.0, 0.0, 0.0, 0.0};
    if (i == 0 && l == 1) o.b = -1.0;
    if (i == 1 && l == 0) o.b = 1.0;
    if (i == 0 && l == 2) o.g0 = 1.0;
    if (i == 1 && l == 2) o.g1 = 1.0;
    if (i == 2 && l == 3) o.k = 1.0;
    if (i == 3 && l == 2) o.k = -(gam * gam);
    return o;
}

// Per-lane constants of the drift Jacobian in the cooperative layout.
struct Coop4SdeJac {
    double lam, gam, fs;
    double kr0;                         // J[i][i]
    OffDiagCoef r1, r2, r3;             // J[i][l_r], r = 1..3
    double ac0, bc0, ac1, bc1;          // J[j][0] = ac0 + bc0 w,  J[j][1] = ac1 + bc1 w
    double wc0, wc1, kc2, kc3;          // J[j][2] = wc0 jv0 + wc1 jv1 + kc2,  J[j][3] = kc3
    CGP_DEV void init(const HarmonicSDE<1>& m, int li, int lj) {
        lam = m.lam; gam = m.gam; fs = m.fs;
        kr0 = (li < 2) ? -lam : (li == 3 ? -2.0 * gam : 0.0);
        r1 = sde_offdiag_coef(li, dpp_i32<kRowRor4>(li), gam);
        r2 = sde_offdiag_coef(li, dpp_i32<kRowRor8>(li), gam);
        r3 = sde_offdiag_coef(li, dpp_i32<kRowRor12>(li), gam);
        ac0 = (lj == 0) ? -lam : 0.0; bc0 = (lj == 1) ? 1.0 : 0.0;
        ac1 = (lj == 1) ? -lam : 0.0; bc1 = (lj == 0) ? -1.0 : 0.0;
        wc0 = (lj == 0) ? 1.0 : 0.0; wc1 = (lj == 1) ? 1.0 : 0.0;
        kc2 = (lj == 3) ? -(gam * gam) : 0.0;
        kc3 = (lj == 2) ? 1.0 : (lj == 3 ? -2.0 * gam : 0.0);
    }
    // Drift a(m) (replicated) and the lane's Jacobian entries at m (wave-uniform state).
    template <class FM>
    CGP_DEV void eval(const FM& fm, const Vec<4>& m, Vec<4>& a, double (&Jr)[4], double (&Jc)[4]) const {
        double sp, dsp;
        softplus_pair_uniform(fm, m.v[2], sp, dsp);
        const double w = (kTwoPi * sp) * fs, dw = (kTwoPi * dsp) * fs;
        a.v[0] = -lam * m.v[0] - w * m.v[1];
        a.v[1] = w * m.v[0] - lam * m.v[1];
        a.v[2] = m.v[3];
        a.v[3] = -(gam * gam) * m.v[2] - 2.0 * gam * m.v[3];
        const double jv0 = -dw * m.v[1], jv1 = dw * m.v[0];
        Jr[0] = kr0;
        Jr[1] = fma(r1.b, w, fma(r1.g0, jv0, fma(r1.g1, jv1, r1.k)));
        Jr[2] = fma(r2.b, w, fma(r2.g0, jv0, fma(r2.g1, jv1, r2.k)));
        Jr[3] = fma(r3.b, w, fma(r3.g0, jv0, fma(r3.g1, jv1, r3.k)));
        Jc[0] = fma(bc0, w, ac0);
        Jc[1] = fma(bc1, w, ac1);
        Jc[2] = fma(wc0, jv0, fma(wc1, jv1, kc2));
        Jc[3] = kc3;
    }
};

// (X P)[i][j] + (P X^T)[i][j] for the lane, given its entries X[i][l_r] (xr) and X[j][l] (xc), plus `add`.
CGP_DEV double coop4_lyapunov(const double (&xr)[4], const double (&xc)[4], double P, double add) {
    double s = fma(xr[0], P, add);
    s = fma(xr[1], dpp_f64<kRowRor4>(P), s);
    s = fma(xr[2], dpp_f64<kRowRor8>(P), s);
    s = fma(xr[3], dpp_f64<kRowRor12>(P), s);
    double t = xc[0] * dpp_f64<kQuadBcast0>(P);
    t = fma(xc[1], dpp_f64<kQuadBcast1>(P), t);
    t = fma(xc[2], dpp_f64<kQuadBcast2>(P), t);
    t = fma(xc[3], dpp_f64<kQuadBcast3>(P), t);
    return s + t;
}

__global__ void __launch_bounds__(64) cdekf4_coop_kernel(FilterIO io, ModelArgs ma) {
    const int lane = threadIdx.x;
    const int li = (lane >> 2) & 3, lj = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    HarmonicSDE<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    Coop4SdeJac jac;
    jac.init(model, li, lj);
    SpecRegs fm;               // the lean wave-uniform softplus: the drift needs no sincos
    fm.init();
    Coop4Meas meas;
    meas.load(io, trial, li, lj);
    const double gam = coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, li, lj);
    const double dt = ma.dt;

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    Vec<4> u;
    u.v[0] = m0p[0]; u.v[1] = m0p[1]; u.v[2] = m0p[2]; u.v[3] = m0p[3];
    double P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, li, lj);
    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    Coop4FilterOut out;
    out.init(io, trial);

    double cum = 0.0, S_l = 1.0, innov_l = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        for (int slot = 0; slot < nsteps; slot++) {
            const double y = readlane_f64(ychunk, slot);
            Vec<4> tm = u, am, km;
            double tP = P, aP = 0.0;
            CGP_UNROLL for (int i = 0; i < 4; i++) am.v[i] = 0.0;
#pragma unroll 1
            for (int stage = 0; stage < 4; stage++) {
                double Jr[4], Jc[4];
                jac.eval(fm, tm, km, Jr, Jc);
                const double kP = coop4_lyapunov(Jr, Jc, tP, gam);                   // J P + P J^T + gamma
                const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
                const double half = (stage == 2) ? 1.0 : 0.5;
                CGP_UNROLL for (int i = 0; i < 4; i++) { am.v[i] = fma(wgt, km.v[i], am.v[i]); tm.v[i] = u.v[i] + (dt * km.v[i]) * half; }
                aP = fma(wgt, kP, aP);
                tP = P + (dt * kP) * half;
            }
            const double f0 = u.v[0] + (dt * am.v[0]) / 6.0, f1 = u.v[1] + (dt * am.v[1]) / 6.0;
            const double f2 = u.v[2] + (dt * am.v[2]) / 6.0, f3 = u.v[3] + (dt * am.v[3]) / 6.0;
            const double Pp = P + (dt * aP) / 6.0;
            double S, innov;
            coop4_update(meas, Pp, f0, f1, f2, f3, y, P, u.v[0], u.v[1], u.v[2], u.v[3], S, innov);
            if (lane == slot) { S_l = S; innov_l = innov; }
            out.store(t0 + slot, lane, P, u.v[0], u.v[1], u.v[2], u.v[3]);
        }
        if (out.want_nll) cum = nll_flush_wave(S_l, innov_l, lane, nsteps, cum, out.nll ? out.nll + t0 : nullptr);
    }
    if (lane == 0 && io.nll && out.nll_final) io.nll[trial] = cum;
}

__global__ void __launch_bounds__(64) cdeks4_coop_kernel(SmootherIO io, ModelArgs ma) {
    __shared__ __attribute__((aligned(16))) double gbuf[64 * kGainPitch];
    const int lane = threadIdx.x;
    const int li = (lane >> 2) & 3, lj = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    HarmonicSDE<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    Coop4SdeJac jac;
    jac.init(model, li, lj);
    SpecRegs fm;               // lean softplus, coefficients left to the compiler (the pinned form costs registers this kernel needs)
    fm.init<false>();
    Sym<4> gamma;
    load_sym<4>(ma.gamma + trial * ma.gamma_stride, gamma);
    const double gam = coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, li, lj);
    const double dt = -ma.dt;
    const int lr1 = dpp_i32<kRowRor4>(li), lr2 = dpp_i32<kRowRor8>(li), lr3 = dpp_i32<kRowRor12>(li);

    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * 4;
    const double* __restrict__ Pfs = io.Pfs + trial * T * 16;
    double* __restrict__ mss = io.mss + trial * T * 4;
    double* __restrict__ Pss = io.Pss + trial * T * 16;

    Vec<4> ms;
    load_vec<4>(mfs + (T - 1) * 4, ms);
    double Ps = coop4_load_sym_entry(Pfs + (T - 1) * 16, li, lj);
    if (lane < 16) Pss[(T - 1) * 16 + lane] = Pfs[(T - 1) * 16 + lane];      // filters_smoothers.py:140-142, verbatim copy
    if (lane < 4) mss[(T - 1) * 4 + lane] = mfs[(T - 1) * 4 + lane];

    for (int64_t t_hi = T - 2; t_hi >= 0; t_hi -= 64) {
    const int nsteps = t_hi + 1 < 64 ? (int)(t_hi + 1) : 64;
    coop4_chunk_gains(gbuf, lane, nsteps, t_hi, mfs, Pfs, gamma);      // Pf^{-1} gamma of the chunk's steps, one step per lane
    for (int slot = 0; slot < nsteps; slot++) {
        const int64_t t = t_hi - slot;
        const double* gl = gbuf + slot * kGainPitch;
        Mat<4> PG; Vec<4> mf;                                            // constant over the 4 stages
        coop4_read_gain(gl, PG, mf);
        // A = J + G^T: the lane's A[i][l_r] = J[i][l_r] + G[l_r][i] and A[j][l] = J[j][l] + G[l][j]
        const double gr[4] = {gl[li * 4 + li], gl[lr1 * 4 + li], gl[lr2 * 4 + li], gl[lr3 * 4 + li]};
        const double gc[4] = {gl[0 * 4 + lj], gl[1 * 4 + lj], gl[2 * 4 + lj], gl[3 * 4 + lj]};

        Vec<4> tm = ms, am, km;
        double tP = Ps, aP = 0.0;
        CGP_UNROLL for (int i = 0; i < 4; i++) am.v[i] = 0.0;
#pragma unroll 1
        for (int stage = 0; stage < 4; stage++) {
            double Jr[4], Jc[4];
            jac.eval(fm, tm, km, Jr, Jc);
            CGP_UNROLL for (int i = 0; i < 4; i++) {
                double s = km.v[i];
                CGP_UNROLL for (int k = 0; k < 4; k++) s = fma(PG.a[k][i], tm.v[k] - mf.v[k], s);
                km.v[i] = s;                                                 // a(m) + gamma Pf^{-1} (m - mf)
            }
            CGP_UNROLL for (int r = 0; r < 4; r++) { Jr[r] += gr[r]; Jc[r] += gc[r]; }
            const double kP = coop4_lyapunov(Jr, Jc, tP, -gam);              // A P + P A^T - gamma
            const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
            const double half = (stage == 2) ? 1.0 : 0.5;
            CGP_UNROLL for (int i = 0; i < 4; i++) { am.v[i] = fma(wgt, km.v[i], am.v[i]); tm.v[i] = ms.v[i] + (dt * km.v[i]) * half; }
            aP = fma(wgt, kP, aP);
            tP = Ps + (dt * kP) * half;
        }
        CGP_UNROLL for (int i = 0; i < 4; i++) ms.v[i] = ms.v[i] + (dt * am.v[i]) / 6.0;
        Ps = Ps + (dt * aP) / 6.0;
        if (lane < 16) Pss[t * 16 + lane] = Ps;
        if (lane == 0) store_vec<4>(mss + t * 4, ms);
    }
    wave_lds_fence();
    }
}

inline int launch_cdekf4_coop(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    hipLaunchKernelGGL(cdekf4_coop_kernel, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}
inline int launch_cdeks4_coop(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    hipLaunchKernelGGL(cdeks4_coop_kernel, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}

}  // namespace cgp
