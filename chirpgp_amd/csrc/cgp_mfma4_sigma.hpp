// cgp_mfma4_sigma.hpp -- sgp_filter (filters_smoothers.py:446-490) for the chirp / La Scala LCD model, d = 4, with the
// quadrature sums on the float64 matrix cores (BASELINE config C3: Gauss-Hermite order 3, 81 points in 27 groups).
//
// The collapsed quadrature of cgp_coop4_sigma.hpp:sgp4_coop_kernel (one evaluation per GROUP of sigma points: the points of
// a group differ in the last coordinate only and the last two state components are linear) in the register layout of
// v_mfma_f64_4x4x4_4b_f64, which computes, for each of its four blocks b,
//     out[r][q] = C[r][q] + sum_k A[k][r] B[k][q],      A at lane 16 k + 4 b + r,  B at lane 16 k + 4 b + q,  out at 16 r + 4 b + q
// (tools/ubench/dpp_layout.hip):
//   * group p = 16 s + 4 b + k (pass s = 0, 1) is evaluated by the four lanes q of (k, b); lane q keeps component q of
//         z(p) = (g_0, g_1, e_0, e_1),   g = rho Rot(theta(chi_v)) chi_{0,1},   e = M (d_2, d_3)     (d = L xi, xi_3 left out)
//     so  Y = sum_p W_p z z^T  is TWO chained matrix instructions (A = W z, B = z) plus a sum over the four blocks (two DPP
//     adds), and the weighted mean  F = sum_p W_p z  two more (A = W, B = z).  The softplus -> sin / cos chain of a group is
//     evaluated once, by lane q & 1 = s of its quad, and broadcast inside the quad.  Nothing goes through LDS (the
//     LDS-reduced kernel spends two LDS round trips and a 16-deep sum per step on the same sums);
//   * with sum W = 1, sum W xi = 0, sum W xi xi^T = I:
//         mp = (F_0, F_1, M m_lin),    Pp = Y - mp_rot mp_rot^T + L_33^2 M[:,1] M[:,1]^T (lin-lin block) + Sigma
//     -- an exact regrouping of filters_smoothers.py:88-121, the same one cgp_coop8.hpp uses for d = 6 / 8;
//   * the covariance lives in the OUTPUT layout of the instruction, lane (r, b, q) holding P[r][q] in every block b, so
//     the scalar-measurement update (filters_smoothers.py:55-68) is three more matrix instructions and no cross-lane
//     moves: Pp H^T by column (A = H[k], B = Pp), by row (A = Pp, B = H[k]; Pp symmetric) and S = H Pp H^T + Xi;
//   * chol(Pf) is needed by every point: 10 entries gathered with v_readlane, factorisation replicated (as L D L^T, which
//     keeps the square roots off the pivot-to-pivot chain); the mean is replicated (four quad broadcasts of Pp H^T per step).
#pragma once
#include <type_traits>
#include "cgp_coop8.hpp"
#include "cgp_coop4_sigma.hpp"

namespace cgp {

// The groups a lane evaluates: p = 16 s + 4 b + k, pass s = 0, 1 (the four lanes q of (k, b) share them); xi_0..2 of the
// group's first member and its total weight, zero beyond the set's groups.
struct Fan4Groups {
    double xi[2][3], W[2];
    template <bool TWO> CGP_DEV void load(const SigmaSet& sg, int k, int b) {
        CGP_UNROLL for (int s = 0; s < 2; s++) {
            W[s] = 0.0;
            CGP_UNROLL for (int c = 0; c < 3; c++) xi[s][c] = 0.0;
            const int p = 16 * s + 4 * b + k;
            if ((s == 0 || TWO) && p < sg.groups()) {
                const int p0 = sg.template begin<true>(p), p1 = sg.template end<true>(p);
                CGP_UNROLL for (int c = 0; c < 3; c++) xi[s][c] = sg.template coord<true>(p0 * 4 + c);
                for (int i = p0; i < p1; i++) W[s] += sg.template weight<true>(i);
            }
        }
    }
};

// chol(P) = l sqrt(diag(dv)) for a covariance held in the matrix-core layout (lane (r, b, q): P[r][q]): 10 entries
// gathered through LDS, the factorisation replicated in every lane as L D L^T -- the square roots sd = sqrt(dv_0..2)
// stay off the pivot-to-pivot chain (cgp_coop8.hpp); dv[3] is returned as is (the collapsed quadratures use L_33 squared
// or not at all).
CGP_DEV void mfma4_factor(double P, Sym<4>& l, double (&sd)[3], double (&dv)[4]) {
    Sym<4> Pr; bool bad;
    // The gather through LDS (round 4; twenty v_readlane before: C3 filter 6.15 -> 6.03 ms, C4 smoother 97.5 -> 94.5 ms): every lane
    // parks its entry (the four blocks are replicas: same slot, same value), the lower triangle comes back as six broadcast reads --
    // the LDS operations of one wavefront execute in order, so nothing waits between them
    __shared__ __attribute__((aligned(16))) double fbuf[16];
    const int lane = threadIdx.x;
    fbuf[4 * (lane >> 4) + (lane & 3)] = P;
    wave_lds_fence();
    const double2 r0 = *reinterpret_cast<const double2*>(fbuf), r1 = *reinterpret_cast<const double2*>(fbuf + 4);
    const double2 r2a = *reinterpret_cast<const double2*>(fbuf + 8), r2b = *reinterpret_cast<const double2*>(fbuf + 10);
    const double2 r3a = *reinterpret_cast<const double2*>(fbuf + 12), r3b = *reinterpret_cast<const double2*>(fbuf + 14);
    wave_lds_fence();
    Pr(0, 0) = r0.x; Pr(1, 0) = r1.x; Pr(1, 1) = r1.y; Pr(2, 0) = r2a.x; Pr(2, 1) = r2a.y; Pr(2, 2) = r2b.x;
    Pr(3, 0) = r3a.x; Pr(3, 1) = r3a.y; Pr(3, 2) = r3b.x; Pr(3, 3) = r3b.y;
    ldl_lower<4>(Pr, l, dv, bad);
    CGP_UNROLL for (int c = 0; c < 3; c++) sd[c] = sqrt_fast(dv[c]);
}

// One pass of the fan: component q of z for the lane's group, from the replicated factor and mean.  SPEC: without the
// regime branches of the softplus and the sin / cos (cgp_models.hpp:precompute_spec), ok = false where that is not exact.
struct Sgp4LaneCoef {
    bool rot_lane;                    // q < 2: the lane keeps a rotating component
    double kc, ks, kn, kM1, kM2;      // z_q = (kc c + ks s + kM1) x1 + (kn s + ks c + kM2) x2
    CGP_DEV void init(int q, const double (&M)[4]) {
        rot_lane = q < 2;
        kc = (q == 0) ? 1.0 : 0.0; ks = (q == 1) ? 1.0 : 0.0; kn = (q == 0) ? -1.0 : 0.0;
        kM1 = (q == 2) ? M[0] : (q == 3) ? M[2] : 0.0; kM2 = (q == 2) ? M[1] : (q == 3) ? M[3] : 0.0;
    }
};
// Both passes of the fan.  The softplus -> sin / cos chain of a group depends on one number, u_v + d_2, so it is evaluated
// ONCE per group: lane q of a quad takes the group of pass q & 1, and the (rho cos, rho sin) pairs reach the quad's four
// lanes as quad broadcasts.  SPEC: without the regime branches (cgp_models.hpp:precompute_spec); ok = false where that is
// not valid.
template <int MODE, bool TWO, class DM>
CGP_DEV void sgp4_mfma_fan(const DM& model, const FanRegs& R, const Sgp4LaneCoef& K, const Sym<4>& l, const double (&sd)[3],
                           double u0, double u1, double u2, const double (&xi)[2][3], bool odd, double& z0, double& z1, bool& ok) {
    double x1[2], x2[2], d2[2];
    CGP_UNROLL for (int s = 0; s < (TWO ? 2 : 1); s++) {                // d = L xi, L = l diag(sd) (l unit lower), xi_3 left out
        const double xs0 = xi[s][0] * sd[0], xs1 = xi[s][1] * sd[1], xs2 = xi[s][2] * sd[2];
        const double d1 = fma(l(1, 0), xs0, xs1);
        d2[s] = fma(l(2, 1), xs1, fma(l(2, 0), xs0, xs2));
        const double d3 = fma(l(3, 2), xs2, fma(l(3, 1), xs1, l(3, 0) * xs0));
        x1[s] = K.rot_lane ? u0 + xs0 : d2[s];
        x2[s] = K.rot_lane ? u1 + d1 : d3;
    }
    typename DM::Pre pre;                                 // rho cos / sin of theta(chi_v)
    const double uv = u2 + ((TWO && odd) ? d2[1] : d2[0]);
    if constexpr (MODE == kFanSpec) model.precompute_spec(R, uv, pre, ok);
    else if constexpr (MODE == kFanAny) model.precompute_any(R, uv, pre, ok);
    else { model.precompute(uv, pre); ok = true; }
    // q = 0: c x1 - s x2;  q = 1: s x1 + c x2;  q = 2, 3: M[q-2][0] x1 + M[q-2][1] x2 (coefficients 0 / +-1 / M: exact)
    const double c0 = TWO ? dpp_f64<kQuadBcast0>(pre.c[0]) : pre.c[0], s0 = TWO ? dpp_f64<kQuadBcast0>(pre.s[0]) : pre.s[0];
    z0 = fma(fma(K.kc, c0, fma(K.ks, s0, K.kM1)), x1[0], fma(K.kn, s0, fma(K.ks, c0, K.kM2)) * x2[0]);
    if constexpr (TWO) {
        const double c1 = dpp_f64<kQuadBcast1>(pre.c[0]), s1 = dpp_f64<kQuadBcast1>(pre.s[0]);
        z1 = fma(fma(K.kc, c1, fma(K.ks, s1, K.kM1)), x1[1], fma(K.kn, s1, fma(K.ks, c1, K.kM2)) * x2[1]);
    }
}

// E1 (round 4): the measurement vector is e_1, as in every chirp / La Scala builder of the reference (models.py:118) -- H then PICKS
// entries of Pp (cgp_mfma4.hpp: ekf4_mfma_finish_j): Pp H by row is a row broadcast of Pp, S = Pp_11 + Xi a row broadcast of
// (H Pp) by column, H . mp = mp_1 -- one matrix instruction where the general update has three, and no dot product.
template <class DM, bool TWO, bool E1, bool SPLIT>
CGP_DEV void sgp4_mfma_trial(const FilterIO& io, const ModelArgs& ma) {
    static_assert(DM::D == 4, "d = 4 kernel");
    __shared__ double2 park[64];                                         // (S, innovation) of the chunk's steps, for the NLL
    __shared__ double ybuf[64 + 2];                                      // the chunk's measurements (+ the read-ahead past the last)
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    // (time-split launches, cgp_filter_time_split: this wavefront is one SEGMENT of its trial's record -- cgp_kernels.hpp: FilterSpan)
    const FilterSpan span = filter_span<SPLIT>(io, blockIdx.x);
    const int64_t trial = span.trial;

    DM model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    model.wide = true;
    SigmaSet sg = ma.sg;
    sg.stage(dyn_lds(), lane, 64, 4);

    Fan4Groups grp;
    grp.template load<TWO>(sg, r, b);
    const double (&xi)[2][3] = grp.xi;
    const double (&W)[2] = grp.W;

    // ---- per-lane constants
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const double H0 = Hp[0], H1 = Hp[1], H2 = Hp[2], H3 = Hp[3];
    const double Hk = Hp[r];                                             // H[k] for the lane's k = lane >> 4, as A or B operand
    const double Xi = io.Xi[trial * io.Xi_stride];
    const double M0 = model.M[0], M1 = model.M[1], M2 = model.M[2], M3 = model.M[3];
    double Sig = 0.0;                                                    // Sigma[r][q] (models.py:302-308)
    {
        Sym<4> Sg;
        CGP_UNROLL for (int k = 0; k < Sym<4>::N; k++) Sg.a[k] = 0.0;
        model.add_sigma(Sg, 1.0);
        CGP_UNROLL for (int i = 0; i < 4; i++) CGP_UNROLL for (int j = 0; j < 4; j++) if (r == i && q == j) Sig = Sg(i, j);
    }
    const double K1 = (r >= 2 && q >= 2) ? model.M[2 * (r - 2) + 1] * model.M[2 * (q - 2) + 1] : 0.0;
    Sgp4LaneCoef K;
    K.init(q, model.M);
    FanRegs R;
    R.init();
    const bool odd = (q & 1) != 0;                                       // the pass whose softplus / sin / cos this lane evaluates
    const double cr0 = (r == 0) ? 1.0 : 0.0, cr1 = (r == 1) ? 1.0 : 0.0;  // mp_rot[r] = cr0 f0 + cr1 f1
    const double mq = K.rot_lane ? 1.0 : 0.0;                             // mp_rot[q] = mq F[q]

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    const double* __restrict__ P0p = io.P0 + trial * io.P0_stride;
    double u0 = m0p[0], u1 = m0p[1], u2 = m0p[2], u3 = m0p[3];
    double P = (r >= q) ? P0p[r * 4 + q] : P0p[q * 4 + r];

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    OobWindow wP, wm, wnull;
    wnull.init(nullptr, 0);                                              // (the burn-in chunks of a time-split segment store through it)
    wP.init(io.Pfs ? io.Pfs + trial * T * 16 : nullptr, T * 128);
    wm.init(io.mfs ? io.mfs + trial * T * 4 : nullptr, T * 32);
    const unsigned offP = (b == 0) ? (unsigned)(4 * r + q) * 8u : kOobOffset;      // block 0 stores the 16 entries: one 128-B row
    const unsigned offm = (lane == 0) ? 0u : kOobOffset;                           // lane 0 stores the mean: two 16-byte stores
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;

    double cum = 0.0;
    for (int64_t t0 = span.t_begin; t0 < span.t_end; t0 += 64) {
        double ychunk = (t0 + lane < span.t_end) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (span.t_end - t0 < 64) ? (int)(span.t_end - t0) : 64;
        // a segment's burn-in chunks (whole chunks: t_out is a multiple of 64) write nothing: their rows belong to the segment before
        const bool burn = t0 < span.t_out;
        const OobWindow wPc = burn ? wnull : wP, wmc = burn ? wnull : wm;   // an empty window drops the stores; the lane offsets stay loop-invariant
        if (span.state && span.seg > 0 && t0 == span.t_out) {          // the junction: the state the burn-in arrived at
            if (lane == 0) { span.state[0] = u0; span.state[1] = u1; span.state[2] = u2; span.state[3] = u3; }
            if (b == 0) span.state[4 + 4 * r + q] = P;
        }
        // the chunk's measurements go through LDS: one broadcast ds_read_b64 per step, issued a step ahead, where a v_readlane
        // pair costs 24 issue cycles (tools/ubench/issue_costs.hip)
        ybuf[lane] = ychunk;
        wave_lds_fence();
        double ynext = ybuf[0];
        for (int slot = 0; slot < nsteps; slot++) {
            const unsigned t = (unsigned)(t0 + slot);
            const double y = ynext;
            ynext = ybuf[slot + 1];
            // ---- sigma-point prediction (filters_smoothers.py:88-121)
            // A pivot <= 0 or NaN turns its square root -- for the last one, which only enters squared, the pivot itself --
            // into NaN, and with it every output of this and all later steps, as the reference's NaN factor does
            Sym<4> l; double sd[3], dv[4];
            mfma4_factor(P, l, sd, dv);
            const double l33sq = (dv[3] > 0.0) ? dv[3] : __builtin_nan("");
            // no branches on the way (one basic block to schedule); the rare lane outside the common regime sends the
            // wavefront through the checked forms afterwards
            bool ok;
            double z0, z1 = 0.0;
            // (round 5: first through the branch-free ANY fan -- the full-accuracy softplus for any |x| < 700, cgp_models.hpp:
            // precompute_any -- and only from there through the checked one with its regime branches.  Starting a wavefront's next
            // steps with the ANY fan, a second copy of the step chosen between steps, was measured too: 7.2 instead of 8.x ms on
            // records outside the lean regime, and 1 - 5 % slower on those inside it in the three kernels that got it: not kept.)
            sgp4_mfma_fan<kFanSpec, TWO>(model, R, K, l, sd, u0, u1, u2, xi, odd, z0, z1, ok);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
                sgp4_mfma_fan<kFanAny, TWO>(model, R, K, l, sd, u0, u1, u2, xi, odd, z0, z1, ok);
                if (__builtin_amdgcn_ballot_w64(!ok) != 0) sgp4_mfma_fan<kFanChecked, TWO>(model, R, K, l, sd, u0, u1, u2, xi, odd, z0, z1, ok);
            }
            double Y = mfma4x4(W[0] * z0, z0, 0.0);
            double F = mfma4x4(W[0], z0, 0.0);
            if constexpr (TWO) {
                Y = mfma4x4(W[1] * z1, z1, Y);
                F = mfma4x4(W[1], z1, F);
            }
            Y = blk_allreduce(Y);                                        // sum_p W z_r z_q
            F = blk_allreduce(F);                                        // sum_p W z_q, in every row r
            const double f0 = row_bcast_f64<0>(F), f1 = row_bcast_f64<1>(F);
            const double f2 = fma(M0, u2, M1 * u3), f3 = fma(M2, u2, M3 * u3);
            const double Fr = fma(cr0, f0, cr1 * f1);
            const double Pp = fma(-Fr, mq * F, Y) + fma(l33sq, K1, Sig);
            // ---- update (filters_smoothers.py:55-68)
            const double PHc = mfma4x4(Hk, Pp, 0.0);                     // sum_k H[k] Pp[k][q]: (Pp H^T)[q] in every row
            double PHr, S, pred;
            if constexpr (E1) {
                PHr = row_bcast_f64<1>(Pp);                              // Pp[r][1]
                S = row_bcast_f64<1>(PHc) + Xi;                          // Pp[1][1] + Xi
                pred = f1;
            } else {
                PHr = mfma4x4(Pp, Hk, 0.0);                              // sum_k Pp[k][r] H[k]: (Pp H^T)[r] in every column
                S = mfma4x4(Hk, PHr, Xi);                                // H Pp H^T + Xi
                pred = fma(H3, f3, fma(H2, f2, fma(H1, f1, H0 * f0)));
            }
            const double innov = y - pred;
            const double rS = rcp_nr1(S);
            P = fma(-(PHr * rS), PHc, Pp);                               // Pf = Pp - K (Pp H)^T
            const double g = rS * innov;
            u0 = fma(row_bcast_f64<0>(PHc), g, f0);                  // mf = mp + K innov
            u1 = fma(row_bcast_f64<1>(PHc), g, f1);
            u2 = fma(row_bcast_f64<2>(PHc), g, f2);
            u3 = fma(row_bcast_f64<3>(PHc), g, f3);
            park[slot] = make_double2(S, innov);                        // every lane holds them: same address, same value
            wPc.store(P, t * 128u + offP);
            wmc.store2(u0, u1, t * 32u + offm);
            wmc.store2(u2, u3, t * 32u + 16u + offm);
        }
        if (want_nll && !burn) {
            wave_lds_fence();
            const double2 si = park[lane < nsteps ? lane : 0];
            cum = nll_flush_wave(si.x, si.y, lane, nsteps, cum, nll ? nll + t0 : nullptr);
            wave_lds_fence();
        }
    }
    if (span.state) {                                                   // the segment's last state and its NLL total, for the fix-up pass
        if (lane == 0) { span.state[20] = u0; span.state[21] = u1; span.state[22] = u2; span.state[23] = u3; span.state[40] = cum; }
        if (b == 0) span.state[24 + 4 * r + q] = P;
    } else if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
}
template <class DM, bool TWO, bool SPLIT>
__global__ void __launch_bounds__(64) sgp4_mfma_kernel(FilterIO io, ModelArgs ma) {
    const int64_t trial = filter_span<SPLIT>(io, blockIdx.x).trial;
    if (trial >= io.B) return;
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const bool e1 = Hp[0] == 0.0 && Hp[1] == 1.0 && Hp[2] == 0.0 && Hp[3] == 0.0;      // a wave-uniform choice
    if (e1) sgp4_mfma_trial<DM, TWO, true, SPLIT>(io, ma);
    else sgp4_mfma_trial<DM, TWO, false, SPLIT>(io, ma);
}

// The matrix-core kernel takes collapsible sets of at most 32 groups whose output windows fit a raw buffer.
inline bool sgp4_mfma_fits(const FilterIO& io, const ModelArgs& ma) {
    return collapsed_ok(ma) && io.T * 128 <= kOobMaxBytes;
}
template <class DM>
inline int launch_sgp4_mfma(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (!sgp4_mfma_fits(io, ma)) return CGP_E_UNSUPPORTED;
    const bool two = ma.sg.n_groups > 16;
    if (io.segs > 1) {                                                              // time-split: one wavefront per (trial, segment)
        const unsigned grid = (unsigned)(io.B * io.segs);
        if (two) hipLaunchKernelGGL((sgp4_mfma_kernel<DM, true, true>), dim3(grid), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
        else hipLaunchKernelGGL((sgp4_mfma_kernel<DM, false, true>), dim3(grid), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    } else if (two) hipLaunchKernelGGL((sgp4_mfma_kernel<DM, true, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    else hipLaunchKernelGGL((sgp4_mfma_kernel<DM, false, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    return hip_rc(hipGetLastError());
}

}  // namespace cgp
