// cgp_kpt8.hpp -- ekf_for_kpt (filters_smoothers.py:267-314) one wavefront per trial in the 8 x 8 tile layout of cgp_coop8.hpp.
//
// The KPT baseline (models.py:522-580; tetralith/jobs/kpt_mle.py, harmonic_kpt_mle.py) has LINEAR dynamics of dimension d = n_harm + 2
// (any dense F, Sigma: filters_smoothers.py:298) and the nonlinear scalar measurement h(x) = sum_k x_k sin(k g(x_0 + x_{d-1})); per step
//     mp = F mf,  Pp = F Pf F^T + Sigma;   H = grad h(mp), pred = h(mp);   S = H Pp H^T + Xi, K = Pp H / S, mf = mp + K (y - pred),
//     Pf = Pp - K K^T S.
// The generic wavefront kernel (cgp_kernels.hpp: filter_kernel<EkfPredict<KptLinear>, KptUpdate>) runs all of that redundantly in every
// lane: d^2-term dot products as dependent chains between two transcendental evaluations.  Here the covariance lies one entry per lane:
//   * the linear part is the matrix-instruction sequence of ekf8_coop_kernel with a CONSTANT Jacobian F -- mean by row and by column, W = P F^T,
//     Pp = F W + Sigma: six v_mfma_f64_4x4x4 with bank-masked DPP block moves;
//   * the measurement is a wave-uniform scalar chain (softplus -> sincos -> the harmonics by angle addition) on d values read off the
//     row-form mean with v_readlane; its gradient enters the update as per-lane selects of those scalars;
//   * the update is coop8_update's with the innovation y - h(mp) handed in (three matrix instructions, one reciprocal, one FMA per lane).
// d = 3, 4, 5 (n_harm = 1, 2, 3: what the reference's drivers run) share the kernel; entries beyond d are zero and stay zero.
#pragma once
#include "cgp_coop8.hpp"
#include "cgp_mfma4.hpp"

namespace cgp {

// coop8_update for a nonlinear scalar measurement: H = grad h(mp) in row / column form, innovation given.
CGP_DEV void coop8_update_innov(double Pp, double mp, double HR, double HC, double XiC, double innov, double& P, double& mrow, double& S_out) {
    double PHc = mfma4x4(HR, Pp, 0.0);                           // sum_k H[4 I + k] Pp[4 I + k][4 J + q]
    PHc += blk_xor2(PHc);                                        // PH[4 J + q]
    double PHr = mfma4x4(blk_swap12(Pp), HC, 0.0);               // sum_k Pp[4 I + r][4 J + k] H[4 J + k]
    PHr += blk_xor1(PHr);                                        // PH[4 I + r]
    double S = mfma4x4(HR, PHr, XiC);                            // sum_k H[4 I + k] PH[4 I + k]
    S += blk_xor2(S);
    const double rS = rcp_nr1(S);
    P = fma(-(PHr * rS), PHc, Pp);                               // Pf = Pp - K (Pp H)^T
    mrow = fma(PHr, rS * innov, mp);
    S_out = S;
}

template <int NH>
__global__ void __launch_bounds__(64) kpt8_coop_kernel(FilterIO io, ModelArgs ma) {
    constexpr int D = NH + 2;
    static_assert(D <= 8, "tile layout: d <= 8");
    __shared__ double2 park[64];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int I = b >> 1, J = b & 1;
    const int i = 4 * I + r, j = 4 * J + q;                              // JT: this lane holds F[j][i]; P: P[i][j]
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;
    const bool entry = i < D && j < D;
    const bool mean_lane = (J == 0 && q == 0 && i < D);

    const double* __restrict__ prm = ma.params + trial * ma.param_stride;         // F (d x d, row-major) | Sigma (d x d)
    const double JT = entry ? prm[j * D + i] : 0.0;
    const double Sig = entry ? ((i >= j) ? prm[D * D + i * D + j] : prm[D * D + j * D + i]) : 0.0;   // lower triangle, like load_sym
    const double Xi = io.Xi[trial * io.Xi_stride];
    const double XiC = (I == 0) ? Xi : 0.0;
    // H = grad h(mp): entries 0 and d - 1 carry dsum, entries 1 .. NH the sines (models.py:575-578); row index i, column-form index 4 J + r
    const int ic = 4 * J + r;
    const double aR = (i == 0 || i == D - 1) ? 1.0 : 0.0, aC = (ic == 0 || ic == D - 1) ? 1.0 : 0.0;

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    const double* __restrict__ P0p = io.P0 + trial * io.P0_stride;
    double mrow = (i < D) ? m0p[i] : 0.0;
    double P = entry ? ((i >= j) ? P0p[i * D + j] : P0p[j * D + i]) : 0.0;

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    OobWindow wP, wm;                                                    // which lanes store is an offset, not a branch
    wP.init(io.Pfs ? io.Pfs + trial * T * D * D : nullptr, T * (D * D * 8));
    wm.init(io.mfs ? io.mfs + trial * T * D : nullptr, T * (D * 8));
    const unsigned offP = entry ? (unsigned)(i * D + j) * 8u : kOobOffset;
    const unsigned offm = mean_lane ? (unsigned)i * 8u : kOobOffset;
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;
    auto row_lane = [](int k) { return 16 * (k & 3) + 4 * (2 * (k >> 2)); };      // a lane whose row-form entry is component k

    double cum = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        for (int slot = 0; slot < nsteps; slot++) {
            const unsigned t = (unsigned)(t0 + slot);
            const double y = readlane_f64(ychunk, slot);
            // ---- linear prediction (filters_smoothers.py:298): mean in column form is not needed (no Jacobian column), row form for the update
            const double xc = blk_swap12(mrow);                          // u[4 J + r]
            double fr = mfma4x4(blk_swap12(JT), xc, 0.0);                // sum_k F[4 I + r][4 J + k] u[4 J + k]
            fr += blk_xor1(fr);
            const double W = mfma4x4(blk_rows_of_k1(P), blk_cols_of_k1(JT), mfma4x4(blk_rows_of_k0(P), blk_cols_of_k0(JT), 0.0));
            const double Pp = mfma4x4(blk_rows_of_k1(JT), blk_cols_of_k1(W), mfma4x4(blk_rows_of_k0(JT), blk_cols_of_k0(W), Sig));
            // ---- measurement at mp: wave-uniform scalar chain (models.py:575-578)
            double gs, dgs, s1, c1;
            softplus_pair_uniform(readlane_f64(fr, row_lane(0)) + readlane_f64(fr, row_lane(D - 1)), gs, dgs);
            fast_sincos_uniform(gs, s1, c1);
            double h = 0.0, dsum = 0.0, sn = s1, cs = c1, sR = 0.0, sC = 0.0;
            CGP_UNROLL for (int k = 1; k <= NH; k++) {
                if (k > 1) {                                              // (sin, cos)(k g) from ((k - 1) g) and (g)
                    const double sk = fma(sn, c1, cs * s1), ck = fma(cs, c1, -(sn * s1));
                    sn = sk; cs = ck;
                }
                const double xk = readlane_f64(fr, row_lane(k));
                h = fma(xk, sn, h);
                dsum = fma(xk * cs, (double)k * dgs, dsum);
                sR = (i == k) ? sn : sR;
                sC = (ic == k) ? sn : sC;
            }
            const double HR = fma(aR, dsum, sR), HC = fma(aC, dsum, sC);
            // ---- update
            double S;
            const double innov = y - h;
            coop8_update_innov(Pp, fr, HR, HC, XiC, innov, P, mrow, S);
            park[slot] = make_double2(S, innov);
            wP.store(P, t * (unsigned)(D * D * 8) + offP);
            wm.store(mrow, t * (unsigned)(D * 8) + offm);
        }
        if (want_nll) {
            wave_lds_fence();
            const double2 si = park[lane < nsteps ? lane : 0];
            cum = nll_flush_wave(si.x, si.y, lane, nsteps, cum, nll ? nll + t0 : nullptr);
            wave_lds_fence();
        }
    }
    if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
}

// d <= 4 (n_harm = 1, 2): the covariance fits ONE 4 x 4 block, so the layout of kf4_mfma_kernel (cgp_mfma4.hpp: lane (r, q) of every
// block holds P[r][q], the mean by row and by column) serves -- a product is one matrix instruction instead of two chained ones with
// block moves: mean (2), P F^T, F (P F^T) + Sigma, Pp H by row and by column, S: seven per step.  The measurement's scalars come off
// the column-form mean with quad broadcasts.
template <int NH>
__global__ void __launch_bounds__(64) kpt4_mfma_kernel(FilterIO io, ModelArgs ma) {
    constexpr int D = NH + 2;
    static_assert(D <= 4, "one 4 x 4 block");
    const int lane = threadIdx.x;
    const int r = lane >> 4, q = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;
    const bool entry = r < D && q < D;
    const double* __restrict__ prm = ma.params + trial * ma.param_stride;         // F (d x d, row-major) | Sigma (d x d)
    const double JT = entry ? prm[q * D + r] : 0.0;                               // F[q][r]: A operand "F", B operand "F^T"
    const double Sig = entry ? ((r >= q) ? prm[D * D + r * D + q] : prm[D * D + q * D + r]) : 0.0;
    const double Xi = io.Xi[trial * io.Xi_stride];
    const double aR = (r == 0 || r == D - 1) ? 1.0 : 0.0;                         // H_0 = H_{d-1} = dsum, H_k = sin(k g) (models.py:575-578)

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    const double* __restrict__ P0p = io.P0 + trial * io.P0_stride;
    double ur = (r < D) ? m0p[r] : 0.0, uq = (q < D) ? m0p[q] : 0.0;
    double P = entry ? ((r >= q) ? P0p[r * D + q] : P0p[q * D + r]) : 0.0;

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    OobWindow mfs, Pfs;
    mfs.init(io.mfs ? io.mfs + trial * T * D : nullptr, T * (D * 8));
    Pfs.init(io.Pfs ? io.Pfs + trial * T * D * D : nullptr, T * (D * D * 8));
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;
    const unsigned p_off = (((lane >> 2) & 3) == 0 && entry) ? 8u * (unsigned)(D * r + q) : kOobOffset;
    const unsigned m_off = (lane < D) ? 8u * lane : kOobOffset;

    __shared__ double2 park[64];
    double cum = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        for (int slot = 0; slot < nsteps; slot++) {
            const double y = readlane_f64(ychunk, slot);
            const double f_r = mfma4(JT, ur, 0.0), f_q = mfma4(ur, JT, 0.0);   // F u by row and by column
            const double Q = mfma4(P, JT, 0.0);                                  // P F^T
            const double Pp = mfma4(JT, Q, Sig);                                 // F P F^T + Sigma
            // ---- measurement at mp (wave-uniform values: every lane holds the same)
            double gs, dgs, s1, c1;
            softplus_pair_uniform(dpp_f64<kQuadBcast0>(f_q) + dpp_f64<(D == 3 ? kQuadBcast2 : kQuadBcast3)>(f_q), gs, dgs);
            fast_sincos_uniform(gs, s1, c1);
            double h = s1 * dpp_f64<kQuadBcast1>(f_q), dsum = (dpp_f64<kQuadBcast1>(f_q) * c1) * dgs, Hs = (r == 1) ? s1 : 0.0;
            if constexpr (NH == 2) {
                const double s2 = (s1 + s1) * c1, c2 = fma(c1, c1, -(s1 * s1));
                const double x2 = dpp_f64<kQuadBcast2>(f_q);
                h = fma(x2, s2, h);
                dsum = fma(x2 * c2, 2.0 * dgs, dsum);
                Hs = (r == 2) ? s2 : Hs;
            }
            const double Hr = fma(aR, dsum, Hs);
            // ---- update (filters_smoothers.py:305-311)
            const double PHr = mfma4(Pp, Hr, 0.0), PHq = mfma4(Hr, Pp, 0.0);
            const double S = mfma4(Hr, PHr, Xi);
            const double innov = y - h;
            const double rS = rcp_nr1(S);
            P = fma(-(PHr * rS), PHq, Pp);
            const double g = rS * innov;
            ur = fma(PHr, g, f_r);
            uq = fma(PHq, g, f_q);
            park[slot] = make_double2(S, innov);
            const unsigned t = (unsigned)(t0 + slot);
            Pfs.store_s(P, p_off, t * (unsigned)(D * D * 8));
            mfs.store_s(uq, m_off, t * (unsigned)(D * 8));
        }
        if (want_nll) {
            wave_lds_fence();
            const double2 si = park[lane < nsteps ? lane : 0];
            cum = nll_flush_wave(si.x, si.y, lane, nsteps, cum, nll ? nll + t0 : nullptr);
            wave_lds_fence();
        }
    }
    if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
}

template <int NH>
inline int launch_kpt8_coop(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (io.T * ((NH + 2) * (NH + 2) * 8) > kOobMaxBytes) return CGP_E_UNSUPPORTED;               // output windows (OobWindow)
    if constexpr (NH <= 2) hipLaunchKernelGGL(kpt4_mfma_kernel<NH>, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    else hipLaunchKernelGGL(kpt8_coop_kernel<NH>, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}

}  // namespace cgp
