// cgp_coop8.hpp -- lane-cooperative kernels for the harmonic chirp models of state dimension 6 and 8 (BASELINE config C5:
// three harmonics, d = 8, cubature rule; demos/ghfs_harmonics_mle.py:25-27).
//
// TILE LAYOUT.  A d x d matrix (d <= 8, padded to 8 x 8 with zeros) lives one entry per lane as 2 x 2 blocks of 4 x 4:
//
//     lane = 16 r + 4 b + q,   b = 2 I + J     holds   X[4 I + r][4 J + q]
//
// i.e. every 16-lane DPP row r carries row r of all four blocks, a DPP bank (4 lanes) is one block row, and the four
// blocks are exactly the four independent products of v_mfma_f64_4x4x4_4b_f64, which computes per block
//     out[r][q] = C[r][q] + sum_k A[k][r] B[k][q]        ("A^T B" in this register indexing; tools/ubench/dpp_layout.hip)
// Blocks move inside a DPP row with row rotations (b <- b - 1: row_ror:4, b <- b ^ 2: row_ror:8) and bank masks.  With
// that, a matrix-vector product of the 8 x 8 covariance is ONE matrix instruction plus one cross-block add, the rank-one
// Kalman update is one FMA per lane, and a step's d^2 covariance entries leave in one coalesced 512-byte store.
//
// sgp8_coop_kernel: sgp_filter (filters_smoothers.py:446-490) for sigma-point sets the host has flagged
// CGP_SIGMA_STANDARD with at most 16 groups (every cubature rule up to d = 8: 2 (d - 1) + 1 groups), in the collapsed
// form of cgp_steps.hpp:sgp4_prediction_collapsed generalised to n harmonics:
//   * the last two state components are linear, f_lin = M chi_lin, and the sigma points of a group differ in the last
//     coordinate only, so each GROUP is evaluated once, at weight W_g, with displacement d = L xi restricted to
//     xi_0..d-2 (L = chol(Pf), lower);
//   * the rows of the "moment matrix" G are g_i = f_i(chi) for the rotating components and e = M (d_v, d_v+1) for the
//     linear pair (the propagated displacement); then with sum W = 1, sum W xi = 0, sum W xi xi^T = I
//         mp  = (sum_p W g_i ; M m_lin)
//         Pp  = sum_p W G G^T - mp_rot mp_rot^T  +  L[d-1][d-1]^2 M[:,1] M[:,1]^T (lin-lin block)  +  Sigma
//     -- an exact regrouping of filters_smoothers.py:88-121.  The sums over the points ARE small matrix products
//     (G W G^T, 8 x 16 x 8): point p = 4 b + r is evaluated by the four lanes q of (r, b), each keeps the rows 4 X + q of
//     G, and tile (X, Y) of the product is one 4x4x4 matrix instruction per block of four points plus a cross-block
//     all-reduce (two DPP adds);
//   * chol(Pf) is needed by every point, so it is computed redundantly by all lanes from an LDS gather of the
//     distributed covariance -- as L D L^T, which keeps square roots off the pivot-to-pivot dependency chain (the
//     reciprocal of a pivot is 3 dependent instructions, a square root 7); a non-positive pivot poisons the step with
//     NaN like the reference's Cholesky does.
#pragma once
#include <atomic>
#include <type_traits>
#include "cgp_coop4.hpp"

namespace cgp {

CGP_DEV double mfma4x4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int CTRL, int BANKS> CGP_DEV double dpp_banks_f64(double old, double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xF, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xF, BANKS, false);
    return __hiloint2double(hi, lo);
}
// value held by block b ^ 2 (same r, q)
CGP_DEV double blk_xor2(double x) { return dpp_f64<kRowRor8>(x); }
// value held by block b ^ 1: odd blocks read b - 1 (row_ror:4), even blocks read b + 1 (row_ror:12)
CGP_DEV double blk_xor1(double x) { return dpp_banks_f64<kRowRor12, 0x5>(dpp_banks_f64<kRowRor4, 0xA>(x, x), x); }
// blocks (0,1) and (1,0) exchanged: the block-transposed arrangement of a symmetric matrix in tile layout
CGP_DEV double blk_swap12(double x) { return dpp_banks_f64<kRowRor4, 0x4>(dpp_banks_f64<kRowRor12, 0x2>(x, x), x); }
// sum over the four blocks, result in all of them
CGP_DEV double blk_allreduce(double x) {
    x += blk_xor2(x);
    return x + dpp_f64<kRowRor4>(x);
}

// An 8 x 8 x 8 product Z = U V in the tile layout is two chained matrix instructions, Z_IJ = U_I0 V_0J + U_I1 V_1J, whose
// operands are block re-arrangements of the TRANSPOSE of U and of V (the instruction computes A^T B per block):
// block (I, J) <- block (K, I): the A operand "U[4 I + r][4 K + k]" from U^T (or from a symmetric U) held in tile layout
CGP_DEV double blk_rows_of_k0(double x) { return dpp_banks_f64<kRowRor8, 0x8>(dpp_banks_f64<kRowRor4, 0x6>(x, x), x); }
CGP_DEV double blk_rows_of_k1(double x) { return dpp_banks_f64<kRowRor12, 0x6>(dpp_banks_f64<kRowRor8, 0x1>(x, x), x); }
// block (I, J) <- block (K, J): the B operand "V[4 K + k][4 J + q]"
CGP_DEV double blk_cols_of_k0(double x) { return dpp_banks_f64<kRowRor8, 0xC>(x, x); }
CGP_DEV double blk_cols_of_k1(double x) { return dpp_banks_f64<kRowRor8, 0x3>(x, x); }

// sqrt(s) by v_rsq_f64 (~2^-24), one Newton (Goldschmidt) step on the root and one residual correction with the seed's own
// half-reciprocal (its 2^-24 error only scales a correction that is already 2^-47): 6 instructions, rounding-level accuracy.
CGP_DEV double sqrt_fast(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    double g = s * y;
    const double h = 0.5 * y;
    g = fma(g, fma(-g, h, 0.5), g);
    return fma(fma(-g, g, s), h, g);
}

// P = L D L^T with unit lower L (strict lower triangle in l) and pivots dv.  Right-looking, fully unrolled; the chain
// from pivot to pivot is reciprocal (3) + scale (1) + update (1).  bad = some pivot <= 0 or NaN.
template <int D> CGP_DEV void ldl_lower(const Sym<D>& P, Sym<D>& l, double (&dv)[D], bool& bad) {
    Sym<D> a = P;
    bad = false;
    CGP_UNROLL for (int j = 0; j < D; j++) {
        const double dj = a(j, j);
        dv[j] = dj;
        bad = bad || !(dj > 0.0);
        if (j < D - 1) {
            const double inv = rcp_nr1(dj);
            double c[D];
            CGP_UNROLL for (int i = j + 1; i < D; i++) { c[i] = a(i, j); l(i, j) = c[i] * inv; }
            CGP_UNROLL for (int i = j + 1; i < D; i++)
                CGP_UNROLL for (int k = j + 1; k <= i; k++) a(i, k) = fma(-l(i, j), c[k], a(i, k));
        }
    }
}

// Scalar-measurement update (filters_smoothers.py:55-68) in the tile layout.  Pp: the lane's entry of the predicted
// covariance; mp: the predicted mean in row form (lane (r, (I, J), q) holds mp[4 I + r]); HR = H[4 I + r], HC = H[4 J + r],
// XiC = Xi on the blocks I = 0 and 0 elsewhere.  Returns the lane's entry of Pf and the updated mean in row form.
CGP_DEV void coop8_update(double Pp, double mp, double HR, double HC, double XiC, double y, double& P, double& mrow, double& S_out, double& innov_out) {
    double PHc = mfma4x4(HR, Pp, 0.0);                           // sum_k H[4 I + k] Pp[4 I + k][4 J + q]
    PHc += blk_xor2(PHc);                                        // PH[4 J + q]
    double PHr = mfma4x4(blk_swap12(Pp), HC, 0.0);               // sum_k Pp[4 I + r][4 J + k] H[4 J + k]
    PHr += blk_xor1(PHr);                                        // PH[4 I + r]
    double S = mfma4x4(HR, PHr, XiC);                            // sum_k H[4 I + k] PH[4 I + k]
    S += blk_xor2(S);
    double pred = mfma4x4(HR, mp, 0.0);
    pred += blk_xor2(pred);
    const double innov = y - pred;
    const double rS = rcp_nr1(S);
    P = fma(-(PHr * rS), PHc, Pp);                               // Pf = Pp - K (Pp H)^T
    mrow = fma(PHr, rS * innov, mp);
    S_out = S;
    innov_out = innov;
}

// AXIAL (round 4): the caller asserts CGP_SIGMA_AXIAL -- every point has at most ONE non-zero coordinate among xi_0..d-2, true of
// every cubature rule (xi = +- sqrt(d) e_k, quadratures.py:138-150).  Then xs = xi . sqrt(diag D') needs one square root per lane,
//     xs_c = sgn(xi_c) sqrt(sum_c xi_c^2 dv_c),
// instead of one per pivot: 20 instead of 49 operations (and one v_rsq_f64 instead of seven) of a step that is bound by
// instruction issue.
template <int NH, bool AXIAL, bool SPLIT>
__global__ void __launch_bounds__(64) sgp8_coop_kernel(FilterIO io, ModelArgs ma) {
    constexpr int D = 2 * NH + 2, NL = 2 * NH, V = NL;
    static_assert(NH == 2 || NH == 3, "d = 6 and d = 8");
    __shared__ __attribute__((aligned(16))) double pbuf[64 + 8 + 64];  // the covariance row-major (pitch 8), the mean, a dump for the lanes without a mean entry
    __shared__ double2 park[64];
    __shared__ double ybuf[64 + 2];                                      // the chunk's measurements (+ the read-ahead past the last)
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int I = b >> 1, J = b & 1;
    const int i = 4 * I + r, j = 4 * J + q;                              // this lane's covariance entry
    const FilterSpan span = filter_span<SPLIT>(io, blockIdx.x);          // (a time-split launch: one SEGMENT of the trial's record)
    const int64_t trial = span.trial;
    if (trial >= io.B) return;

    HarmonicLCD<NH> model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    model.wide = true;
    SigmaSet sg = ma.sg;
    sg.stage(dyn_lds(), lane, 64, D);

    // ---- the group this lane evaluates: p = 4 b + r (the four lanes q share it); xi_0..D-2 of its first member, total weight
    double xi[D - 1], W = 0.0;
    CGP_UNROLL for (int c = 0; c < D - 1; c++) xi[c] = 0.0;
    {
        const int p = 4 * b + r;
        if (p < sg.groups()) {
            const int p0 = sg.template begin<true>(p), p1 = sg.template end<true>(p);
            CGP_UNROLL for (int c = 0; c < D - 1; c++) xi[c] = sg.template coord<true>(p0 * D + c);
            for (int k = p0; k < p1; k++) W += sg.template weight<true>(k);
        }
    }

    double xi2[D - 1], xsg[D - 1], xz = 1.0;                             // AXIAL: xi_c^2, sgn(xi_c), 1 where the lane's point sits on the mean
    CGP_UNROLL for (int c = 0; c < D - 1; c++) {
        xi2[c] = xi[c] * xi[c];
        xsg[c] = (xi[c] > 0.0) ? 1.0 : (xi[c] < 0.0 ? -1.0 : 0.0);
        if (xi[c] != 0.0) xz = 0.0;
    }
    // ---- per-lane constants
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const double HR = (i < D) ? Hp[i] : 0.0;                             // H[4 I + r]
    const double HC = (4 * J + r < D) ? Hp[4 * J + r] : 0.0;             // H[4 J + r]
    const double Xi = io.Xi[trial * io.Xi_stride];
    const double XiC = (I == 0) ? Xi : 0.0;                              // added once in the cross-block sum of S
    double Sig = 0.0;                                                    // Sigma[i][j] (models.py:370-386)
    if (i == j && i < NL) Sig = model.q;
    else if (i == V && j == V) Sig = model.MS[0];
    else if ((i == V + 1 && j == V) || (i == V && j == V + 1)) Sig = model.MS[1];
    else if (i == V + 1 && j == V + 1) Sig = model.MS[2];
    const bool lin_i = (i == V || i == V + 1), lin_j = (j == V || j == V + 1);
    const double K1 = (lin_i && lin_j) ? model.M[2 * (i - V) + 1] * model.M[2 * (j - V) + 1] : 0.0;
    const double Ma = lin_i ? model.M[2 * (i - V)] : 0.0, Mb = lin_i ? model.M[2 * (i - V) + 1] : 0.0;
    const bool entry = i < D && j < D;
    const bool mean_lane = (J == 0 && q == 0 && i < D);
    const int mslot = mean_lane ? 64 + i : 72 + lane;

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    const double* __restrict__ P0p = io.P0 + trial * io.P0_stride;
    double mrow = (i < D) ? m0p[i] : 0.0;                                // mean in row form: m[4 I + r]
    double P = entry ? ((i >= j) ? P0p[i * D + j] : P0p[j * D + i]) : 0.0;

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    double* __restrict__ mfs = io.mfs ? io.mfs + trial * T * D : nullptr;
    double* __restrict__ Pfs = io.Pfs ? io.Pfs + trial * T * D * D : nullptr;
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;
    const double IDENT = (r == q) ? 1.0 : 0.0;                           // per block: A^T B with B = identity transposes A
    FanRegs R;
    R.init();
    // outputs as raw buffer windows: which lanes store is an offset, not a branch (cgp_coop4.hpp:OobWindow)
    OobWindow wP, wm, wnull;
    wnull.init(nullptr, 0);                                              // (the burn-in chunks of a time-split segment store through it)
    wP.init(Pfs, T * (D * D * 8));
    wm.init(mfs, T * (D * 8));
    const unsigned offP = entry ? (unsigned)(i * D + j) * 8u : kOobOffset;
    const unsigned offm = mean_lane ? (unsigned)i * 8u : kOobOffset;

    double cum = 0.0;
    constexpr int kN = D + D * D;                                        // doubles of one (m, P) record of FilterIO::seg_state
    for (int64_t t0 = span.t_begin; t0 < span.t_end; t0 += 64) {
        double ychunk = (t0 + lane < span.t_end) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (span.t_end - t0 < 64) ? (int)(span.t_end - t0) : 64;
        const bool burn = t0 < span.t_out;                               // burn-in chunks of a segment write nothing
        const OobWindow wPc = burn ? wnull : wP, wmc = burn ? wnull : wm;   // an empty window drops the stores; the lane offsets stay loop-invariant
        if (span.state && span.seg > 0 && t0 == span.t_out) {            // the junction: the state the burn-in arrived at
            if (entry) span.state[D + i * D + j] = P;
            if (mean_lane) span.state[i] = mrow;
        }
        ybuf[lane] = ychunk;
        wave_lds_fence();
        double ynext = ybuf[0];
        for (int slot = 0; slot < nsteps; slot++) {
            const unsigned t = (unsigned)(t0 + slot);
            const double y = ynext;                                      // through LDS, a step ahead (a v_readlane pair costs 24 issue cycles)
            ynext = ybuf[slot + 1];
            // ---- distributed (Pf, mf) -> every lane: through LDS, read back as broadcasts
            pbuf[i * 8 + j] = P;
            pbuf[mslot] = mrow;                                          // the lanes without a mean entry write to their dump slot: no branch
            wave_lds_fence();
            Sym<D> Pr; Vec<D> m;
            CGP_UNROLL for (int a = 0; a < D; a++) {
                CGP_UNROLL for (int c = 0; c <= a; c += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(pbuf + a * 8 + c);
                    Pr(a, c) = v.x;
                    if (c + 1 <= a) Pr(a, c + 1) = v.y;
                }
            }
            CGP_UNROLL for (int a = 0; a < D; a += 2) {
                const double2 v = *reinterpret_cast<const double2*>(pbuf + 64 + a);
                m.v[a] = v.x; m.v[a + 1] = v.y;
            }
            wave_lds_fence();
            // ---- chol(Pf) = L sqrt(D'), replicated
            Sym<D> l; double dv[D]; bool bad;
            ldl_lower<D>(Pr, l, dv, bad);
            const double poison = bad ? __builtin_nan("") : 0.0;
            // ---- this lane's point: displacement d = L xi (xi_0..D-2), rotating pairs, propagated linear pair
            double xs[D - 1];
            if constexpr (AXIAL) {
                double dvx = fma(xi2[0], dv[0], xz);                     // xi_c^2 dv_c of the lane's axis c (1 for the point on the mean)
                CGP_UNROLL for (int c = 1; c < D - 1; c++) dvx = fma(xi2[c], dv[c], dvx);
                const double rt = sqrt_fast(dvx);
                CGP_UNROLL for (int c = 0; c < D - 1; c++) xs[c] = fma(xsg[c], rt, poison);
            } else {
                CGP_UNROLL for (int c = 0; c < D - 1; c++) xs[c] = fma(xi[c], sqrt_fast(dv[c]), poison);
            }
            double dd[D];
            CGP_UNROLL for (int a = 0; a < D; a++) {
                double s = (a <= D - 2) ? xs[a] : 0.0;                   // unit diagonal; xi_{D-1} does not take part
                CGP_UNROLL for (int c = 0; c < (a <= D - 2 ? a : D - 1); c++) s = fma(l(a, c), xs[c], s);
                dd[a] = s;
            }
            // rho cos / sin of k theta(chi_v), k = 1..NH: without regime branches (one basic block to schedule); a lane outside the
            // lean regime sends the wavefront through the branch-free ANY form (round 5, cgp_models.hpp: precompute_any) and only from
            // there through the checked one
            typename HarmonicLCD<NH>::Pre pre;
            bool ok;
            model.precompute_spec(R, m.v[V] + dd[V], pre, ok);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
                model.precompute_any(R, m.v[V] + dd[V], pre, ok);
                if (__builtin_amdgcn_ballot_w64(!ok) != 0) model.precompute(m.v[V] + dd[V], pre);
            }
            double gg[NL];
            CGP_UNROLL for (int k = 0; k < NH; k++) {
                const double h0 = m.v[2 * k] + dd[2 * k], h1 = m.v[2 * k + 1] + dd[2 * k + 1];
                gg[2 * k] = pre.c[k] * h0 - pre.s[k] * h1;
                gg[2 * k + 1] = pre.s[k] * h0 + pre.c[k] * h1;
            }
            const double e0 = fma(model.M[0], dd[V], model.M[1] * dd[V + 1]);
            const double e1 = fma(model.M[2], dd[V], model.M[3] * dd[V + 1]);
            // rows 4 X + q of G kept by this lane
            const double A0 = (q == 0) ? gg[0] : (q == 1) ? gg[1] : (q == 2) ? gg[2] : gg[3];
            double A1;
            if constexpr (NH == 3) A1 = (q == 0) ? gg[4] : (q == 1) ? gg[5] : (q == 2) ? e0 : e1;
            else A1 = (q == 0) ? e0 : (q == 1) ? e1 : 0.0;
            // ---- G W G^T by tiles, row sums in row and column form.  Block (I, J) wants tile (I, J) summed over the points
            // of ALL four blocks.  Instead of four all-reduced tiles and a select, the tile each partner needs is picked on
            // the OPERANDS -- "mine" (A_I, B_J) or "the partner's" (A_{I^1}, B_{J^1}) -- so the sum is a reduce-scatter:
            //     tile = [mine x mine] + xor2[other x mine] + xor1([mine x other] + xor2[other x other])
            const double AI = I ? A1 : A0, AIx = I ? A0 : A1;
            const double BJ = W * (J ? A1 : A0), BJx = W * (J ? A0 : A1);
            const double Tt = (mfma4x4(AI, BJ, 0.0) + blk_xor2(mfma4x4(AIx, BJ, 0.0)))
                            + blk_xor1(mfma4x4(AI, BJx, 0.0) + blk_xor2(mfma4x4(AIx, BJx, 0.0)));
            const double Rh = mfma4x4(AI, W, 0.0) + blk_xor2(mfma4x4(AIx, W, 0.0));
            const double S1r = Rh + blk_xor1(Rh);                                    // sum W G[4 I + r], every q
            const double S1c = mfma4x4(blk_swap12(S1r), IDENT, 0.0);                 // sum W G[4 J + q], every r: [r][q] <- [q][r] of row block J
            // ---- predicted moments in tile layout / row form
            const double Pp = fma(-S1r, S1c, Tt) + fma(dv[D - 1], K1, Sig);
            const double mp = S1r + (fma(Ma, m.v[V], Mb * m.v[V + 1]) + poison);
            // ---- update (filters_smoothers.py:55-68)
            double S, innov;
            coop8_update(Pp, mp, HR, HC, XiC, y, P, mrow, S, innov);
            park[slot] = make_double2(S, innov);
            wPc.store(P, t * (unsigned)(D * D * 8) + offP);
            wmc.store(mrow, t * (unsigned)(D * 8) + offm);
        }
        if (want_nll && !burn) {
            wave_lds_fence();
            const double2 si = park[lane < nsteps ? lane : 0];
            cum = nll_flush_wave(si.x, si.y, lane, nsteps, cum, nll ? nll + t0 : nullptr);
            wave_lds_fence();
        }
    }
    if (span.state) {                                                    // the segment's last state and its NLL total, for the fix-up pass
        if (entry) span.state[kN + D + i * D + j] = P;
        if (mean_lane) span.state[kN + i] = mrow;
        if (lane == 0) span.state[2 * kN] = cum;
    } else if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
}

// ekf8_coop_kernel: ekf (filters_smoothers.py:222-264) for the harmonic chirp LCD model with two or three harmonics, in
// the tile layout.  J = blockdiag(rho Rot(k theta), M32) + (d f / d u_v column) is held TRANSPOSED, one entry
// per lane (JT: lane (i, j) holds J[j][i]), assembled from the wave-uniform rotations with per-lane 0 / +-1 / constant
// coefficients; the predicted mean comes out of two matrix-vector MFMAs (row form for the update, column form for the
// Jacobian column, which is a quad swap of it, as in cgp_mfma4.hpp); and
//     W = P J^T,   Pp = J W + Sigma
// are two 8 x 8 x 8 products = four v_mfma_f64_4x4x4 with bank-masked DPP block moves.  The scalar chain softplus ->
// sincos of the frequency state (one v_readlane pair from its lane) is evaluated once, wave-uniformly.
template <int NH>
__global__ void __launch_bounds__(64) ekf8_coop_kernel(FilterIO io, ModelArgs ma) {
    constexpr int D = 2 * NH + 2, NL = 2 * NH, V = NL;
    static_assert(NH == 2 || NH == 3, "d = 6 and d = 8");
    __shared__ double2 park[64];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int I = b >> 1, J = b & 1;
    const int i = 4 * I + r, j = 4 * J + q;                              // JT: this lane holds J[j][i]; P: P[i][j]
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    HarmonicLCD<NH> model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const double HR = (i < D) ? Hp[i] : 0.0;
    const double HC = (4 * J + r < D) ? Hp[4 * J + r] : 0.0;
    const double Xi = io.Xi[trial * io.Xi_stride];
    const double XiC = (I == 0) ? Xi : 0.0;
    double Sig = 0.0;
    if (i == j && i < NL) Sig = model.q;
    else if (i == V && j == V) Sig = model.MS[0];
    else if ((i == V + 1 && j == V) || (i == V && j == V + 1)) Sig = model.MS[1];
    else if (i == V + 1 && j == V + 1) Sig = model.MS[2];
    // J0[row][col] with row = j, col = i (transposed holding): c on the diagonal of a rotation block, -s above, +s below it
    const int row = j, col = i;
    const int hk = (row < NL) ? row >> 1 : 0;                            // harmonic (0-based) of this lane's rotation block
    const bool rot = row < NL && col < NL && (row >> 1) == (col >> 1);
    const double kc = (rot && row == col) ? 1.0 : 0.0;
    const double ks = rot ? ((row == col + 1) ? 1.0 : (col == row + 1 ? -1.0 : 0.0)) : 0.0;
    double kk = 0.0;
    if (row >= V && row < D && col >= V && col < D) kk = model.M[2 * (row - V) + (col - V)];
    // d f_row / d u_v = (hk + 1) dtheta/du_v * (-f_{row+1} for even rows, +f_{row-1} for odd rows): a quad swap of f in column form
    const double kj = (col == V && row < NL) ? ((row & 1) ? 1.0 : -1.0) * (double)(hk + 1) : 0.0;
    const double ang = (model.dt * kTwoPi) * model.fs;
    const bool entry = i < D && j < D;
    const bool mean_lane = (J == 0 && q == 0 && i < D);
    constexpr int kVLane = 16 * (V & 3) + 4 * (2 * (V >> 2));            // a lane whose row-form entry is u_v
    FanRegs R;
    R.init();

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

    double cum = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        for (int slot = 0; slot < nsteps; slot++) {
            const unsigned t = (unsigned)(t0 + slot);
            const double y = readlane_f64(ychunk, slot);                 // (through LDS as in sgp8_coop_kernel: 5.26 against 5.00 ms here)
            // ---- wave-uniform scalar chain: rotations at the frequency g(u_v) (models.py:370-376)
            const double uv = readlane_f64(mrow, kVLane);
            // evaluated without regime branches (lean softplus on [1.5, 700), sin / cos on the reduced range |x| <= pi/4 in
            // Estrin form, coefficients pinned): one block to schedule; outside the regime the checked forms run afterwards
            const double et = exp_neg_lean(R, uv);
            double lq, dsp, s1, c1;
            softplus_tail_lean(R, et, lq, dsp);
            double sp = fma(lq, et, uv);
            sincos_reduced(R, ang * sp, s1, c1);
            const bool ok = softplus_lane_common(uv) && fabs(ang * sp) <= kPiOver4;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
                softplus_pair_uniform(uv, sp, dsp);
                fast_sincos_uniform(ang * sp, s1, c1);
            }
            double ck = c1, sk = s1, csel = c1, ssel = s1;
            CGP_UNROLL for (int k = 1; k < NH; k++) {
                const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
                ck = cn; sk = sn;
                csel = (hk == k) ? ck : csel; ssel = (hk == k) ? sk : ssel;
            }
            const double JT0 = fma(kc, csel * model.rho, fma(ks, ssel * model.rho, kk));        // J0[j][i]
            // ---- predicted mean: column form f[4 J + q] (for the Jacobian column), row form f[4 I + r] (for the update)
            double fc = mfma4x4(mrow, JT0, 0.0);                         // sum_k u[4 I + k] J0[4 J + q][4 I + k]
            fc += blk_xor2(fc);
            const double xc = blk_swap12(mrow);                          // u[4 J + r]
            double fr = mfma4x4(blk_swap12(JT0), xc, 0.0);               // sum_k J0[4 I + r][4 J + k] u[4 J + k]
            fr += blk_xor1(fr);
            const double JT = fma(kj * (ang * dsp), dpp_f64<kQuadSwap1>(fc), JT0);
            // ---- W = P J^T, Pp = J W + Sigma
            const double W = mfma4x4(blk_rows_of_k1(P), blk_cols_of_k1(JT), mfma4x4(blk_rows_of_k0(P), blk_cols_of_k0(JT), 0.0));
            const double Pp = mfma4x4(blk_rows_of_k1(JT), blk_cols_of_k1(W), mfma4x4(blk_rows_of_k0(JT), blk_cols_of_k0(W), Sig));
            // ---- update
            double S, innov;
            coop8_update(Pp, fr, HR, HC, XiC, y, P, mrow, S, innov);
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

template <int NH>
inline int launch_ekf8_coop(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (io.T * ((2 * NH + 2) * (2 * NH + 2) * 8) > kOobMaxBytes) return CGP_E_UNSUPPORTED;        // output windows (OobWindow)
    hipLaunchKernelGGL((ekf8_coop_kernel<NH>), dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}

// The collapsed quadrature of the kernel above needs the caller's CGP_SIGMA_STANDARD assertion, groups, and at most one
// group per (DPP row, block) pair.
inline bool coop8_sigma_ok(const ModelArgs& ma) {
    return (ma.sg.flags & CGP_SIGMA_STANDARD) && ma.sg.group_start && ma.sg.n_groups >= 1 && ma.sg.n_groups <= 16;
}

template <int NH>
inline int launch_sgp8_coop(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (io.T * ((2 * NH + 2) * (2 * NH + 2) * 8) > kOobMaxBytes) return CGP_E_UNSUPPORTED;        // output windows (OobWindow)
    const bool axial = (ma.sg.flags & CGP_SIGMA_AXIAL) != 0;
    const size_t lds = sigma_lds_bytes(ma, 2 * NH + 2);
    if (io.segs > 1) {                                                              // time-split: one wavefront per (trial, segment)
        const unsigned grid = (unsigned)(io.B * io.segs);
        if (axial) hipLaunchKernelGGL((sgp8_coop_kernel<NH, true, true>), dim3(grid), dim3(64), lds, stream, io, ma);
        else hipLaunchKernelGGL((sgp8_coop_kernel<NH, false, true>), dim3(grid), dim3(64), lds, stream, io, ma);
    } else if (axial) hipLaunchKernelGGL((sgp8_coop_kernel<NH, true, false>), dim3((unsigned)io.B), dim3(64), lds, stream, io, ma);
    else hipLaunchKernelGGL((sgp8_coop_kernel<NH, false, false>), dim3((unsigned)io.B), dim3(64), lds, stream, io, ma);
    return hip_rc(hipGetLastError());
}

// ====================================================================================================================
// Discrete smoothers (rts / eks / sgp_smoother, filters_smoothers.py:187-219, 317-349, 493-531) for 5 <= d <= 8.
//
// The backward recursion (filters_smoothers.py:83-84) is an affine map of the carry whose coefficients depend on the filtering
// results only:   ms_t = G_t ms_{t+1} + c_t,   Ps_t = G_t Ps_{t+1} G_t^T + C_t,   c_t = mf_t - G_t mp_t,   C_t = Pf_t - G_t Pp_t G_t^T.
// The time-parallel smoother of cgp_kernels.hpp composes these maps with a six-round suffix scan in which EVERY lane
// multiplies 8 x 8 matrices every round: 10 000 of its 25 000 wave-instructions per tile at d = 8, with two maps live per lane
// (register spills).  Here the lanes still do the expensive, independent part of their own steps in parallel -- prediction
// (model / sigma fan) at (mf, Pf), Cholesky of Pp, the gain G = (Pp^{-1} D^T)^T, then c and C: 64 chains, one per lane -- but
// nothing is composed: (G, C, c) of the steps go to LDS (108 doubles a step) and the wavefront walks the tile backwards
// COOPERATIVELY, carrying (ms, Ps) in the tile layout and applying the recursion on the matrix cores,
//     W = Ps' G^T,   Ps = G W + C,      ms = G ms' + c
// two 8 x 8 x 8 products = four v_mfma_f64_4x4x4 (the G operands come from LDS already arranged per block, the carry is
// re-arranged between blocks with bank-masked DPP moves).  Two record forms:
//   * one wavefront per trial (coop8_smoother_kernel): records (G, Pp, mp); the walk does X = Ps' - Pp, W = X G^T,
//     Ps = G W + Pf, ms = G (ms' - mp) + mf.  Round 2 read (Pf, mf) of a step a SECOND time from the input arrays (1.5 x the
//     algorithmic traffic); round 3: every lane parks its filtering row in LDS when it loads it for the gain (44 doubles a
//     step, 23 KB a tile; the records go in quarters of 16 to make room) and the walk takes it from there -- every row is read
//     from HBM ONCE;
//   * the affine form (coop8_split_kernel): records (G, C, c) with the step's constants folded by the lanes that build the
//     gains, so the walk reads nothing from HBM and has no subtraction on its chain -- every filtering row is read ONCE.  As
//     the whole-record kernel it measured SLOWER (BASELINE C5's smoother 5.11 against 4.75 ms, EKS 3.9 against 3.5: the 860
//     extra multiply-adds a lane-step for C = Pf - G Pp G^T cost more than the second read, which was hidden), so it serves
//     the time-split passes, where the maps must compose.
//
// Small batches: the time-split form (see cgp_walk4.hpp) -- pass 1 (kWalkCompose) walks every segment with the carry (0, 0)
// and one more product a step, A <- G A, and leaves the segment's composed map (A, C, c) in the workspace in the layout of a
// step record; pass 2 (kWalkApply) applies the later segments' maps to the record's last filtering row (one "walk step"
// each, operands straight from the workspace) and walks its segment.
constexpr int kElemDoubles = 109;                 // G (8 x 8, pitch 8) | C (packed lower, 36) | c (8) | one zero; odd: conflict-free lane stride
constexpr int kElemC = 64, kElemc = 100, kElemZero = 108;
constexpr int kMap8Doubles = 112;                 // a segment's map in the workspace: the same layout, padded to 16-byte multiples
enum { kWalkWhole = 0, kWalkCompose = 1, kWalkApply = 2 };

// Wavefronts per trial for the time-split form: as many as keep every workgroup resident at once (what the occupancy query
// says a CU holds of the kernel: registers, the LDS records plus the staged sigma-point set), each with at least
// io.min_tiles tiles; 1 = one wave per trial.
inline int walk_segments(const SmootherIO& io, int blocks_per_cu) {
    if (io.segs == 1) return 1;
    const int64_t tiles = (io.T - 1 + 63) / 64;
    int64_t per_cu = blocks_per_cu > 0 ? blocks_per_cu : 4;
    if (per_cu > 8) per_cu = 8;
    int64_t segs = ((int64_t)io.num_cus * per_cu) / (io.B > 0 ? io.B : 1);
    if (io.segs > 1 && segs > io.segs) segs = io.segs;
    const int64_t min_tiles = io.min_tiles > 0 ? io.min_tiles : 1;
    if (segs > tiles / min_tiles) segs = tiles / min_tiles;
    if (segs > 64) segs = 64;
    return segs < 2 ? 1 : (int)segs;
}

// ---- one wavefront per trial: records (G, Pp, mp) in quarters of 16, the tile's filtering rows (Pf, mf) parked beside them
constexpr int kRowDoubles = 45;                   // Pf (packed lower, 36) | mf (8) | one zero; odd: conflict-free lane stride
constexpr int kRowmf = 36, kRowZero = 44;
struct Elem8WalkOperands { double gA0, gA1, gB0, gB1, gM, Ppv, mpc, Pfv, mfr; };

// SEL (cgp_smoother_select): the lanes that hold the selected component's smoothed mean and variance leave them in LDS step by step;
// behind a tile's walk every lane finishes ITS step (the marginal itself, E[f(V)] by 1-D Gauss-Hermite) -- one coalesced 512-byte
// store per output and tile.  mss / Pss may be NULL then (a window of zero bytes drops their stores).
template <class Elem, bool SEL = false>
__global__ void __launch_bounds__(64) coop8_smoother_kernel(SmootherIO io, ModelArgs ma) {      // one wavefront per trial, the whole record
    constexpr int D = Elem::D;
    static_assert(D >= 5 && D <= 8, "tile layout of an 8 x 8 matrix");
    __shared__ double elems[16 * kElemDoubles];      // (G, Pp, mp) of a quarter tile: 13.9 KB
    __shared__ double rows[64 * kRowDoubles];        // (Pf, mf) of the whole tile: 23.0 KB -- together 36.9 KB: four workgroups a CU
    __shared__ double selbuf[SEL ? 128 : 1];
    __shared__ double ghrule[SEL ? 2 * kGhMaxOrder : 1];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int I = b >> 1, J = b & 1;
    const int i = 4 * I + r, j = 4 * J + q;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    Elem elem;
    elem.setup(ma, trial);
    for (int k = lane; k < 16 * kElemDoubles; k += 64) elems[k] = 0.0;      // pads of G / Pp / Pf and the zero slots stay zero
    for (int k = lane; k < 64 * kRowDoubles; k += 64) rows[k] = 0.0;
    if constexpr (SEL) sel_stage_rule(io.sel, ghrule, lane);
    if constexpr (Elem::USES_SIGMA) elem.sg.stage(dyn_lds(), lane, 64, D); else __syncthreads();
    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * D;
    const double* __restrict__ Pfs = io.Pfs + trial * T * D * D;
    double* __restrict__ mss = io.mss + trial * T * D;
    double* __restrict__ Pss = io.Pss + trial * T * D * D;
    const bool entry = i < D && j < D;
    const bool mean_lane = (J == 0 && q == 0 && i < D);

    // per-lane offsets (doubles) of the operands of one step inside its LDS record and inside its parked row
    const int oA = (4 * I + q) * 8 + r;              // G[4 I + q'][4 K + r'] at + 4 K
    const int oB = (4 * J + q) * 8 + r;              // G[4 J + q'][4 K + r'] at + 4 K
    const int oM = (4 * I + q) * 8 + 4 * J + r;      // G[4 I + q'][4 J + r']
    const int oP = entry ? kElemC + Sym<8>::idx(i, j) : kElemZero;
    const int om = (4 * J + r < D) ? kElemc + 4 * J + r : kElemZero;          // mp[4 J + r]
    const int oPf = entry ? Sym<8>::idx(i, j) : kRowZero;                      // Pf[i][j] (from its lower triangle, like the other kernels)
    const int omf = (i < D) ? kRowmf + i : kRowZero;                           // mf[4 I + r]
    // The output rows leave through buffer windows with per-lane byte offsets: a lane that has nothing to write carries an
    // out-of-range offset (dropped), so no store sits behind an exec-mask branch.
    const unsigned bS = entry ? 8u * (unsigned)(i * D + j) : kOobOffset;
    const unsigned bms = mean_lane ? 8u * (unsigned)i : kOobOffset;
    OobWindow wPs, wms;
    wPs.init(io.Pss ? Pss : nullptr, T * D * D * 8); wms.init(io.mss ? mss : nullptr, T * D * 8);

    // carry: Ps in tile layout, ms with lane (r, (I, J), q) holding ms[4 J + r]; last row copied verbatim (filters_smoothers.py:140-142)
    double Ps = entry ? Pfs[(T - 1) * D * D + ((i >= j) ? i * D + j : j * D + i)] : 0.0;
    double xc = (4 * J + r < D) ? mfs[(T - 1) * D + 4 * J + r] : 0.0;
    if (entry && (!SEL || io.Pss)) Pss[(T - 1) * D * D + i * D + j] = Pfs[(T - 1) * D * D + i * D + j];
    if (mean_lane && (!SEL || io.mss)) mss[(T - 1) * D + i] = mfs[(T - 1) * D + i];
    if constexpr (SEL) { if (lane == 0) sel_write(io.sel, ghrule, trial * T + T - 1, mfs[(T - 1) * D + io.sel.comp], Pfs[(T - 1) * D * D + io.sel.comp * (D + 1)]); }
    const bool sel_var_lane = SEL && entry && i == io.sel.comp && j == io.sel.comp, sel_mean_lane = SEL && mean_lane && i == io.sel.comp;

    for (int64_t hi = T - 2; hi >= 0; hi -= 64) {
        const int64_t base = hi - 63;                                      // step of lane 0 (may be negative in the last tile)
        // ---- every lane: its filtering row -- parked in LDS for the walk, which round 2 had read a SECOND time from HBM
        // (1.5 x the algorithmic traffic of BASELINE C5's smoother) -- then prediction and gain of its own step
        const int64_t mystep = base + lane;
        Mat<D> G; Vec<D> mp; Sym<D> Pp;
        {
            double* mine = rows + lane * kRowDoubles;
            if (mystep >= 0) {
                Vec<D> mf; Sym<D> Pf;
                load_vec<D>(mfs + mystep * D, mf);
                load_sym<D>(Pfs + mystep * D * D, Pf);
                CGP_UNROLL for (int a = 0; a < D; a++) CGP_UNROLL for (int c = 0; c <= a; c++) mine[Sym<8>::idx(a, c)] = Pf(a, c);
                CGP_UNROLL for (int a = 0; a < D; a++) mine[kRowmf + a] = mf.v[a];
                elem.gain(mf, Pf, G, mp, Pp);
            } else {                                                       // before the start of the record: a zero row and a zero record
                CGP_UNROLL for (int a = 0; a < D; a++) CGP_UNROLL for (int c = 0; c <= a; c++) mine[Sym<8>::idx(a, c)] = 0.0;
                CGP_UNROLL for (int a = 0; a < D; a++) { mine[kRowmf + a] = 0.0; mp.v[a] = 0.0; CGP_UNROLL for (int c = 0; c < D; c++) G.a[a][c] = 0.0; }
                CGP_UNROLL for (int a = 0; a < Sym<D>::N; a++) Pp.a[a] = 0.0;
            }
        }
        // The records go to LDS in four quarters of 16 (13.9 KB): lanes 48..63, the latest steps, first; the other lanes keep
        // theirs in registers until their quarter has its turn.
        CGP_UNROLL for (int quarter = 3; quarter >= 0; quarter--) {
            if ((lane >> 4) == quarter) {
                double* mine = elems + (lane & 15) * kElemDoubles;
                CGP_UNROLL for (int a = 0; a < D; a++) CGP_UNROLL for (int c = 0; c < D; c++) mine[a * 8 + c] = G.a[a][c];
                CGP_UNROLL for (int a = 0; a < D; a++) CGP_UNROLL for (int c = 0; c <= a; c++) mine[kElemC + Sym<8>::idx(a, c)] = Pp(a, c);
                CGP_UNROLL for (int a = 0; a < D; a++) mine[kElemc + a] = mp.v[a];
            }
            wave_lds_fence();
            // ---- the wavefront walks the quarter from its last step to its first, the LDS operands one step ahead.  Steps
            // before the start of the record (last tile) are walked too: their records and rows are zero and their stores
            // fall outside the windows (the step index wraps).
            auto fetch = [&](int s, Elem8WalkOperands& o) {
                const double* p = elems + (s & 15) * kElemDoubles;
                const double* w = rows + (s & 63) * kRowDoubles;
                o.gA0 = p[oA]; o.gA1 = p[oA + 4]; o.gB0 = p[oB]; o.gB1 = p[oB + 4]; o.gM = p[oM]; o.Ppv = p[oP]; o.mpc = p[om];
                o.Pfv = w[oPf]; o.mfr = w[omf];
            };
            Elem8WalkOperands cur, nxt;
            fetch(16 * quarter + 15, cur);
#pragma unroll 8
            for (int u = 0; u < 16; u++) {
                const int s = 16 * quarter + 15 - u;
                fetch(16 * quarter + ((s - 1) & 15), nxt);                 // (the record fetched after the quarter's last step is not used)
                // W = (Ps' - Pp) G^T
                const double X = Ps - cur.Ppv;
                const double W = mfma4x4(blk_rows_of_k1(X), cur.gB1, mfma4x4(blk_rows_of_k0(X), cur.gB0, 0.0));
                // mean: ym[4 I + r] = sum_J sum_k G[4 I + r][4 J + k] (ms' - mp)[4 J + k] + mf
                double ym = mfma4x4(cur.gM, xc - cur.mpc, 0.0);
                ym = (ym + blk_xor1(ym)) + cur.mfr;
                // Ps = G W + Pf
                Ps = mfma4x4(cur.gA1, blk_cols_of_k1(W), mfma4x4(cur.gA0, blk_cols_of_k0(W), cur.Pfv));
                xc = blk_swap12(ym);
                const unsigned step = (unsigned)(base + s);
                wPs.store(Ps, bS + step * (unsigned)(D * D * 8));
                wms.store(ym, bms + step * (unsigned)(D * 8));
                if constexpr (SEL) {
                    if (sel_var_lane) selbuf[64 + s] = Ps;
                    if (sel_mean_lane) selbuf[s] = ym;
                }
                cur = nxt;
            }
            wave_lds_fence();
        }
        if constexpr (SEL) {
            if (mystep >= 0) sel_write(io.sel, ghrule, trial * T + mystep, selbuf[lane], selbuf[64 + lane]);
            wave_lds_fence();
        }
    }
}

// ---- the affine form: records (G, C, c); the time-split passes, and the whole record with CGP_TIME_SPLIT-style maps
struct Elem8Operands { double gA0, gA1, gB0, gB1, gM, Cv, cr; };

template <class Elem, int MODE, bool SEL = false>
__global__ void __launch_bounds__(64) coop8_split_kernel(SmootherIO io, ModelArgs ma) {
    constexpr int D = Elem::D;
    static_assert(D >= 5 && D <= 8, "tile layout of an 8 x 8 matrix");
    static_assert(!(SEL && MODE == kWalkCompose), "pass 1 of the time-split form writes nothing");
    __shared__ double elems[32 * kElemDoubles];
    __shared__ double selbuf[SEL ? 128 : 1];
    __shared__ double ghrule[SEL ? 2 * kGhMaxOrder : 1];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int I = b >> 1, J = b & 1;
    const int i = 4 * I + r, j = 4 * J + q;
    const int64_t trial = (MODE == kWalkWhole) ? (int64_t)blockIdx.x : (int64_t)(blockIdx.x / (unsigned)io.segs);
    const int seg = (MODE == kWalkWhole) ? 0 : (int)(blockIdx.x % (unsigned)io.segs);
    if (trial >= io.B) return;

    Elem elem;
    elem.setup(ma, trial);
    for (int k = lane; k < 32 * kElemDoubles; k += 64) elems[k] = 0.0;      // pads of G / C and the zero slot stay zero
    if constexpr (SEL) sel_stage_rule(io.sel, ghrule, lane);
    if constexpr (Elem::USES_SIGMA) elem.sg.stage(dyn_lds(), lane, 64, D); else __syncthreads();
    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * D;
    const double* __restrict__ Pfs = io.Pfs + trial * T * D * D;
    double* __restrict__ mss = io.mss + trial * T * D;
    double* __restrict__ Pss = io.Pss + trial * T * D * D;
    const bool entry = i < D && j < D;
    const bool mean_lane = (J == 0 && q == 0 && i < D);
    // the segment's tiles: tile n covers the steps T - 2 - 64 n - 63 .. T - 2 - 64 n (segment 0 is the LAST in time)
    const int64_t hi_first = (MODE == kWalkWhole) ? T - 2 : T - 2 - 64 * (int64_t)seg * io.tiles_per_seg;
    const int64_t hi_stop = (MODE == kWalkWhole) ? 0 : max((int64_t)0, hi_first - 64 * (int64_t)io.tiles_per_seg + 1);
    if (hi_first < 0 && MODE != kWalkWhole) return;

    // per-lane offsets (doubles) of the operands of one step inside its record (LDS, or a segment's map in the workspace)
    const int oA = (4 * I + q) * 8 + r;              // G[4 I + q'][4 K + r'] at + 4 K
    const int oB = (4 * J + q) * 8 + r;              // G[4 J + q'][4 K + r'] at + 4 K
    const int oM = (4 * I + q) * 8 + 4 * J + r;      // G[4 I + q'][4 J + r']
    const int oC = entry ? kElemC + Sym<8>::idx(i, j) : kElemZero;
    const int oc = (i < D) ? kElemc + i : kElemZero;                          // c[4 I + r]
    // The walk addresses the output rows through buffer windows with per-lane byte offsets: a lane that has nothing to write
    // carries an out-of-range offset (dropped), so no store sits behind an exec-mask branch.
    const unsigned bS = entry ? 8u * (unsigned)(i * D + j) : kOobOffset;
    const unsigned bms = mean_lane ? 8u * (unsigned)i : kOobOffset;
    OobWindow wPs, wms;
    wPs.init((MODE == kWalkCompose || !io.Pss) ? nullptr : Pss, T * D * D * 8); wms.init((MODE == kWalkCompose || !io.mss) ? nullptr : mss, T * D * 8);

    // carry: Ps in tile layout, ms with lane (r, (I, J), q) holding ms[4 J + r]; last row copied verbatim (filters_smoothers.py:140-142)
    double Ps = entry ? Pfs[(T - 1) * D * D + ((i >= j) ? i * D + j : j * D + i)] : 0.0;
    double xc = (4 * J + r < D) ? mfs[(T - 1) * D + 4 * J + r] : 0.0;
    double Acc = (entry && i == j) ? 1.0 : 0.0;      // pass 1: the composed linear part, A <- G A
    if constexpr (MODE == kWalkCompose) { Ps = 0.0; xc = 0.0; }
    if (MODE != kWalkCompose && seg == 0) {
        if (entry && (!SEL || io.Pss)) Pss[(T - 1) * D * D + i * D + j] = Pfs[(T - 1) * D * D + i * D + j];
        if (mean_lane && (!SEL || io.mss)) mss[(T - 1) * D + i] = mfs[(T - 1) * D + i];
        if constexpr (SEL) { if (lane == 0) sel_write(io.sel, ghrule, trial * T + T - 1, mfs[(T - 1) * D + io.sel.comp], Pfs[(T - 1) * D * D + io.sel.comp * (D + 1)]); }
    }
    const bool sel_var_lane = SEL && entry && i == io.sel.comp && j == io.sel.comp, sel_mean_lane = SEL && mean_lane && i == io.sel.comp;
    // one step of the recursion with the operands o: W = Ps' G^T, Ps = G W + C, ms = G ms' + c
    auto step = [&](const Elem8Operands& o, double& ym) {
        const double W = mfma4x4(blk_rows_of_k1(Ps), o.gB1, mfma4x4(blk_rows_of_k0(Ps), o.gB0, 0.0));
        // mean: ym[4 I + r] = sum_J sum_k G[4 I + r][4 J + k] ms'[4 J + k] + c
        ym = mfma4x4(o.gM, xc, 0.0);
        ym = (ym + blk_xor1(ym)) + o.cr;
        if constexpr (MODE == kWalkCompose) Acc = mfma4x4(o.gA1, blk_cols_of_k1(Acc), mfma4x4(o.gA0, blk_cols_of_k0(Acc), 0.0));      // G A
        Ps = mfma4x4(o.gA1, blk_cols_of_k1(W), mfma4x4(o.gA0, blk_cols_of_k0(W), o.Cv));
        xc = blk_swap12(ym);
    };
    if constexpr (MODE == kWalkApply) {
        // carry-in of this segment: the maps of the segments later in time, applied in order to the last filtering row
        const double* __restrict__ maps = io.ws + trial * io.segs * kMap8Doubles;
        for (int s2 = 0; s2 < seg; s2++) {
            const double* __restrict__ p = maps + s2 * kMap8Doubles;
            Elem8Operands o;
            o.gA0 = p[oA]; o.gA1 = p[oA + 4]; o.gB0 = p[oB]; o.gB1 = p[oB + 4]; o.gM = p[oM]; o.Cv = p[oC]; o.cr = p[oc];
            double ym;
            step(o, ym);
        }
    }

    for (int64_t hi = hi_first; hi >= hi_stop; hi -= 64) {
        const int64_t base = hi - 63;                                      // step of lane 0 (may be negative in the last tile)
        // ---- every lane: the map of its own step
        const int64_t mystep = base + lane;
        Mat<D> G; Vec<D> c; Sym<D> C;
        if (mystep >= 0) {
            Vec<D> mf; Sym<D> Pf;
            load_vec<D>(mfs + mystep * D, mf);
            load_sym<D>(Pfs + mystep * D * D, Pf);
            elem.map(mf, Pf, G, c, C);
        } else {
            CGP_UNROLL for (int a = 0; a < D; a++) { c.v[a] = 0.0; CGP_UNROLL for (int k = 0; k < D; k++) G.a[a][k] = 0.0; }
            CGP_UNROLL for (int a = 0; a < Sym<D>::N; a++) C.a[a] = 0.0;
        }
        // The records go to LDS in two halves of 32 (27.9 KB: five workgroups fit a CU; all 64 at once would be 55.8 KB, two
        // workgroups per CU, and a batch of 1000 trials would run in two rounds): lanes 32..63, the later steps, first; lanes
        // 0..31 keep theirs in registers until the first half has been walked.
        CGP_UNROLL for (int half = 1; half >= 0; half--) {
            if ((lane >> 5) == half) {
                double* mine = elems + (lane & 31) * kElemDoubles;
                CGP_UNROLL for (int a = 0; a < D; a++) CGP_UNROLL for (int k = 0; k < D; k++) mine[a * 8 + k] = G.a[a][k];
                CGP_UNROLL for (int a = 0; a < D; a++) CGP_UNROLL for (int k = 0; k <= a; k++) mine[kElemC + Sym<8>::idx(a, k)] = C(a, k);
                CGP_UNROLL for (int a = 0; a < D; a++) mine[kElemc + a] = c.v[a];
            }
            wave_lds_fence();
            // ---- the wavefront walks the half from its last step to its first; the LDS operands are fetched one step ahead.
            // Steps before the start of the record (last tile) are walked too: their records are zero and their stores fall
            // outside the windows (the step index wraps).
            auto fetch = [&](int s, Elem8Operands& o) {
                const double* p = elems + (s & 31) * kElemDoubles;
                o.gA0 = p[oA]; o.gA1 = p[oA + 4]; o.gB0 = p[oB]; o.gB1 = p[oB + 4]; o.gM = p[oM]; o.Cv = p[oC]; o.cr = p[oc];
            };
            Elem8Operands cur, nxt;
            fetch(31, cur);
#pragma unroll 8
            for (int u = 0; u < 32; u++) {
                const int s = 32 * half + 31 - u;
                fetch((s - 1) & 31, nxt);                                  // (the record fetched after the half's last step is not used)
                double ym;
                step(cur, ym);
                if constexpr (MODE != kWalkCompose) {
                    const unsigned st = (unsigned)(base + s);
                    wPs.store(Ps, bS + st * (unsigned)(D * D * 8));
                    wms.store(ym, bms + st * (unsigned)(D * 8));
                    if constexpr (SEL) {
                        if (sel_var_lane) selbuf[64 + s] = Ps;
                        if (sel_mean_lane) selbuf[s] = ym;
                    }
                }
                cur = nxt;
            }
            wave_lds_fence();
        }
        if constexpr (SEL) {
            if (mystep >= 0) sel_write(io.sel, ghrule, trial * T + mystep, selbuf[lane], selbuf[64 + lane]);
            wave_lds_fence();
        }
    }
    if constexpr (MODE == kWalkCompose) {
        // the segment's map in the layout of a step record: A (pitch 8), C (packed lower), c
        double* __restrict__ out = io.ws + (trial * io.segs + seg) * kMap8Doubles;
        if (entry) {
            out[i * 8 + j] = Acc;
            if (i >= j) out[kElemC + Sym<8>::idx(i, j)] = Ps;
        } else out[(4 * I + r) * 8 + 4 * J + q] = 0.0;                     // the pads of A (d < 8)
        if (I == 0 && q == 0 && 4 * J + r < D) out[kElemc + 4 * J + r] = xc;
        if (lane < kMap8Doubles - kElemc && lane >= D) out[kElemc + lane] = 0.0;      // c's pads and the zero slot
        if (lane < 36) { int a_ = 0, k_ = lane; while (k_ > a_) { k_ -= a_ + 1; a_++; } if (a_ >= D) out[kElemC + lane] = 0.0; }   // C's pads (d < 8)
    }
}

template <class Elem>
inline hipError_t launch_coop8_smoother(const SmootherIO& io_in, const ModelArgs& ma, hipStream_t stream) {
    if (io_in.B <= 0 || io_in.T <= 0) return hipSuccess;
    if (io_in.T * Elem::D * Elem::D * 8 > kOobMaxBytes) return hipErrorInvalidValue;      // 2 GiB buffer windows (callers check coop8_smoother_ok)
    const size_t dyn = Elem::USES_SIGMA ? sigma_lds_bytes(ma, Elem::D) : 0;
    // workgroups of the split kernel a CU holds (registers, LDS): asked once per kernel and dynamic-LDS size, then remembered
    static std::atomic<long long> occ_cache{-1};                           // (dyn << 8) | per_cu
    int per_cu = 0;
    if (io_in.segs != 1) {
        const long long seen = occ_cache.load(std::memory_order_relaxed);
        if (seen >= 0 && (size_t)(seen >> 8) == dyn) per_cu = (int)(seen & 0xFF);
        else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, coop8_split_kernel<Elem, kWalkApply>, 64, dyn) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
            occ_cache.store(((long long)dyn << 8) | (per_cu & 0xFF), std::memory_order_relaxed);
        }
    }
    const int segs = walk_segments(io_in, per_cu);
    const bool sel = io_in.sel.comp >= 0;
    auto whole = [&]() {
        if (sel) hipLaunchKernelGGL((coop8_smoother_kernel<Elem, true>), dim3((unsigned)io_in.B), dim3(64), dyn, stream, io_in, ma);
        else hipLaunchKernelGGL((coop8_smoother_kernel<Elem>), dim3((unsigned)io_in.B), dim3(64), dyn, stream, io_in, ma);
        return hipGetLastError();
    };
    if (segs <= 1) return whole();
    // time-split: two passes with the segments' maps in the context's per-stream workspace (nothing the caller sees)
    SmootherIO io = io_in;
    io.segs = segs;
    const int64_t tiles = (io.T - 1 + 63) / 64;
    io.tiles_per_seg = (int)((tiles + io.segs - 1) / io.segs);
    io.segs = (int)((tiles + io.tiles_per_seg - 1) / io.tiles_per_seg);           // no empty segments
    void* ws = ctx_workspace(io.host_ctx, stream, sizeof(double) * kMap8Doubles * (size_t)io.B * io.segs);
    hipError_t e;
    if (!ws) return whole();             // no workspace (allocation failed, pinned too small, growth inside a graph capture): the one-wave-per-trial form needs none
    io.ws = (double*)ws;
    const unsigned grid = (unsigned)(io.B * io.segs);
    hipLaunchKernelGGL((coop8_split_kernel<Elem, kWalkCompose>), dim3(grid), dim3(64), dyn, stream, io, ma);
    if (sel) hipLaunchKernelGGL((coop8_split_kernel<Elem, kWalkApply, true>), dim3(grid), dim3(64), dyn, stream, io, ma);
    else hipLaunchKernelGGL((coop8_split_kernel<Elem, kWalkApply>), dim3(grid), dim3(64), dyn, stream, io, ma);
    e = hipGetLastError();
    return e;
}

}  // namespace cgp
