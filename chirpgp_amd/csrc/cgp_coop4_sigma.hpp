// cgp_coop4_sigma.hpp -- lane-cooperative d = 4 sigma-point kernels (BASELINE configs C3 and C4):
//     sgp4_coop_kernel      sgp_filter       on the chirp / La Scala LCD model      (filters_smoothers.py:446-490)
//     cdsgp4_coop_kernel    cd_sgp_filter    on the chirp / La Scala SDE model      (filters_smoothers.py:534-582)
//     cdsgps4_coop_kernel   cd_sgp_smoother  on the same                            (filters_smoothers.py:585-632)
//
// One wavefront per trial.  Three layouts coexist in a step:
//   * the covariance is DISTRIBUTED, lane (i, j) of every 16-lane row owning P[i][j], for everything that is matrix
//     algebra (RK4 bookkeeping, G^T P + P G, the Kalman update of cgp_coop4.hpp): one instruction instead of ten;
//   * the sigma-point FAN runs one group of points per lane (the points of a group share the nonlinear coordinate chi_v,
//     cgp_steps.hpp); it needs the Cholesky factor in every lane, so the 10 covariance entries are gathered with
//     v_readlane and the 4 x 4 Cholesky is replicated;
//   * the fan's partial sums are reduced through LDS (cgp_steps.hpp:wave_allreduce's scheme) and LEFT there: each lane
//     then reads exactly the totals it owns -- its own second-moment entry, the means -- so the "reduce-scatter" back
//     to the distributed layout costs one ds_read per lane.
// The sigma-point set is staged in dynamic LDS by the launch (cgp_kernels.hpp:dyn_lds).
#pragma once
#include "cgp_coop4.hpp"

namespace cgp {

// Lower triangle of the distributed covariance -> replicated packed matrix (10 wave-uniform values).
CGP_DEV void coop4_gather(double P, Sym<4>& S) {
    CGP_UNROLL for (int i = 0; i < 4; i++)
        CGP_UNROLL for (int j = 0; j <= i; j++) S(i, j) = readlane_f64(P, 4 * i + j);
}

// Sum of 16 consecutive doubles (eight 16-byte reads) as a balanced tree: four independent chains instead of one
// 16-deep chain of dependent adds (8 cycles each on the serial path of a step).
CGP_DEV double row_sum16(const double2* row) {
    double2 a = row[0], b = row[1], c = row[2], d = row[3];
    const double2 e = row[4], f = row[5], g = row[6], h = row[7];
    a.x += e.x; a.y += e.y; b.x += f.x; b.y += f.y; c.x += g.x; c.y += g.y; d.x += h.x; d.y += h.y;
    a.x += c.x; a.y += c.y; b.x += d.x; b.y += d.y;
    a.x += b.x; a.y += b.y;
    return a.x + a.y;
}

// Wave sum of R <= 32 per-lane partials; the totals stay in LDS at the returned pointer (tot[0..R-1]).
// Same fixed summation order as wave_allreduce.  When only the first 32 lanes carry partials (`narrow`, wave-uniform:
// at most 32 groups of sigma points, e.g. the 27 groups of Gauss-Hermite order 3 in d = 4) two lanes share a row --
// lane (r, h) = (lane >> 1, lane & 1) sums entries [16 h, 16 h + 16) -- and all R <= 32 rows are done in ONE pass.
template <int R>
CGP_DEV const double* coop_reduce_to_lds(double (&acc)[R], double* lds, int lane, bool narrow) {
    static_assert(R <= kRedChunk, "totals area holds kRedChunk doubles");
    double* tot = lds + kRedChunk * kRedLd;
    if (narrow) {
        CGP_UNROLL for (int k = 0; k < R; k++) lds[k * kRedLd + lane] = acc[k];
        wave_lds_fence();
        const int r = lane >> 1, h = lane & 1;
        const double2* row = reinterpret_cast<const double2*>(lds + r * kRedLd + h * 16);
        double s = 0.0;
        if (r < R) {
            s = row_sum16(row);
        }
        s += dpp_f64<kQuadSwap1>(s);
        if (h == 0 && r < R) tot[r] = s;
        wave_lds_fence();
        return tot;
    }
    const int r = lane >> 2, q = lane & 3;
    CGP_UNROLL for (int base = 0; base < R; base += kRedPass) {
        CGP_UNROLL for (int k = 0; k < kRedPass; k++)
            if (base + k < R) lds[k * kRedLd + lane] = acc[base + k];
        wave_lds_fence();
        const double2* row = reinterpret_cast<const double2*>(lds + r * kRedLd + q * 16);
        double s = 0.0;
        if (base + r < R) {
            s = row_sum16(row);
        }
        s += dpp_f64<kQuadSwap1>(s);
        s += dpp_f64<kQuadSwap2>(s);
        if (q == 0 && base + r < R) tot[base + r] = s;
        wave_lds_fence();
    }
    return tot;
}

// The sigma points a lane owns never change (lane g always evaluates group g), so they are read from LDS once into
// registers: up to kCacheM members of (xi[4], w).  `ok` is wave-uniform: it is false when the set has more than 64
// groups or a group has more members than the cache holds, and the kernels then read the staged set every step.
constexpr int kCacheM = 3;
struct LanePoints {
    double xi[kCacheM][4], w[kCacheM];
    int n;
    bool ok;
    CGP_DEV void load(const SigmaSet& sg, int lane) {
        const int ng = sg.groups();
        n = 0;
        int longest = 0;
        CGP_UNROLL for (int k = 0; k < kCacheM; k++) { w[k] = 0.0; CGP_UNROLL for (int j = 0; j < 4; j++) xi[k][j] = 0.0; }
        if (lane < ng) {
            const int p0 = sg.template begin<true>(lane), p1 = sg.template end<true>(lane);
            longest = p1 - p0;
            n = longest < kCacheM ? longest : kCacheM;
            CGP_UNROLL for (int k = 0; k < kCacheM; k++)
                if (k < n) {
                    w[k] = sg.template weight<true>(p0 + k);
                    CGP_UNROLL for (int j = 0; j < 4; j++) xi[k][j] = sg.template coord<true>((p0 + k) * 4 + j);
                }
        }
        CGP_UNROLL for (int delta = 1; delta < 64; delta *= 2) {
            const int o = __shfl_xor(longest, delta, 64);
            longest = o > longest ? o : longest;
        }
        ok = ng <= 64 && longest <= kCacheM;
    }
    // chi = m + L xi[k]
    CGP_DEV void point(int k, const Vec<4>& m, const Sym<4>& L, Vec<4>& chi) const {
        CGP_UNROLL for (int i = 0; i < 4; i++) {
            double t = L(i, 0) * xi[k][0];
            CGP_UNROLL for (int j = 1; j <= i; j++) t = fma(L(i, j), xi[k][j], t);
            chi.v[i] = m.v[i] + t;
        }
    }
};

// The representative of the group a lane owns in the collapsed quadratures: xi_0..2 of its first member and the group's
// total weight (zero for lanes without a group).
struct CollapsedPoint {
    double xi0, xi1, xi2, W;
    CGP_DEV void load(const SigmaSet& sg, int lane) {
        xi0 = xi1 = xi2 = W = 0.0;
        if (lane < sg.groups()) {
            const int p0 = sg.template begin<true>(lane), p1 = sg.template end<true>(lane);
            xi0 = sg.template coord<true>(p0 * 4); xi1 = sg.template coord<true>(p0 * 4 + 1); xi2 = sg.template coord<true>(p0 * 4 + 2);
            for (int p = p0; p < p1; p++) W += sg.template weight<true>(p);
        }
    }
};
// Shared prologue of the filter kernels: outputs, measurement chunking and NLL latch live in the kernels themselves.
struct Coop4FilterOut {
    double* __restrict__ mfs; double* __restrict__ Pfs; double* __restrict__ nll;
    bool nll_final, want_nll;
    CGP_DEV void init(const FilterIO& io, int64_t trial) {
        const int64_t T = io.T;
        mfs = io.mfs ? io.mfs + trial * T * 4 : nullptr;
        Pfs = io.Pfs ? io.Pfs + trial * T * 16 : nullptr;
        nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
        nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
        want_nll = io.nll != nullptr;
    }
    CGP_DEV void store(int64_t t, int lane, double P, double u0, double u1, double u2, double u3) const {
        if (Pfs && lane < 16) Pfs[t * 16 + lane] = P;
        if (mfs && lane == 0) {
            *reinterpret_cast<double2*>(mfs + t * 4) = make_double2(u0, u1);
            *reinterpret_cast<double2*>(mfs + t * 4 + 2) = make_double2(u2, u3);
        }
    }
};

// ------------------------------------------------------------------------------------------------ sgp_filter, d = 4
// COLLAPSED (CGP_SIGMA_STANDARD sets with at most 32 groups): the quadrature of cgp_steps.hpp:sgp4_prediction_collapsed
// with one group per lane -- 9 partial sums of ONE point instead of 15 sums over the group's members; the linear
// components' moments are closed forms evaluated from the gathered covariance.
template <class DM, bool COLLAPSED>
__global__ void __launch_bounds__(64) sgp4_coop_kernel(FilterIO io, ModelArgs ma) {
    static_assert(DM::D == 4, "d = 4 kernel");
    __shared__ double red[kFanLdsDoubles];
    const int lane = threadIdx.x;
    const int li = (lane >> 2) & 3, lj = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    DM model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    model.wide = true;
    SigmaSet sg = ma.sg;
    sg.stage(dyn_lds(), lane, 64, 4);
    Coop4Meas meas;
    meas.load(io, trial, li, lj);
    double Sig = 0.0;                                   // the lane's entry of the transition covariance
    {
        Sym<4> Sg;
        CGP_UNROLL for (int k = 0; k < Sym<4>::N; k++) Sg.a[k] = 0.0;
        model.add_sigma(Sg, 1.0);
        CGP_UNROLL for (int i = 0; i < 4; i++) CGP_UNROLL for (int j = 0; j < 4; j++) if (li == i && lj == j) Sig = Sg(i, j);
    }
    const int s2_idx = 5 + Sym<4>::idx(li, lj);         // this lane's entry among the totals: [wsum, mean x4, second moment x10]

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    double u0 = m0p[0], u1 = m0p[1], u2 = m0p[2], u3 = m0p[3];
    double P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, li, lj);
    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    Coop4FilterOut out;
    out.init(io, trial);
    const int ng = sg.groups();
    LanePoints pts;
    CollapsedPoint cpt;
    if constexpr (COLLAPSED) cpt.load(sg, lane); else pts.load(sg, lane);
    // collapsed combine, per lane (i, j): which totals / closed forms its entry of Pp is made of
    const bool both_nl = li < 2 && lj < 2, both_lin = li >= 2 && lj >= 2;
    const int lo = li < lj ? li : lj, hi_ = li < lj ? lj : li;
    const int idx_s = 2 + li + lj;                                     // s00, s10, s11 at totals 2, 3, 4
    const int idx_x2 = 5 + (lo < 2 ? lo : 0), idx_x3 = 7 + (lo < 2 ? lo : 0);
    const double cm0 = (!both_nl && !both_lin) ? (hi_ == 2 ? model.M[0] : model.M[2]) : 0.0;
    const double cm1 = (!both_nl && !both_lin) ? (hi_ == 2 ? model.M[1] : model.M[3]) : 0.0;
    const double v22c = (li == 2 && lj == 2) ? 1.0 : 0.0, v33c = (li == 3 && lj == 3) ? 1.0 : 0.0;
    const double v32c = (both_lin && li != lj) ? 1.0 : 0.0;

    double cum = 0.0, S_l = 1.0, innov_l = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        for (int slot = 0; slot < nsteps; slot++) {
            const double y = readlane_f64(ychunk, slot);
            // ---- sigma-point prediction (filters_smoothers.py:88-121)
            Sym<4> Pr, L; Vec<4> inv, m;
            coop4_gather(P, Pr);
            m.v[0] = u0; m.v[1] = u1; m.v[2] = u2; m.v[3] = u3;
            cholesky<4>(Pr, L, inv);
            if constexpr (COLLAPSED) {
                const double d0 = L(0, 0) * cpt.xi0;
                const double d1 = fma(L(1, 1), cpt.xi1, L(1, 0) * cpt.xi0);
                const double d2 = fma(L(2, 2), cpt.xi2, fma(L(2, 1), cpt.xi1, L(2, 0) * cpt.xi0));
                const double d3 = fma(L(3, 2), cpt.xi2, fma(L(3, 1), cpt.xi1, L(3, 0) * cpt.xi0));
                const double h0 = u0 + d0, h1 = u1 + d1;
                typename DM::Pre pre;
                model.precompute(u2 + d2, pre);      // one group per lane: an anchored rotation would only add the anchor to the chain
                const double g0 = pre.c[0] * h0 - pre.s[0] * h1, g1 = pre.s[0] * h0 + pre.c[0] * h1;
                const double w0 = cpt.W * g0, w1 = cpt.W * g1;
                double* tot = red + kRedChunk * kRedLd;
                red[0 * kRedLd + lane] = w0; red[1 * kRedLd + lane] = w1;
                red[2 * kRedLd + lane] = w0 * g0; red[3 * kRedLd + lane] = w1 * g0; red[4 * kRedLd + lane] = w1 * g1;
                red[5 * kRedLd + lane] = w0 * d2; red[6 * kRedLd + lane] = w1 * d2;
                red[7 * kRedLd + lane] = w0 * d3; red[8 * kRedLd + lane] = w1 * d3;
                wave_lds_fence();
                {
                    const int r = lane >> 1, h = lane & 1;
                    const double2* row = reinterpret_cast<const double2*>(red + r * kRedLd + h * 16);
                    double sum = 0.0;
                    if (r < 9) sum = row_sum16(row);
                    sum += dpp_f64<kQuadSwap1>(sum);
                    if (h == 0 && r < 9) tot[r] = sum;
                }
                wave_lds_fence();
                const double poison = L(0, 0) - L(0, 0);
                const double M0 = model.M[0], M1 = model.M[1], M2 = model.M[2], M3 = model.M[3];
                const double f0 = tot[0], f1 = tot[1];
                const double f2 = fma(M0, u2, M1 * u3) + poison, f3 = fma(M2, u2, M3 * u3) + poison;
                const double t20 = fma(M0, Pr(2, 2), M1 * Pr(3, 2)), t21 = fma(M0, Pr(3, 2), M1 * Pr(3, 3));
                const double t30 = fma(M2, Pr(2, 2), M3 * Pr(3, 2)), t31 = fma(M2, Pr(3, 2), M3 * Pr(3, 3));
                const double V22 = fma(t20, M0, t21 * M1), V32 = fma(t30, M0, t31 * M1), V33 = fma(t30, M2, t31 * M3);
                const double nl = tot[idx_s] - tot[li < 2 ? li : 0] * tot[lj < 2 ? lj : 0];
                const double mixed = fma(cm0, tot[idx_x2], cm1 * tot[idx_x3]);
                const double lin = fma(v22c, V22, fma(v32c, V32, v33c * V33));
                const double Pp = ((both_nl ? nl : (both_lin ? lin : mixed)) + Sig) + poison;
                wave_lds_fence();
                double S, innov;
                coop4_update(meas, Pp, f0, f1, f2, f3, y, P, u0, u1, u2, u3, S, innov);
                if (lane == slot) { S_l = S; innov_l = innov; }
                out.store(t0 + slot, lane, P, u0, u1, u2, u3);
                continue;
            }
            double acc[15];
            CGP_UNROLL for (int k = 0; k < 15; k++) acc[k] = 0.0;
            if (pts.ok) {
                if (pts.n > 0) {
                    Vec<4> chi, f;
                    pts.point(0, m, L, chi);
                    typename DM::Pre pre;
                    model.precompute(chi.v[DM::IVC], pre);
                    CGP_UNROLL for (int k = 0; k < kCacheM; k++) {
                        if (k < pts.n) {
                            if (k > 0) pts.point(k, m, L, chi);
                            model.mean_pre(chi, pre, f);
                            const double w = pts.w[k];
                            acc[0] += w;
                            double wf[4];
                            CGP_UNROLL for (int i = 0; i < 4; i++) { wf[i] = w * f.v[i]; acc[1 + i] += wf[i]; }
                            CGP_UNROLL for (int i = 0; i < 4; i++)
                                CGP_UNROLL for (int j = 0; j <= i; j++)
                                    acc[5 + Sym<4>::idx(i, j)] = fma(wf[i], f.v[j], acc[5 + Sym<4>::idx(i, j)]);
                        }
                    }
                }
            } else
            for (int g = lane; g < ng; g += 64) {
                int p = sg.template begin<true>(g);
                const int pe = sg.template end<true>(g);
                Vec<4> chi, f;
                sigma_point<4, true>(m, L, sg, p, chi);
                typename DM::Pre pre;
                model.precompute(chi.v[DM::IVC], pre);
                for (;;) {
                    model.mean_pre(chi, pre, f);
                    const double w = sg.template weight<true>(p);
                    acc[0] += w;
                    double wf[4];
                    CGP_UNROLL for (int i = 0; i < 4; i++) { wf[i] = w * f.v[i]; acc[1 + i] += wf[i]; }
                    CGP_UNROLL for (int i = 0; i < 4; i++)
                        CGP_UNROLL for (int j = 0; j <= i; j++)
                            acc[5 + Sym<4>::idx(i, j)] = fma(wf[i], f.v[j], acc[5 + Sym<4>::idx(i, j)]);
                    if (++p >= pe) break;
                    sigma_point<4, true>(m, L, sg, p, chi);
                }
            }
            const double* tot = coop_reduce_to_lds<15>(acc, red, lane, ng <= 32);
            const double wsum = tot[0];
            const double f0 = tot[1], f1 = tot[2], f2 = tot[3], f3 = tot[4];
            // Pp = E[f f^T + Sigma] - mp mp^T, this lane's entry
            const double Pp = fma(wsum, Sig, tot[s2_idx]) - tot[1 + li] * tot[1 + lj];
            // ---- update
            double S, innov;
            coop4_update(meas, Pp, f0, f1, f2, f3, y, P, u0, u1, u2, u3, S, innov);
            if (lane == slot) { S_l = S; innov_l = innov; }
            out.store(t0 + slot, lane, P, u0, u1, u2, u3);
        }
        if (out.want_nll) cum = nll_flush_wave(S_l, innov_l, lane, nsteps, cum, out.nll ? out.nll + t0 : nullptr);
    }
    if (lane == 0 && io.nll && out.nll_final) io.nll[trial] = cum;
}

// ------------------------------------------------------------------------------------------------ cd sigma-point moment ODE
// One evaluation of the sigma-point moment ODE (filters_smoothers.py:124-137) at (m, distributed P):
// km[l] = E[a_l] (replicated) and this lane's entry of C + C^T + gamma with C = E[(chi - m) a^T].
template <class SM>
CGP_DEV void coop4_cd_sgp_rhs(const SM& model, const SigmaSet& sg, const LanePoints& pts, int ng, double* red, int lane,
                              int cij_idx, int cji_idx, double gam, const Vec<4>& m, double P, Vec<4>& km, double& kP) {
    Sym<4> Pr, L; Vec<4> inv;
    coop4_gather(P, Pr);
    cholesky<4>(Pr, L, inv);
    double acc[20];
    CGP_UNROLL for (int k = 0; k < 20; k++) acc[k] = 0.0;
    if (pts.ok) {
        if (pts.n > 0) {
            Vec<4> chi, a;
            pts.point(0, m, L, chi);
            typename SM::Pre pre;
            model.precompute(chi.v[SM::IVC], pre);
            CGP_UNROLL for (int k = 0; k < kCacheM; k++) {
                if (k < pts.n) {
                    if (k > 0) pts.point(k, m, L, chi);
                    model.drift_pre(chi, pre, a);
                    const double w = pts.w[k];
                    double wa[4];
                    CGP_UNROLL for (int i = 0; i < 4; i++) { wa[i] = w * a.v[i]; acc[i] += wa[i]; }
                    CGP_UNROLL for (int i = 0; i < 4; i++) {
                        const double ci = chi.v[i] - m.v[i];
                        CGP_UNROLL for (int j = 0; j < 4; j++) acc[4 + i * 4 + j] = fma(ci, wa[j], acc[4 + i * 4 + j]);
                    }
                }
            }
        }
    } else
    for (int g = lane; g < ng; g += 64) {
        int p = sg.template begin<true>(g);
        const int pe = sg.template end<true>(g);
        Vec<4> chi, a;
        sigma_point<4, true>(m, L, sg, p, chi);
        typename SM::Pre pre;
        model.precompute(chi.v[SM::IVC], pre);
        for (;;) {
            model.drift_pre(chi, pre, a);
            const double w = sg.template weight<true>(p);
            double wa[4];
            CGP_UNROLL for (int i = 0; i < 4; i++) { wa[i] = w * a.v[i]; acc[i] += wa[i]; }
            CGP_UNROLL for (int i = 0; i < 4; i++) {
                const double ci = chi.v[i] - m.v[i];
                CGP_UNROLL for (int j = 0; j < 4; j++) acc[4 + i * 4 + j] = fma(ci, wa[j], acc[4 + i * 4 + j]);
            }
            if (++p >= pe) break;
            sigma_point<4, true>(m, L, sg, p, chi);
        }
    }
    const double* tot = coop_reduce_to_lds<20>(acc, red, lane, ng <= 32);
    km.v[0] = tot[0]; km.v[1] = tot[1]; km.v[2] = tot[2]; km.v[3] = tot[3];
    kP = (tot[cij_idx] + tot[cji_idx]) + gam;
}


// ---- the same right-hand side with the linear structure of the chirp SDE taken out of the quadrature ---------------
// For a CGP_SIGMA_STANDARD set (sum w = 1, sum w xi = 0, sum w xi xi^T = I; members of a group differ in xi_3 only, with
// zero weighted mean) and the chirp drift a = (-lam chi_0 - om chi_1, om chi_0 - lam chi_1, chi_3, -g^2 chi_2 - 2 g chi_3),
// om = om(chi_2), the sums of filters_smoothers.py:124-137 regroup EXACTLY (L lower-triangular, so chi_0..2 and hence
// a_0, a_1 do not depend on xi_3):
//     E[a_2] = m_3,  E[a_3] = -g^2 m_2 - 2 g m_3,        C[:, 2] = P[:, 3],  C[:, 3] = -g^2 P[:, 2] - 2 g P[:, 3]
//     E[a_j] = sum_g W_g a_j(g),   C[i][j] = sum_g W_g d_i(g) a_j(g)   (j < 2),   d = L xi restricted to xi_0..2
// One point per lane instead of three, 10 partial sums instead of 20, no L[3][3].  A failed Cholesky poisons every
// output with NaN like the literal sums do.
// per-lane selectors of the combine: which closed-form column the lane's (i, j) and (j, i) entries come from
struct CollapsedRole {
    int tij, tji;                 // indices of C[i][j], C[j][i] among the totals (valid when the column is < 2)
    int pi, pj;                   // LDS offsets of P[i][2..3], P[j][2..3]
    double c2j, c3j, c2i, c3i;    // closed-form column coefficients on (P[.][2], P[.][3])
    bool nlj, nli;
    CGP_DEV void init(int li, int lj, double g) {
        tij = 2 + 2 * li + (lj < 2 ? lj : 0); tji = 2 + 2 * lj + (li < 2 ? li : 0);
        pi = 4 * li + 2; pj = 4 * lj + 2;
        nlj = lj < 2; nli = li < 2;
        c2j = lj == 3 ? -(g * g) : 0.0; c3j = lj == 3 ? -2.0 * g : (lj == 2 ? 1.0 : 0.0);
        c2i = li == 3 ? -(g * g) : 0.0; c3i = li == 3 ? -2.0 * g : (li == 2 ? 1.0 : 0.0);
    }
};
template <class SM>
CGP_DEV void coop4_cd_sgp_rhs_collapsed(const SM& model, const CollapsedPoint& pt, const CollapsedRole& role, double* red, int lane,
                                        double gam, const Vec<4>& m, double P, Vec<4>& km, double& kP) {
    Sym<4> Pr, L; Vec<4> inv;
    coop4_gather(P, Pr);
    cholesky<4>(Pr, L, inv);                       // L[3][3] itself is never used: its square root is dead code
    const double d0 = L(0, 0) * pt.xi0;
    const double d1 = fma(L(1, 1), pt.xi1, L(1, 0) * pt.xi0);
    const double d2 = fma(L(2, 2), pt.xi2, fma(L(2, 1), pt.xi1, L(2, 0) * pt.xi0));
    const double d3 = fma(L(3, 2), pt.xi2, fma(L(3, 1), pt.xi1, L(3, 0) * pt.xi0));
    const double c0 = m.v[0] + d0, c1 = m.v[1] + d1, c2 = m.v[2] + d2;
    typename SM::Pre pre;
    model.precompute(c2, pre);
    const double wa0 = pt.W * (-model.lam * c0 - pre.w * c1);
    const double wa1 = pt.W * (pre.w * c0 - model.lam * c1);
    double* tot = red + kRedChunk * kRedLd;
    double* prow = red + (kRedChunk - 1) * kRedLd;      // row 31 of the partials area is free here (10 rows used)
    red[0 * kRedLd + lane] = wa0;
    red[1 * kRedLd + lane] = wa1;
    red[2 * kRedLd + lane] = d0 * wa0; red[3 * kRedLd + lane] = d0 * wa1;
    red[4 * kRedLd + lane] = d1 * wa0; red[5 * kRedLd + lane] = d1 * wa1;
    red[6 * kRedLd + lane] = d2 * wa0; red[7 * kRedLd + lane] = d2 * wa1;
    red[8 * kRedLd + lane] = d3 * wa0; red[9 * kRedLd + lane] = d3 * wa1;
    if (lane < 16) prow[lane] = P;
    wave_lds_fence();
    {
        const int r = lane >> 1, h = lane & 1;
        const double2* row = reinterpret_cast<const double2*>(red + r * kRedLd + h * 16);
        double s = 0.0;
        if (r < 10) {
            s = row_sum16(row);
        }
        s += dpp_f64<kQuadSwap1>(s);
        if (h == 0 && r < 10) tot[r] = s;
    }
    wave_lds_fence();
    const double poison = L(0, 0) - L(0, 0);      // 0, or NaN when the factorisation failed (cholesky() poisons L)
    const double g = model.gam;
    km.v[0] = tot[0]; km.v[1] = tot[1];
    km.v[2] = m.v[3] + poison;
    km.v[3] = fma(-(g * g), m.v[2], -2.0 * g * m.v[3]) + poison;
    const double2 Pi = *reinterpret_cast<const double2*>(prow + role.pi), Pj = *reinterpret_cast<const double2*>(prow + role.pj);
    const double cij = role.nlj ? tot[role.tij] : fma(role.c2j, Pi.x, role.c3j * Pi.y);
    const double cji = role.nli ? tot[role.tji] : fma(role.c2i, Pj.x, role.c3i * Pj.y);
    kP = ((cij + cji) + gam) + poison;
    wave_lds_fence();
}

// ------------------------------------------------------------------------------------------------ cd_sgp_filter, d = 4
template <class SM, bool COLLAPSED>
__global__ void __launch_bounds__(64) cdsgp4_coop_kernel(FilterIO io, ModelArgs ma) {
    static_assert(SM::D == 4, "d = 4 kernel");
    __shared__ double red[kFanLdsDoubles];
    const int lane = threadIdx.x;
    const int li = (lane >> 2) & 3, lj = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    SM model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    model.wide = true;
    SigmaSet sg = ma.sg;
    sg.stage(dyn_lds(), lane, 64, 4);
    Coop4Meas meas;
    meas.load(io, trial, li, lj);
    const double gam = coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, li, lj);
    const double dt = ma.dt;
    const int cij_idx = 4 + li * 4 + lj, cji_idx = 4 + lj * 4 + li;

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    Vec<4> u;
    u.v[0] = m0p[0]; u.v[1] = m0p[1]; u.v[2] = m0p[2]; u.v[3] = m0p[3];
    double P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, li, lj);
    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    Coop4FilterOut out;
    out.init(io, trial);
    const int ng = sg.groups();
    LanePoints pts;
    CollapsedPoint cpt;
    CollapsedRole role;
    if constexpr (COLLAPSED) { cpt.load(sg, lane); role.init(li, lj, model.gam); }
    else pts.load(sg, lane);

    double cum = 0.0, S_l = 1.0, innov_l = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        for (int slot = 0; slot < nsteps; slot++) {
            const double y = readlane_f64(ychunk, slot);
            // ---- RK4 on (m, P) (quadratures.py:34-54), same operation order as cgp_steps.hpp:rk4_m_cov
            Vec<4> tm = u, am, km;
            double tP = P, aP = 0.0, kP;
            CGP_UNROLL for (int i = 0; i < 4; i++) am.v[i] = 0.0;
#pragma unroll 1
            for (int stage = 0; stage < 4; stage++) {
                if constexpr (COLLAPSED) coop4_cd_sgp_rhs_collapsed<SM>(model, cpt, role, red, lane, gam, tm, tP, km, kP);
                else coop4_cd_sgp_rhs<SM>(model, sg, pts, ng, red, lane, cij_idx, cji_idx, gam, tm, tP, km, kP);
                const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
                const double half = (stage == 2) ? 1.0 : 0.5;
                CGP_UNROLL for (int i = 0; i < 4; i++) { am.v[i] = fma(wgt, km.v[i], am.v[i]); tm.v[i] = u.v[i] + (dt * km.v[i]) * half; }
                aP = fma(wgt, kP, aP);
                tP = P + (dt * kP) * half;
            }
            const double f0 = u.v[0] + (dt * am.v[0]) / 6.0, f1 = u.v[1] + (dt * am.v[1]) / 6.0;
            const double f2 = u.v[2] + (dt * am.v[2]) / 6.0, f3 = u.v[3] + (dt * am.v[3]) / 6.0;
            const double Pp = P + (dt * aP) / 6.0;
            // ---- update
            double S, innov;
            coop4_update(meas, Pp, f0, f1, f2, f3, y, P, u.v[0], u.v[1], u.v[2], u.v[3], S, innov);
            if (lane == slot) { S_l = S; innov_l = innov; }
            out.store(t0 + slot, lane, P, u.v[0], u.v[1], u.v[2], u.v[3]);
        }
        if (out.want_nll) cum = nll_flush_wave(S_l, innov_l, lane, nsteps, cum, out.nll ? out.nll + t0 : nullptr);
    }
    if (lane == 0 && io.nll && out.nll_final) io.nll[trial] = cum;
}

// ------------------------------------------------------------------------------------------------ backward gains, a chunk at a time
// G_k = Pf_k^{-1} gamma of the continuous-discrete smoothers (filters_smoothers.py:429, 617-618) depends on the filtering result
// of step k alone.  Computed inside the step it is a Cholesky and four solves that all 64 lanes repeat identically (150 of the
// ~800 / 1200 instructions of a cd_eks / cd_sgp_smoother step); instead each chunk of 64 steps factorises its 64 covariances
// lane-parallel -- lane l takes step t_hi - l -- and parks (G, mf) in LDS, where the serial walk picks them up as broadcast reads.
constexpr int kGainPitch = 22;                     // 16 (G, row-major) + 4 (mf) + 2: 16-byte aligned records, conflict-free lane stride
CGP_DEV void coop4_chunk_gains(double* gbuf, int lane, int n, int64_t t_hi, const double* __restrict__ mfs, const double* __restrict__ Pfs,
                               const Sym<4>& gamma) {
    if (lane < n) {
        const int64_t t = t_hi - lane;
        Vec<4> mf; Sym<4> Pf;
        load_vec<4>(mfs + t * 4, mf);
        load_sym<4>(Pfs + t * 16, Pf);
        Mat<4> PG;
        pinv_gamma<4>(Pf, gamma, PG);
        double* g = gbuf + lane * kGainPitch;
        CGP_UNROLL for (int i = 0; i < 4; i++) CGP_UNROLL for (int j = 0; j < 4; j += 2) *reinterpret_cast<double2*>(g + i * 4 + j) = make_double2(PG.a[i][j], PG.a[i][j + 1]);
        *reinterpret_cast<double2*>(g + 16) = make_double2(mf.v[0], mf.v[1]);
        *reinterpret_cast<double2*>(g + 18) = make_double2(mf.v[2], mf.v[3]);
    }
    wave_lds_fence();
}
// the record of one step: G and mf as wave-uniform values
CGP_DEV void coop4_read_gain(const double* g, Mat<4>& PG, Vec<4>& mf) {
    CGP_UNROLL for (int i = 0; i < 4; i++) CGP_UNROLL for (int j = 0; j < 4; j += 2) {
        const double2 v = *reinterpret_cast<const double2*>(g + i * 4 + j);
        PG.a[i][j] = v.x; PG.a[i][j + 1] = v.y;
    }
    const double2 a = *reinterpret_cast<const double2*>(g + 16), b = *reinterpret_cast<const double2*>(g + 18);
    mf.v[0] = a.x; mf.v[1] = a.y; mf.v[2] = b.x; mf.v[3] = b.y;
}

// ------------------------------------------------------------------------------------------------ cd_sgp_smoother, d = 4
// Backward RK4 with  dm = _m + G^T (m - mf),  dP = _P + G^T P + P G - 2 gamma,  G = Pf^{-1} gamma  (filters_smoothers.py:615-621).
// G is constant over the four stages (hoisted, as in cgp_steps.hpp); its entries reach the lanes through LDS:
//     (G^T P)[i][j] = sum_r G[l_r][i] P[l_r][j]   (rows l_r by DPP row rotations, G[l_r][i] per lane)
//     (P G)[i][j]   = sum_l P[i][l] G[l][j]       (quad broadcasts, G[l][j] per lane)
template <class SM, bool COLLAPSED>
__global__ void __launch_bounds__(64) cdsgps4_coop_kernel(SmootherIO io, ModelArgs ma) {
    static_assert(SM::D == 4, "d = 4 kernel");
    __shared__ double red[kFanLdsDoubles];
    __shared__ __attribute__((aligned(16))) double gbuf[64 * kGainPitch];
    const int lane = threadIdx.x;
    const int li = (lane >> 2) & 3, lj = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    SM model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    model.wide = true;
    SigmaSet sg = ma.sg;
    sg.stage(dyn_lds(), lane, 64, 4);
    Sym<4> gamma;
    load_sym<4>(ma.gamma + trial * ma.gamma_stride, gamma);
    const double gam = coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, li, lj);
    const double dt = -ma.dt;
    const int cij_idx = 4 + li * 4 + lj, cji_idx = 4 + lj * 4 + li;
    // source rows of the three row rotations, discovered by rotating the row index itself
    const int lr1 = dpp_i32<kRowRor4>(li), lr2 = dpp_i32<kRowRor8>(li), lr3 = dpp_i32<kRowRor12>(li);
    const int ng = sg.groups();
    LanePoints pts;
    CollapsedPoint cpt;
    CollapsedRole role;
    if constexpr (COLLAPSED) { cpt.load(sg, lane); role.init(li, lj, model.gam); }
    else pts.load(sg, lane);

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
    coop4_chunk_gains(gbuf, lane, nsteps, t_hi, mfs, Pfs, gamma);
    for (int slot = 0; slot < nsteps; slot++) {
        const int64_t t = t_hi - slot;
        const double* gl = gbuf + slot * kGainPitch;
        Mat<4> PG; Vec<4> mf;
        coop4_read_gain(gl, PG, mf);
        const double gr0 = gl[li * 4 + li], gr1 = gl[lr1 * 4 + li], gr2 = gl[lr2 * 4 + li], gr3 = gl[lr3 * 4 + li];
        const double gc0 = gl[0 * 4 + lj], gc1 = gl[1 * 4 + lj], gc2 = gl[2 * 4 + lj], gc3 = gl[3 * 4 + lj];

        Vec<4> tm = ms, am, km;
        double tP = Ps, aP = 0.0, kP;
        CGP_UNROLL for (int i = 0; i < 4; i++) am.v[i] = 0.0;
#pragma unroll 1
        for (int stage = 0; stage < 4; stage++) {
            // (_m, _P), _P includes + gamma
            if constexpr (COLLAPSED) coop4_cd_sgp_rhs_collapsed<SM>(model, cpt, role, red, lane, gam, tm, tP, km, kP);
            else coop4_cd_sgp_rhs<SM>(model, sg, pts, ng, red, lane, cij_idx, cji_idx, gam, tm, tP, km, kP);
            CGP_UNROLL for (int i = 0; i < 4; i++) {
                double s = km.v[i];
                CGP_UNROLL for (int k = 0; k < 4; k++) s = fma(PG.a[k][i], tm.v[k] - mf.v[k], s);
                km.v[i] = s;                                                                        // _m + G^T (m - mf)
            }
            double tij = gr0 * tP;                                                                  // (G^T P)[i][j]
            tij = fma(gr1, dpp_f64<kRowRor4>(tP), tij);
            tij = fma(gr2, dpp_f64<kRowRor8>(tP), tij);
            tij = fma(gr3, dpp_f64<kRowRor12>(tP), tij);
            double tji = gc0 * dpp_f64<kQuadBcast0>(tP);                                            // (P G)[i][j] = (G^T P)[j][i]
            tji = fma(gc1, dpp_f64<kQuadBcast1>(tP), tji);
            tji = fma(gc2, dpp_f64<kQuadBcast2>(tP), tji);
            tji = fma(gc3, dpp_f64<kQuadBcast3>(tP), tji);
            kP = (kP + (tij + tji)) - 2.0 * gam;
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

// The collapsed quadrature needs the caller's CGP_SIGMA_STANDARD assertion, groups, and one group per lane of a half wave.
inline bool collapsed_ok(const ModelArgs& ma) {
    return (ma.sg.flags & CGP_SIGMA_STANDARD) && ma.sg.group_start && ma.sg.n_groups >= 1 && ma.sg.n_groups <= 32;
}

template <class DM>
inline int launch_sgp4_coop(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (collapsed_ok(ma)) hipLaunchKernelGGL((sgp4_coop_kernel<DM, true>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    else hipLaunchKernelGGL((sgp4_coop_kernel<DM, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    return hip_rc(hipGetLastError());
}
template <class SM>
inline int launch_cdsgp4_coop(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (collapsed_ok(ma)) hipLaunchKernelGGL((cdsgp4_coop_kernel<SM, true>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    else hipLaunchKernelGGL((cdsgp4_coop_kernel<SM, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    return hip_rc(hipGetLastError());
}
template <class SM>
inline int launch_cdsgps4_coop(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (collapsed_ok(ma)) hipLaunchKernelGGL((cdsgps4_coop_kernel<SM, true>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    else hipLaunchKernelGGL((cdsgps4_coop_kernel<SM, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    return hip_rc(hipGetLastError());
}

}  // namespace cgp
