// cgp_mfma4_cd.hpp -- cd_sgp_filter / cd_sgp_smoother (filters_smoothers.py:534-632) for the chirp / La Scala SDE model,
// d = 4, with the sigma-point moment ODE (filters_smoothers.py:124-137) on the float64 matrix cores (BASELINE config C4:
// Gauss-Hermite order 3, RK4, T = 50 000).
//
// Same layout and fan as cgp_mfma4_sigma.hpp: the covariance one entry per lane in the OUTPUT layout of
// v_mfma_f64_4x4x4_4b_f64 (lane (r, b, q): P[r][q]), group p = 16 s + 4 b + k evaluated by the four lanes of (k, b), the
// softplus of a group evaluated once (lane q & 1 = s) and broadcast inside the quad, nothing through LDS.  One RK4 stage:
//   * the collapsed quadrature of cgp_coop4_sigma.hpp:coop4_cd_sgp_rhs_collapsed.  The drift's last two components are
//     linear, a_2 = x_3, a_3 = -g^2 x_2 - 2 g x_3, so their expectations and the columns 2, 3 of C = E[(x - m) a^T] are
//     closed forms in (m, P): C[r][2] = P[r][3], C[r][3] = -g^2 P[r][2] - 2 g P[r][3] -- two quad broadcasts of the lane's
//     own row.  The columns 0, 1 are sums over the groups (xi_3 drops out: a_0, a_1 do not depend on it and its mean is 0),
//         C[r][q] = sum_p W_p d_r a_q,   d = L xi (xi_0..2),   a_0 = -lam c_0 - w c_1,  a_1 = w c_0 - lam c_1,  c = m + d,
//     i.e. two chained matrix instructions (A = W d_r, B = a_q) plus the sum over the four blocks, and E[a_q] two more;
//   * dP/dt = C + C^T + gamma: the transpose is one more matrix instruction (B = identity), with C + gamma riding in as its
//     accumulator;
//   * the smoother's  G^T P + P G  (filters_smoothers.py:615-621, G = Pf^{-1} gamma) is two chained matrix instructions on
//     the distributed G and P (P symmetric) instead of eight cross-lane moves and eight multiply-adds.
//   * the mean lives in COLUMN form, lane (r, b, q) holding m[q]: the RK4 bookkeeping and the mean update are one
//     instruction for all four components, E[a] comes out of the matrix instruction in that form, the fan takes m_0..3 as
//     quad broadcasts, and the smoother's G^T (m - mf) is two matrix instructions (a transpose to row form, then A = the
//     row form, B = G).
// A factorisation that fails (a pivot <= 0) poisons every output with NaN like the reference's NaN Cholesky factor does.
#pragma once
#include "cgp_mfma4_sigma.hpp"

namespace cgp {

// RK4's "/ 6" (quadratures.py:53) as a multiplication: within an ulp of the division, a dozen instructions less per use.
constexpr double kSixth = 1.0 / 6.0;

// Per-lane constants of the moment ODE in the matrix-core layout.
struct Cd4LaneCoef {
    bool odd;                          // q & 1: the pass whose softplus this lane evaluates
    double we[2][4];                   // W_s on the component of d the lane feeds as the A operand (lane & 3), 0 elsewhere
    double k1, kn, kl0, kl1;           // a_q = (k1 w + kl0) c_0 + (kn w + kl1) c_1
    double cc2, cc3;                   // closed-form columns: C[r][q] = cc2 P[r][2] + cc3 P[r][3] for q >= 2
    double ident, gam;                 // (r == q), gamma[r][q]
    double g2, g1;                     // -g^2, -2 g
    CGP_DEV void init(int row, int q, const double (&W)[2], double lam, double g, double gamma_rq) {
        odd = (q & 1) != 0;
        CGP_UNROLL for (int s = 0; s < 2; s++) CGP_UNROLL for (int c = 0; c < 4; c++) we[s][c] = (q == c) ? W[s] : 0.0;
        k1 = (q == 1) ? 1.0 : 0.0; kn = (q == 0) ? -1.0 : 0.0;
        kl0 = (q == 0) ? -lam : 0.0; kl1 = (q == 1) ? -lam : 0.0;
        g2 = -(g * g); g1 = -2.0 * g;
        cc2 = (q == 3) ? g2 : 0.0; cc3 = (q == 2) ? 1.0 : (q == 3) ? g1 : 0.0;
        ident = (row == q) ? 1.0 : 0.0;
        gam = gamma_rq;
    }
};

// The fan of one stage: B operands a_q and A operands W d_r of both passes.
template <int MODE, bool TWO, class SM>
CGP_DEV void cd4_mfma_fan(const SM& model, const SoftplusRegs& R, const Cd4LaneCoef& K, const Fan4Groups& grp, const Sym<4>& l,
                          const double (&sd)[3], double m0, double m1, double m2, double (&a)[2], double (&wd)[2], bool& ok) {
    double c0[2], c1[2], d2[2];
    CGP_UNROLL for (int s = 0; s < (TWO ? 2 : 1); s++) {                // d = L xi, L = l diag(sd) (l unit lower), xi_3 left out
        const double xs0 = grp.xi[s][0] * sd[0], xs1 = grp.xi[s][1] * sd[1], xs2 = grp.xi[s][2] * sd[2];
        const double d1 = fma(l(1, 0), xs0, xs1);
        d2[s] = fma(l(2, 1), xs1, fma(l(2, 0), xs0, xs2));
        const double d3 = fma(l(3, 2), xs2, fma(l(3, 1), xs1, l(3, 0) * xs0));
        c0[s] = m0 + xs0; c1[s] = m1 + d1;
        wd[s] = fma(K.we[s][3], d3, fma(K.we[s][2], d2[s], fma(K.we[s][1], d1, K.we[s][0] * xs0)));   // W d_r: a select as arithmetic
    }
    typename SM::Pre pre;
    const double uv = m2 + ((TWO && K.odd) ? d2[1] : d2[0]);
    if constexpr (MODE == kFanSpec) model.precompute_spec(R, uv, pre, ok);
    else if constexpr (MODE == kFanAny) model.precompute_any(R, uv, pre, ok);
    else { model.precompute(uv, pre); ok = true; }
    const double w0 = TWO ? dpp_f64<kQuadBcast0>(pre.w) : pre.w;
    a[0] = fma(fma(K.k1, w0, K.kl0), c0[0], fma(K.kn, w0, K.kl1) * c1[0]);
    if constexpr (TWO) {
        const double w1 = dpp_f64<kQuadBcast1>(pre.w);
        a[1] = fma(fma(K.k1, w1, K.kl0), c0[1], fma(K.kn, w1, K.kl1) * c1[1]);
    }
}

// One evaluation of the sigma-point moment ODE at (m in column form, P distributed): km = E[a] in column form and the
// lane's entry of C + C^T + gamma.
// (a lane outside the lean regime [1.5, 700) sends the wavefront through the branch-free ANY fan -- round 5, cgp_models.hpp:
// precompute_any -- and only from there through the checked fan with its regime branches)
template <bool TWO, class SM>
CGP_DEV void cd4_mfma_rhs(const SM& model, const SoftplusRegs& R, const Cd4LaneCoef& K, const Fan4Groups& grp,
                          double mcol, double P, double& kmcol, double& kP) {
    // A pivot <= 0 among the first three makes its square root NaN, and through d every sum of the fan; the last pivot's
    // root is never taken (xi_3 is left out), so its sign is checked and folded into sd[2]: C, E[a] -- all entries,
    // their zero columns included (0 x NaN) -- and with them every output turn NaN, as the reference's NaN factor does
    Sym<4> l; double sd[3], dv[4];
    mfma4_factor(P, l, sd, dv);
    sd[2] += (dv[3] > 0.0) ? 0.0 : __builtin_nan("");
    const double m0 = row_bcast_f64<0>(mcol), m1 = row_bcast_f64<1>(mcol), m2 = row_bcast_f64<2>(mcol), m3 = row_bcast_f64<3>(mcol);
    double a[2], wd[2]; bool ok;
    cd4_mfma_fan<kFanSpec, TWO>(model, R, K, grp, l, sd, m0, m1, m2, a, wd, ok);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
        cd4_mfma_fan<kFanAny, TWO>(model, R, K, grp, l, sd, m0, m1, m2, a, wd, ok);
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) cd4_mfma_fan<kFanChecked, TWO>(model, R, K, grp, l, sd, m0, m1, m2, a, wd, ok);
    }
    double C = mfma4x4(wd[0], a[0], 0.0);
    double F = mfma4x4(grp.W[0], a[0], 0.0);
    if constexpr (TWO) {
        C = mfma4x4(wd[1], a[1], C);
        F = mfma4x4(grp.W[1], a[1], F);
    }
    C = blk_allreduce(C);                                                // sum_p W d_r a_q   (columns 0, 1; zero elsewhere)
    F = blk_allreduce(F);                                                // sum_p W a_q, in every row (zero for q >= 2)
    kmcol = fma(K.cc2, m2, fma(K.cc3, m3, F));                           // E[a_2] = m_3, E[a_3] = -g^2 m_2 - 2 g m_3: the closed-form columns' coefficients
    const double P2 = row_bcast_f64<2>(P), P3 = row_bcast_f64<3>(P);      // P[r][2], P[r][3]
    const double Cf = C + fma(K.cc2, P2, K.cc3 * P3);
    kP = mfma4x4(Cf, K.ident, Cf + K.gam);                               // C^T + (C + gamma)
}

// Scalar-measurement update (filters_smoothers.py:55-68) in the matrix-core layout: three matrix instructions (see
// cgp_mfma4_sigma.hpp); the mean in column form (Hq = H[q]: H f is a sum over the quad).
CGP_DEV void mfma4_update_col(double Pp, double fcol, double Hk, double Hq, double Xi, double y,
                              double& P, double& ucol, double& S, double& innov) {
    const double PHc = mfma4x4(Hk, Pp, 0.0);                             // (Pp H^T)[q] in every row
    const double PHr = mfma4x4(Pp, Hk, 0.0);                             // (Pp H^T)[r] in every column
    S = mfma4x4(Hk, PHr, Xi);                                            // H Pp H^T + Xi
    double pred = Hq * fcol;
    pred += dpp_f64<kQuadSwap1>(pred);
    pred += dpp_f64<kQuadSwap2>(pred);
    innov = y - pred;
    const double rS = rcp_nr1(S);
    P = fma(-(PHr * rS), PHc, Pp);                                       // Pf = Pp - K (Pp H)^T
    ucol = fma(PHc, rS * innov, fcol);                                   // mf = mp + K innov
}

// ------------------------------------------------------------------------------------------------ cd_sgp_filter, d = 4
template <class SM, bool TWO, bool SPLIT>
__global__ void __launch_bounds__(64) cdsgp4_mfma_kernel(FilterIO io, ModelArgs ma) {
    static_assert(SM::D == 4, "d = 4 kernel");
    __shared__ double2 park[64];                                         // (S, innovation) of the chunk's steps, for the NLL
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const FilterSpan span = filter_span<SPLIT>(io, blockIdx.x);          // (a time-split launch: one SEGMENT of the trial's record)
    const int64_t trial = span.trial;
    if (trial >= io.B) return;

    SM model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    model.wide = true;
    SigmaSet sg = ma.sg;
    sg.stage(dyn_lds(), lane, 64, 4);
    Fan4Groups grp;
    grp.template load<TWO>(sg, r, b);
    Cd4LaneCoef K;
    K.init(r, q, grp.W, model.lam, model.gam, coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, r, q));
    SoftplusRegs R;
    R.init();
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const double Hk = Hp[r], Hq = Hp[q];                                 // H[k] for the lane's k = lane >> 4 (A or B operand); H[q]
    const double Xi = io.Xi[trial * io.Xi_stride];
    const double dt = ma.dt;

    double u = io.m0[trial * io.m0_stride + q];                          // the mean in column form
    double P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, r, q);
    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    OobWindow wP, wm, wnull;
    wnull.init(nullptr, 0);                                              // (the burn-in chunks of a time-split segment store through it)
    wP.init(io.Pfs ? io.Pfs + trial * T * 16 : nullptr, T * 128);
    wm.init(io.mfs ? io.mfs + trial * T * 4 : nullptr, T * 32);
    const unsigned offP = (b == 0) ? (unsigned)(4 * r + q) * 8u : kOobOffset;      // block 0 stores the 16 entries: one 128-B row
    const unsigned offm = (lane < 4) ? (unsigned)lane * 8u : kOobOffset;           // lanes 0..3 store the mean
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;

    double cum = 0.0;
    for (int64_t t0 = span.t_begin; t0 < span.t_end; t0 += 64) {
        double ychunk = (t0 + lane < span.t_end) ? ys[t0 + lane] : 0.0;
        asm volatile("" : "+v"(ychunk));
        const int nsteps = (span.t_end - t0 < 64) ? (int)(span.t_end - t0) : 64;
        const bool burn = t0 < span.t_out;                               // burn-in chunks of a segment write nothing
        const OobWindow wPc = burn ? wnull : wP, wmc = burn ? wnull : wm;   // an empty window drops the stores; the lane offsets stay loop-invariant
        if (span.state && span.seg > 0 && t0 == span.t_out) {            // the junction: the state the burn-in arrived at
            if (lane < 4) span.state[lane] = u;
            if (b == 0) span.state[4 + 4 * r + q] = P;
        }
        for (int slot = 0; slot < nsteps; slot++) {
            const unsigned t = (unsigned)(t0 + slot);
            const double y = readlane_f64(ychunk, slot);
            // ---- RK4 on (m, P) (quadratures.py:34-54), same operation order as cgp_steps.hpp:rk4_m_cov
            double tm = u, am = 0.0, km, tP = P, aP = 0.0, kP;
#pragma unroll 1
            for (int stage = 0; stage < 4; stage++) {
                cd4_mfma_rhs<TWO>(model, R, K, grp, tm, tP, km, kP);
                const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
                const double dth = (stage == 2) ? dt : 0.5 * dt;                 // (dt k) half == k (dt half): half is a power of two
                am = fma(wgt, km, am); tm = u + dth * km;
                aP = fma(wgt, kP, aP);
                tP = P + dth * kP;
            }
            const double f = u + (dt * am) * kSixth;
            const double Pp = P + (dt * aP) * kSixth;
            // ---- update
            double S, innov;
            mfma4_update_col(Pp, f, Hk, Hq, Xi, y, P, u, S, innov);
            park[slot] = make_double2(S, innov);                        // every lane holds them: same address, same value
            wPc.store(P, t * 128u + offP);
            wmc.store(u, t * 32u + offm);
        }
        if (want_nll && !burn) {
            wave_lds_fence();
            const double2 si = park[lane < nsteps ? lane : 0];
            cum = nll_flush_wave(si.x, si.y, lane, nsteps, cum, nll ? nll + t0 : nullptr);
            wave_lds_fence();
        }
    }
    if (span.state) {                                                    // the segment's last state and its NLL total, for the fix-up pass
        if (lane < 4) span.state[20 + lane] = u;
        if (b == 0) span.state[24 + 4 * r + q] = P;
        if (lane == 0) span.state[40] = cum;
    } else if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
}

// ------------------------------------------------------------------------------------------------ cd_sgp_smoother, d = 4
// Backward RK4 with  dm = _m + G^T (m - mf),  dP = _P + G^T P + P G - 2 gamma,  G = Pf^{-1} gamma  (filters_smoothers.py:615-621).
// G is constant over the four stages and computed a chunk of 64 steps at a time, lane-parallel (cgp_coop4_sigma.hpp:
// coop4_chunk_gains); the walk reads it back from LDS one entry per lane, and mf in column form.
// The chunks of workgroup (trial, seg) of a time-split smoother launch (SmootherIO::bsegs): [j_first, j_last] walked, [j_own, j_last] stored;
// chunk j covers the rows T - 2 - 64 j - 63 .. T - 2 - 64 j, and the carry starts from the filtering row t_start.
struct SmootherSpan { int64_t j_own, j_first, j_last, t_start; bool empty; };
template <bool SPLIT> CGP_DEV SmootherSpan smoother_span(const SmootherIO& io, int seg) {
    const int64_t n_chunks = (io.T - 1 + 63) / 64;
    SmootherSpan sp{0, 0, n_chunks - 1, io.T - 1, false};
    if constexpr (SPLIT) {
        sp.j_own = (int64_t)seg * io.chunks_per_bseg;
        sp.j_last = sp.j_own + io.chunks_per_bseg - 1 < n_chunks - 1 ? sp.j_own + io.chunks_per_bseg - 1 : n_chunks - 1;
        sp.j_first = sp.j_own - io.burn_chunks > 0 ? sp.j_own - io.burn_chunks : 0;
        sp.empty = sp.j_own >= n_chunks;                                 // (more segments than chunks: nothing to do)
        sp.t_start = io.T - 1 - 64 * sp.j_first;
    }
    return sp;
}
// the state a segment's burn-in arrived at (the carry in front of its first own chunk): m in column form, P one entry per lane
CGP_DEV void smoother_junction_store(const SmootherIO& io, int64_t trial, int seg, int r, int b, int q, double ms, double Ps) {
    if (seg > 0 && io.junction) {
        double* __restrict__ jn = io.junction + (trial * io.bsegs + seg) * 20;
        if (r == 0 && b == 0) jn[q] = ms;
        if (b == 0) jn[4 + 4 * r + q] = Ps;
    }
}

// SPLIT (round 6; cgp_smoother_time_split): one wavefront per (trial, segment) -- see SmootherIO::bsegs.  Chunk j covers the rows
// T - 2 - 64 j - 63 .. T - 2 - 64 j; segment s owns the chunks [s cps, (s + 1) cps) and starts burn_chunks chunks earlier in its walk
// (later in time) from the FILTERING row there; the burn-in chunks store through a zero-byte window.
template <class SM, bool TWO, bool SPLIT = false>
__global__ void __launch_bounds__(64) cdsgps4_mfma_kernel(SmootherIO io, ModelArgs ma) {
    static_assert(SM::D == 4, "d = 4 kernel");
    __shared__ __attribute__((aligned(16))) double gbuf[64 * kGainPitch];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int64_t trial = SPLIT ? (int64_t)(blockIdx.x / (unsigned)io.bsegs) : (int64_t)blockIdx.x;
    const int seg = SPLIT ? (int)(blockIdx.x % (unsigned)io.bsegs) : 0;
    if (trial >= io.B) return;

    SM model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    model.wide = true;
    SigmaSet sg = ma.sg;
    sg.stage(dyn_lds(), lane, 64, 4);
    Fan4Groups grp;
    grp.template load<TWO>(sg, r, b);
    Sym<4> gamma;
    load_sym<4>(ma.gamma + trial * ma.gamma_stride, gamma);
    Cd4LaneCoef K;
    K.init(r, q, grp.W, model.lam, model.gam, coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, r, q));
    SoftplusRegs R;
    R.init();
    const double dt = -ma.dt;

    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * 4;
    const double* __restrict__ Pfs = io.Pfs + trial * T * 16;
    double* __restrict__ mss = io.mss + trial * T * 4;
    double* __restrict__ Pss = io.Pss + trial * T * 16;
    OobWindow wP, wm, wPnull, wmnull;
    wP.init(Pss, T * 128);
    wm.init(mss, T * 32);
    wPnull.init(nullptr, 0); wmnull.init(nullptr, 0);                    // the burn-in chunks of a segment store through these
    const unsigned offP = (b == 0) ? (unsigned)(4 * r + q) * 8u : kOobOffset;
    const unsigned offm = (lane < 4) ? (unsigned)lane * 8u : kOobOffset;

    const SmootherSpan sp = smoother_span<SPLIT>(io, seg);
    if (sp.empty) return;
    double ms = mfs[sp.t_start * 4 + q];                                 // the mean in column form
    double Ps = coop4_load_sym_entry(Pfs + sp.t_start * 16, r, q);
    if (sp.j_first == 0 && sp.j_own == 0) {
        if (lane < 16) Pss[(T - 1) * 16 + lane] = Pfs[(T - 1) * 16 + lane];      // filters_smoothers.py:140-142, verbatim copy
        if (lane < 4) mss[(T - 1) * 4 + lane] = mfs[(T - 1) * 4 + lane];
    }

    for (int64_t j = sp.j_first; j <= sp.j_last; j++) {
        const int64_t t_hi = T - 2 - 64 * j;
        const bool own = j >= sp.j_own;
        if constexpr (SPLIT) { if (j == sp.j_own) smoother_junction_store(io, trial, seg, r, b, q, ms, Ps); }
        const OobWindow& oP = own ? wP : wPnull;
        const OobWindow& om = own ? wm : wmnull;
        const int nsteps = t_hi + 1 < 64 ? (int)(t_hi + 1) : 64;
        coop4_chunk_gains(gbuf, lane, nsteps, t_hi, mfs, Pfs, gamma);
        for (int slot = 0; slot < nsteps; slot++) {
            const unsigned t = (unsigned)(t_hi - slot);
            const double* gl = gbuf + slot * kGainPitch;
            const double Gd = gl[r * 4 + q];                             // G[r][q]: A operand G[k][r'], B operand G[k][q']
            const double mf = gl[16 + q];

            double tm = ms, am = 0.0, km, tP = Ps, aP = 0.0, kP;
#pragma unroll 1
            for (int stage = 0; stage < 4; stage++) {
                cd4_mfma_rhs<TWO>(model, R, K, grp, tm, tP, km, kP);    // (_m, _P), _P includes + gamma
                const double drow = mfma4x4(tm - mf, K.ident, 0.0);     // m - mf from column to row form: [r][q] <- [q][r]
                km = mfma4x4(drow, Gd, km);                             // _m + G^T (m - mf): sum_k (m - mf)[k] G[k][q]
                const double sym = mfma4x4(Gd, tP, mfma4x4(tP, Gd, 0.0));                           // G^T P + P G (P symmetric)
                kP = (kP + sym) - 2.0 * K.gam;
                const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
                const double dth = (stage == 2) ? dt : 0.5 * dt;                 // (dt k) half == k (dt half): half is a power of two
                am = fma(wgt, km, am); tm = ms + dth * km;
                aP = fma(wgt, kP, aP);
                tP = Ps + dth * kP;
            }
            ms = ms + (dt * am) * kSixth;
            Ps = Ps + (dt * aP) * kSixth;
            oP.store(Ps, t * 128u + offP);
            om.store(ms, t * 32u + offm);
        }
        wave_lds_fence();
    }
}

// Fix-up pass of a time-split smoother launch (one wavefront per trial): junction_err[b] = the largest relative mismatch, over the junctions of
// trial b, between the state a segment's burn-in arrived at and the row the segment before it (later in time) wrote there -- max |difference| /
// max |reference| over the mean and, separately, the covariance; inf if a NaN sits at a junction.  A heuristic like the filters' (include/chirpgp_hip.h).
template <int UNIT = 0>      // (a template: the header is included by more than one translation unit)
__global__ void __launch_bounds__(64) smoother_split_fixup_kernel(SmootherIO io, double* __restrict__ junction_err) {
    const int64_t trial = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int64_t T = io.T;
    double worst = 0.0;
    for (int s = 1; s < io.bsegs; s++) {
        const int64_t j_own = (int64_t)s * io.chunks_per_bseg;
        if (j_own >= (T - 1 + 63) / 64) break;
        const int64_t row = T - 1 - 64 * j_own;                          // the row both sides hold: segment s - 1 wrote it, segment s arrived at it
        const double* __restrict__ jn = io.junction + (trial * io.bsegs + s) * 20;
        const double* __restrict__ mr = io.mss + (trial * T + row) * 4;
        const double* __restrict__ Pr = io.Pss + (trial * T + row) * 16;
        double dm = 0.0, rm = 0.0, dp = 0.0, rp = 0.0;
        bool nan = false;
        for (int i = 0; i < 4; i++) { const double e = fabs(jn[i] - mr[i]); nan = nan || !(e == e); dm = fmax(dm, e); rm = fmax(rm, fabs(mr[i])); }
        for (int i = 0; i < 16; i++) { const double e = fabs(jn[4 + i] - Pr[i]); nan = nan || !(e == e); dp = fmax(dp, e); rp = fmax(rp, fabs(Pr[i])); }
        double err = fmax(rm > 0.0 ? dm / rm : dm, rp > 0.0 ? dp / rp : dp);
        if (nan) err = __builtin_inf();
        worst = fmax(worst, err);
    }
    junction_err[trial] = worst;
}

// ------------------------------------------------------------------------------------------------ cd_ekf / cd_eks, d = 4
// dm = a(m), dP = J P + P J^T + gamma (filters_smoothers.py:384-394) and the smoother's dm = a(m) + G^T (m - mf),
// dP = A P + P A^T - gamma, A = J + G^T (filters_smoothers.py:427-438) in the same layout: the drift Jacobian
//     J_a = [[-lam, -w, -dw u1, 0], [w, -lam, dw u0, 0], [0, 0, 0, 1], [0, 0, -g^2, -2 g]]    
// is held TRANSPOSED, one entry per lane (lane (r, b, q): J[q][r], i.e. the A operand "J[r'][k]"), assembled from the
// wave-uniform (w, dw) with per-lane 0 / +-1 / constant coefficients; then J P is ONE matrix instruction and
// (J P)^T + J P + gamma another (B = identity, the accumulator carries J P + gamma).  G enters A^T as the lane's own G[r][q].
struct Cd4JacCoef {
    double cb, cg0, cg1, ck;           // J[q][r] = cb w + cg0 jv0 + cg1 jv1 + ck
    CGP_DEV void init(int r, int q, double lam, double g) {
        cb = (q == 0 && r == 1) ? -1.0 : (q == 1 && r == 0) ? 1.0 : 0.0;
        cg0 = (q == 0 && r == 2) ? 1.0 : 0.0; cg1 = (q == 1 && r == 2) ? 1.0 : 0.0;
        ck = ((q == 0 && r == 0) || (q == 1 && r == 1)) ? -lam : (q == 2 && r == 3) ? 1.0 : (q == 3 && r == 2) ? -(g * g) : (q == 3 && r == 3) ? -2.0 * g : 0.0;
    }
};
// a(m) in column form and the lane's entry of J^T at the mean (column form in, quad broadcasts).
CGP_DEV void cd4_ekf_eval(const SoftplusRegs& R, const Cd4LaneCoef& K, const Cd4JacCoef& Jc, double fs, double mcol, double& acol, double& JT) {
    const double m0 = row_bcast_f64<0>(mcol), m1 = row_bcast_f64<1>(mcol), m2 = row_bcast_f64<2>(mcol), m3 = row_bcast_f64<3>(mcol);
    double sp, dsp;
    softplus_pair_uniform(R, m2, sp, dsp);
    const double w = (kTwoPi * sp) * fs, dw = (kTwoPi * dsp) * fs;
    acol = fma(fma(K.k1, w, K.kl0), m0, fma(K.kn, w, K.kl1) * m1) + fma(K.cc2, m2, K.cc3 * m3);
    const double jv0 = -dw * m1, jv1 = dw * m0;
    JT = fma(Jc.cb, w, fma(Jc.cg0, jv0, fma(Jc.cg1, jv1, Jc.ck)));
}

#ifndef CGP_NO_CD_EKF_KERNELS       // the two non-template kernels below live in ONE translation unit (cgp_inst_mfma4.hip)
__global__ void __launch_bounds__(64) cdekf4_mfma_kernel(FilterIO io, ModelArgs ma) {
    __shared__ double2 park[64];                                         // (S, innovation) of the chunk's steps, for the NLL
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    HarmonicSDE<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    const double W0[2] = {0.0, 0.0};
    Cd4LaneCoef K;
    K.init(r, q, W0, model.lam, model.gam, coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, r, q));
    Cd4JacCoef Jc;
    Jc.init(r, q, model.lam, model.gam);
    SoftplusRegs R;
    R.init();
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const double Hk = Hp[r], Hq = Hp[q];
    const double Xi = io.Xi[trial * io.Xi_stride];
    const double dt = ma.dt, fs = model.fs;

    double u = io.m0[trial * io.m0_stride + q];                          // the mean in column form
    double P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, r, q);
    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    OobWindow wP, wm;
    wP.init(io.Pfs ? io.Pfs + trial * T * 16 : nullptr, T * 128);
    wm.init(io.mfs ? io.mfs + trial * T * 4 : nullptr, T * 32);
    const unsigned offP = (b == 0) ? (unsigned)(4 * r + q) * 8u : kOobOffset;
    const unsigned offm = (lane < 4) ? (unsigned)lane * 8u : kOobOffset;
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
            const double y = readlane_f64(ychunk, slot);
            double tm = u, am = 0.0, tP = P, aP = 0.0;
#pragma unroll 1
            for (int stage = 0; stage < 4; stage++) {
                double km, JT;
                cd4_ekf_eval(R, K, Jc, fs, tm, km, JT);
                const double JP = mfma4x4(JT, tP, 0.0);                  // sum_k J[r][k] P[k][q]
                const double kP = mfma4x4(JP, K.ident, JP + K.gam);      // (J P)^T + J P + gamma
                const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
                const double dth = (stage == 2) ? dt : 0.5 * dt;                 // (dt k) half == k (dt half): half is a power of two
                am = fma(wgt, km, am); tm = u + dth * km;
                aP = fma(wgt, kP, aP);
                tP = P + dth * kP;
            }
            const double f = u + (dt * am) * kSixth;
            const double Pp = P + (dt * aP) * kSixth;
            double S, innov;
            mfma4_update_col(Pp, f, Hk, Hq, Xi, y, P, u, S, innov);
            park[slot] = make_double2(S, innov);
            wP.store(P, t * 128u + offP);
            wm.store(u, t * 32u + offm);
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

template <bool SPLIT = false>      // SPLIT: cgp_smoother_time_split, as cdsgps4_mfma_kernel
__global__ void __launch_bounds__(64) cdeks4_mfma_kernel(SmootherIO io, ModelArgs ma) {
    __shared__ __attribute__((aligned(16))) double gbuf[64 * kGainPitch];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int64_t trial = SPLIT ? (int64_t)(blockIdx.x / (unsigned)io.bsegs) : (int64_t)blockIdx.x;
    const int seg = SPLIT ? (int)(blockIdx.x % (unsigned)io.bsegs) : 0;
    if (trial >= io.B) return;

    HarmonicSDE<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.model_id);
    const double W0[2] = {0.0, 0.0};
    Cd4LaneCoef K;
    K.init(r, q, W0, model.lam, model.gam, coop4_load_sym_entry(ma.gamma + trial * ma.gamma_stride, r, q));
    Cd4JacCoef Jc;
    Jc.init(r, q, model.lam, model.gam);
    SoftplusRegs R;
    R.init();
    Sym<4> gamma;
    load_sym<4>(ma.gamma + trial * ma.gamma_stride, gamma);
    const double dt = -ma.dt, fs = model.fs;

    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * 4;
    const double* __restrict__ Pfs = io.Pfs + trial * T * 16;
    double* __restrict__ mss = io.mss + trial * T * 4;
    double* __restrict__ Pss = io.Pss + trial * T * 16;
    OobWindow wP, wm, wPnull, wmnull;
    wP.init(Pss, T * 128);
    wm.init(mss, T * 32);
    wPnull.init(nullptr, 0); wmnull.init(nullptr, 0);                    // the burn-in chunks of a segment store through these
    const unsigned offP = (b == 0) ? (unsigned)(4 * r + q) * 8u : kOobOffset;
    const unsigned offm = (lane < 4) ? (unsigned)lane * 8u : kOobOffset;

    const SmootherSpan sp = smoother_span<SPLIT>(io, seg);
    if (sp.empty) return;
    double ms = mfs[sp.t_start * 4 + q];
    double Ps = coop4_load_sym_entry(Pfs + sp.t_start * 16, r, q);
    if (sp.j_first == 0 && sp.j_own == 0) {
        if (lane < 16) Pss[(T - 1) * 16 + lane] = Pfs[(T - 1) * 16 + lane];      // filters_smoothers.py:140-142, verbatim copy
        if (lane < 4) mss[(T - 1) * 4 + lane] = mfs[(T - 1) * 4 + lane];
    }

    for (int64_t j = sp.j_first; j <= sp.j_last; j++) {
        const int64_t t_hi = T - 2 - 64 * j;
        const bool own = j >= sp.j_own;
        if constexpr (SPLIT) { if (j == sp.j_own) smoother_junction_store(io, trial, seg, r, b, q, ms, Ps); }
        const OobWindow& oP = own ? wP : wPnull;
        const OobWindow& om = own ? wm : wmnull;
        const int nsteps = t_hi + 1 < 64 ? (int)(t_hi + 1) : 64;
        coop4_chunk_gains(gbuf, lane, nsteps, t_hi, mfs, Pfs, gamma);  // Pf^{-1} gamma of the chunk's steps, one step per lane
        for (int slot = 0; slot < nsteps; slot++) {
            const unsigned t = (unsigned)(t_hi - slot);
            const double* gl = gbuf + slot * kGainPitch;
            const double Gd = gl[r * 4 + q];                             // G[r][q] = (G^T)[q][r]: the lane's entry of A^T - J^T
            const double mf = gl[16 + q];
            double tm = ms, am = 0.0, tP = Ps, aP = 0.0;
#pragma unroll 1
            for (int stage = 0; stage < 4; stage++) {
                double km, JT;
                cd4_ekf_eval(R, K, Jc, fs, tm, km, JT);
                const double drow = mfma4x4(tm - mf, K.ident, 0.0);     // m - mf from column to row form
                km = mfma4x4(drow, Gd, km);                             // a(m) + G^T (m - mf)
                const double AP = mfma4x4(JT + Gd, tP, 0.0);            // sum_k A[r][k] P[k][q], A = J + G^T
                const double kP = mfma4x4(AP, K.ident, AP - K.gam);     // (A P)^T + A P - gamma
                const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
                const double dth = (stage == 2) ? dt : 0.5 * dt;                 // (dt k) half == k (dt half): half is a power of two
                am = fma(wgt, km, am); tm = ms + dth * km;
                aP = fma(wgt, kP, aP);
                tP = Ps + dth * kP;
            }
            ms = ms + (dt * am) * kSixth;
            Ps = Ps + (dt * aP) * kSixth;
            oP.store(Ps, t * 128u + offP);
            om.store(ms, t * 32u + offm);
        }
        wave_lds_fence();
    }
}

inline int launch_cdekf4_mfma(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (io.T * 128 > kOobMaxBytes) return CGP_E_UNSUPPORTED;
    hipLaunchKernelGGL(cdekf4_mfma_kernel, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}
inline int launch_cdeks4_mfma(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (io.T * 128 > kOobMaxBytes) return CGP_E_UNSUPPORTED;
    if (io.bsegs > 1) {
        if (io.B * io.bsegs > 0x7fffffffLL) return CGP_E_UNSUPPORTED;
        hipLaunchKernelGGL(cdeks4_mfma_kernel<true>, dim3((unsigned)(io.B * io.bsegs)), dim3(64), 0, stream, io, ma);
    } else {
        hipLaunchKernelGGL(cdeks4_mfma_kernel<false>, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    }
    return hip_rc(hipGetLastError());
}
#endif  // CGP_NO_CD_EKF_KERNELS

template <class SM>
inline int launch_cdsgp4_mfma(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (!sgp4_mfma_fits(io, ma)) return CGP_E_UNSUPPORTED;
    const bool two = ma.sg.n_groups > 16;
    if (io.segs > 1) {                                                              // time-split: one wavefront per (trial, segment)
        const unsigned grid = (unsigned)(io.B * io.segs);
        if (two) hipLaunchKernelGGL((cdsgp4_mfma_kernel<SM, true, true>), dim3(grid), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
        else hipLaunchKernelGGL((cdsgp4_mfma_kernel<SM, false, true>), dim3(grid), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    } else if (two) hipLaunchKernelGGL((cdsgp4_mfma_kernel<SM, true, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    else hipLaunchKernelGGL((cdsgp4_mfma_kernel<SM, false, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    return hip_rc(hipGetLastError());
}
template <class SM>
inline int launch_cdsgps4_mfma(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (!collapsed_ok(ma) || io.T * 128 > kOobMaxBytes) return CGP_E_UNSUPPORTED;
    if (io.bsegs > 1) {                                                             // time-split with burn-in: one wavefront per (trial, segment)
        const unsigned grid = (unsigned)(io.B * io.bsegs);
        if (ma.sg.n_groups > 16) hipLaunchKernelGGL((cdsgps4_mfma_kernel<SM, true, true>), dim3(grid), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
        else hipLaunchKernelGGL((cdsgps4_mfma_kernel<SM, false, true>), dim3(grid), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
        return hip_rc(hipGetLastError());
    }
    if (ma.sg.n_groups > 16) hipLaunchKernelGGL((cdsgps4_mfma_kernel<SM, true>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    else hipLaunchKernelGGL((cdsgps4_mfma_kernel<SM, false>), dim3((unsigned)io.B), dim3(64), sigma_lds_bytes(ma, 4), stream, io, ma);
    return hip_rc(hipGetLastError());
}
// (the fix-up pass of a split launch: instantiated with the kernel, called by cgp_smoother_time_split)
inline int launch_smoother_split_fixup(const SmootherIO& io, double* junction_err, hipStream_t stream) {
    hipLaunchKernelGGL((smoother_split_fixup_kernel<0>), dim3((unsigned)io.B), dim3(64), 0, stream, io, junction_err);
    return hip_rc(hipGetLastError());
}

}  // namespace cgp
