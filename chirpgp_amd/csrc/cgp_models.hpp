// cgp_models.hpp -- device-side models of the path with hand-derived Jacobians (SURVEY.md notes N1, N2).
//
// Discrete models  (the reference's cond_m_cov(u, dt) closures, chirpgp/models.py:264-311, 332-386, 419-434):
//     mean(u)                          conditional mean f(u)
//     propagate(u, P, f, T, Pp)        f(u), T = J(u) P, Pp = T J(u)^T + Sigma   (EKF predict; T is the smoother's DT)
//     add_sigma(S, w)                  S += w * Sigma
// SDE models       (drift a(u), models.py:104-110, 164-168; constant dispersion passed as gamma = b b^T):
//     drift(u, a), drift_jp(u, P, a, T)   a(u), T = J_a(u) P,  jac_dense(u, J)
//
// The harmonic family has J = blockdiag(rho Rot(k theta), M32) + one dense column (d/d u_v), which the products below
// exploit: ~4.5 d^2 FMAs instead of 4 d^3.
#pragma once
#include "cgp_math.hpp"
#include "cgp_fastmath.hpp"

namespace cgp {

// softplus log(exp(x) + 1) (models.py:50) and its derivative exp(x) / (exp(x) + 1), per lane, for the latency-bound kernels.
// Common regime, 1.5 <= x < 700 (frequencies above 1.7 Hz):  x + t q(t)  and  1 / (1 + t)  with t = exp(-x) <= 0.223 and
// q = log1p(t) / t, both by the LEAN degree-7 polynomials of cgp_fastmath.hpp (relative error 7e-12 on the softplus,
// 1.2e-11 on the derivative: these kernels' results sat eight orders inside the 1e-5 gate) -- 23 instructions instead of
// the 63 of exp + full log.  If ANY active lane is outside that regime (or NaN) the wavefront also evaluates the
// reference's naive form, whose overflow behaviour (inf, NaN) is the reference's, and those lanes take it: one
// wave-uniform branch, not taken in the common case.
CGP_DEV double exp_neg_lean_lane(double x) {
    const double nx = -x;
    const double k = __builtin_rint(nx * kLog2e);
    double r = fma(-k, kLn2Hi, nx);
    r = fma(-k, kLn2Lo, r);
    const double r2 = r * r;
    const double a0 = horner(kExpLean[1], r, kExpLean[0]), a1 = horner(kExpLean[3], r, kExpLean[2]);
    const double a2 = horner(kExpLean[5], r, kExpLean[4]), a3 = horner(kExpLean[7], r, kExpLean[6]);
    const double r4 = r2 * r2;
    return __builtin_amdgcn_ldexp(horner(horner(a3, r2, a2), r4, horner(a1, r2, a0)), (int)k);
}
CGP_DEV double log1p_over_t_lean(double t) {
    const double t2 = t * t;
    const double a0 = horner(kLog1pOverTLean[1], t, kLog1pOverTLean[0]), a1 = horner(kLog1pOverTLean[3], t, kLog1pOverTLean[2]);
    const double a2 = horner(kLog1pOverTLean[5], t, kLog1pOverTLean[4]), a3 = horner(kLog1pOverTLean[7], t, kLog1pOverTLean[6]);
    const double t4 = t2 * t2;
    return horner(horner(a3, t2, a2), t4, horner(a1, t2, a0));
}
// the three forms of a sigma-point fan's softplus -> sin / cos chain in the one-wavefront kernels (precompute / precompute_spec /
// precompute_any below): spec first; if a lane is outside its regime, any; checked as the last resort
constexpr int kFanChecked = 0, kFanSpec = 1, kFanAny = 2;
CGP_DEV bool softplus_lane_common(double x) { return x >= 1.5 && x < 700.0; }
// The lean softplus on BOTH sides (round 5; the sigma-point fans, which need no derivative): with t = exp(-|x|) <= 0.223 for |x| >= 1.5,
//     softplus(x) = max(x, 0) + log1p(t) = max(x, 0) + t q(t)
// -- for x <= -1.5 the same polynomials in the same t-range as for x >= 1.5, one v_max_f64 off the chain.  Records the filter has lost sit at
// a NEGATIVE frequency state most of the time (tools/state_histogram.py): their fans stay in the lean tier.
CGP_DEV bool softplus_lane_lean2(double x) { return fabs(x) >= 1.5 && fabs(x) < 700.0; }
CGP_DEV void softplus_pair_wide(double x, double& sp, double& dsp) {
    const double t = exp_neg_lean_lane(x);
    sp = fma(log1p_over_t_lean(t), t, x);
    dsp = rcp_nr1(1.0 + t);
    const bool common = softplus_lane_common(x);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!common) != 0, 0)) {
        const double e = fast_exp(x);
        const double z = e + 1.0;
        const double sp_n = fast_log_ge1(z), dsp_n = e * rcp_nr(z);     // inf * NaN = NaN where the reference has inf / inf = NaN
        sp = common ? sp : sp_n;
        dsp = common ? dsp : dsp_n;
    }
}
CGP_DEV double softplus_wide(double x) {
    const double t = exp_neg_lean_lane(x);
    double sp = fma(log1p_over_t_lean(t), t, x);
    const bool common = softplus_lane_common(x);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!common) != 0, 0)) {
        const double sp_n = fast_log_ge1(fast_exp(x) + 1.0);
        sp = common ? sp : sp_n;
    }
    return sp;
}
// The naive form as is: fewer live registers than the pair above (no second path, no 16 polynomial coefficients), which
// is what counts in the one-lane-per-trial kernels -- they live on occupancy, and the wide form costs them a wave per SIMD
// (measured: lane-per-trial EKS 2.3 -> 4.6 ms at B = 65536).
CGP_DEV void softplus_pair(double x, double& sp, double& dsp) {
    const double e = fast_exp(x);
    const double z = e + 1.0;
    sp = fast_log_ge1(z);
    dsp = e * rcp_nr(z);            // inf * NaN = NaN where the reference has inf / inf = NaN
}
CGP_DEV double softplus(double x) { return fast_log_ge1(fast_exp(x) + 1.0); }
// The naive form for the large-batch lane kernels (cgp_lane4.hpp), which are bound by their vector instruction count: exp and log without
// their overflow / underflow / inf selects and with the direct exponent extraction (cgp_fastmath.hpp: fast_log_ge1_finite) -- the same
// values for x < 700 (an ulp in the log where the mantissa interval's ends are met), 12 instructions less; if ANY lane is at or beyond
// 700 (or NaN) the wavefront also evaluates softplus_pair and those lanes take it, overflow behaviour and all.
CGP_DEV double softplus_batch(double x) {
    double sp = fast_log_ge1_finite(fast_exp_core(x) + 1.0);
    const bool regular = x < 700.0;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!regular) != 0, 0)) {
        const double sp_n = softplus(x);
        sp = regular ? sp : sp_n;
    }
    return sp;
}
CGP_DEV void softplus_pair_batch(double x, double& sp, double& dsp) {
    const double e = fast_exp_core(x);
    const double z = e + 1.0;
    sp = fast_log_ge1_finite(z);
    dsp = e * rcp_nr(z);
    const bool regular = x < 700.0;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!regular) != 0, 0)) {
        double sp_n, dsp_n;
        softplus_pair(x, sp_n, dsp_n);
        sp = regular ? sp : sp_n;
        dsp = regular ? dsp : dsp_n;
    }
}
// `uniform` = the argument is the same in all lanes of the wavefront (a wave-per-trial kernel evaluating the model at the
// trial's mean): one scalar branch then picks the cheap large-x form (cgp_fastmath.hpp).  `wide` = the kernel is latency-
// rather than occupancy-bound (one wavefront per trial, time-parallel smoother): take the wide common-regime form.
CGP_DEV void softplus_pair_sel(bool uniform, bool wide, double x, double& sp, double& dsp) {
    if (uniform) softplus_pair_uniform(x, sp, dsp);
    else if (wide) softplus_pair_wide(x, sp, dsp);
    else softplus_pair(x, sp, dsp);
}
CGP_DEV double softplus_sel(bool wide, double x) { return wide ? softplus_wide(x) : softplus(x); }

// Closed-form Matern-3/2 discretisation, models.py:61-73.
CGP_DEV void m32_solution(double ell, double sigma, double dt, double (&M)[4], double (&S)[3]) {
    const double gamma = sqrt(3.0) / ell, eta = dt * gamma;
    const double beta = sigma * sigma * exp(-2.0 * eta), e = exp(-eta);
    M[0] = (1.0 + eta) * e; M[1] = dt * e; M[2] = -dt * gamma * gamma * e; M[3] = (1.0 - eta) * e;
    S[0] = sigma * sigma - beta * (2.0 * eta + 2.0 * eta * eta + 1.0);
    S[1] = 2.0 * dt * dt * gamma * gamma * gamma * beta;
    S[2] = gamma * gamma * (sigma * sigma + beta * (2.0 * eta - 2.0 * eta * eta - 1.0));
}

// ------------------------------------------------------------------------------------------------ linear discrete
template <int D_> struct LinearDisc {
    static constexpr int D = D_;
    Mat<D> F;
    Sym<D> Sigma;
    bool uniform = false, wide = false;
    CGP_DEV void setup(const double* __restrict__ p, double /*dt*/, int /*model_id*/) {
        load_mat<D>(p, F);
        load_sym<D>(p + D * D, Sigma);
    }
    // sigma-point interface: nothing to share between the points of a group
    static constexpr int IVC = 0;
    struct Pre {};
    struct Anchor {};
    CGP_DEV void anchor(double, Anchor&) const {}
    CGP_DEV void precompute(double, Pre&) const {}
    CGP_DEV void precompute(double, const Anchor&, Pre&) const {}
    CGP_DEV void mean_pre(const Vec<D>& u, const Pre&, Vec<D>& f) const { matvec<D>(F, u, f); }
    CGP_DEV void mean(const Vec<D>& u, Vec<D>& f) const { matvec<D>(F, u, f); }
    CGP_DEV void propagate(const Vec<D>& u, const Sym<D>& P, Vec<D>& f, Mat<D>& T, Sym<D>& Pp) const {
        matvec<D>(F, u, f);
        mul_dense_sym<D>(F, P, T);
        mul_nt_sym_add<D>(T, F, Sigma, Pp);
    }
    CGP_DEV void add_sigma(Sym<D>& S, double w) const {
        CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) S.a[i] = fma(w, Sigma.a[i], S.a[i]);
    }
};

// The linear dynamics of the KPT model (models.py:561-570): F = I + e_{d-1} e_0^T -- the phase accumulates the frequency --
// passed like any linear model as dense (F, Sigma).  When F has exactly that form (every build_kpt_chirp_model result) the
// prediction is d + 1 additions instead of two dense d x d x d products (250 multiply-adds at d = 5, most of a step of
// ekf_for_kpt); any other F takes LinearDisc's dense path.  The choice is uniform over the wavefront / per lane's trial.
template <int D_> struct KptLinear : LinearDisc<D_> {
    static constexpr int D = D_;
    using Base = LinearDisc<D_>;
    bool shift = false;
    CGP_DEV void setup(const double* __restrict__ p, double dt, int model_id) {
        Base::setup(p, dt, model_id);
        bool ok = true;
        CGP_UNROLL for (int i = 0; i < D; i++)
            CGP_UNROLL for (int j = 0; j < D; j++) ok = ok && (this->F.a[i][j] == ((i == j || (i == D - 1 && j == 0)) ? 1.0 : 0.0));
        shift = ok;
    }
    CGP_DEV void propagate(const Vec<D>& u, const Sym<D>& P, Vec<D>& f, Mat<D>& T, Sym<D>& Pp) const {
        if (!shift) { Base::propagate(u, P, f, T, Pp); return; }
        f = u;
        f.v[D - 1] = u.v[D - 1] + u.v[0];
        // T = F P: row d - 1 += row 0;  Pp = T F^T + Sigma: column d - 1 += column 0
        CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) T.a[i][j] = P(i, j) + ((i == D - 1) ? P(0, j) : 0.0);
        CGP_UNROLL for (int i = 0; i < D; i++)
            CGP_UNROLL for (int j = 0; j <= i; j++) {
                double v = T.a[i][j];
                if (i == D - 1 && j == D - 1) v += T.a[D - 1][0];
                Pp(i, j) = v + this->Sigma(i, j);
            }
    }
};

// ------------------------------------------------------------------------------------------------ harmonic chirp, LCD
// disc_harmonic_chirp_lcd (models.py:332-386); NH = 1, freq_scale = 1 is disc_chirp_lcd (models.py:264-311);
// model_id CGP_M_LASCALA_LCD (params = ell, sigma) is disc_model_lascala_lcd (models.py:419-434): rho = 1, q = 0.
template <int NH> struct HarmonicLCD {
    static constexpr int D = 2 * NH + 2;
    static constexpr int IV = D - 2;
    double rho, q, fs, dt;
    double M[4], MS[3];
    double MM[9];              // the quadratic form M C M^T of the symmetric 2 x 2 block C written out (propagate_blocks): set by setup_blocks()
    CGP_DEV void setup_blocks() {
        MM[0] = M[0] * M[0]; MM[1] = 2.0 * (M[0] * M[1]); MM[2] = M[1] * M[1];
        MM[3] = M[0] * M[2]; MM[4] = fma(M[0], M[3], M[1] * M[2]); MM[5] = M[1] * M[3];
        MM[6] = M[2] * M[2]; MM[7] = 2.0 * (M[2] * M[3]); MM[8] = M[3] * M[3];
    }
    bool uniform = false;      // set by wave-per-trial EKF-type callers: propagate() then sees a wave-uniform state
    bool wide = false;         // set by latency-bound callers (wave per trial, time-parallel smoother): softplus_pair_sel
    bool large_batch = false; // set by the large-batch lane kernels (cgp_lane4.hpp): fast_sincos_small in rotations(), softplus_batch / softplus_pair_batch
    CGP_DEV void setup(const double* __restrict__ p, double dt_, int model_id) {
        dt = dt_;
        if (model_id == 2 /* CGP_M_LASCALA_LCD */) {
            rho = 1.0; q = 0.0; fs = 1.0;
            m32_solution(p[0], p[1], dt, M, MS);
        } else {
            const double lam = p[0], b = p[1];
            fs = p[4];
            rho = exp(-lam * dt);
            q = (lam == 0.0) ? b * b * dt : b * b / (2.0 * lam) * (1.0 - exp(-2.0 * lam * dt));   // models.py:302-308
            m32_solution(p[2], p[3], dt, M, MS);
        }
    }
    // cos/sin of dt*k*w scaled by rho, for every harmonic: one sincos for the fundamental, the angle-addition
    // recurrence for the overtones (same values as cos(dt k w), sin(dt k w) up to rounding).
    CGP_DEV void rotations(double w, double (&c)[NH], double (&s)[NH]) const {
        double s1, c1;
        if (large_batch) fast_sincos_small(dt * w, s1, c1);
        else fast_sincos(dt * w, s1, c1);
        double ck = c1, sk = s1;
        c[0] = c1 * rho; s[0] = s1 * rho;
        CGP_UNROLL for (int k = 1; k < NH; k++) {
            const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
            ck = cn; sk = sn;
            c[k] = ck * rho; s[k] = sk * rho;
        }
    }
    // sigma-point interface: the rotations depend on u_v only, so points that share chi_v share them
    static constexpr int IVC = IV;
    struct Pre { double c[NH], s[NH]; };
    CGP_DEV void precompute(double uv, Pre& p) const { rotations((kTwoPi * softplus_sel(wide, uv)) * fs, p.c, p.s); }
    // precompute() WITHOUT its two regime branches and with pinned coefficients, for kernels that interleave several fans
    // in one basic block: the lean softplus and the reduced-range sin / cos taken as is.  ok = the lane is in the regime
    // where that is valid (1.5 <= uv < 700 -- |uv| for the multi-harmonic models -- and a rotation angle within pi/4, i.e. n = 0 in the Cody-Waite reduction); the
    // caller re-evaluates with precompute() if any lane is not.
    CGP_DEV void precompute_spec(const FanRegs& R, double uv, Pre& p, bool& ok) const {
        // (the two-sided lean form, softplus_lane_lean2, for the multi-harmonic models only: the d = 8 filter on records outside the
        // common regime 12.7 -> 10.6 ms; the d = 4 matrix-core filter, whose fan straddles the band's end anyway, LOST 3 % to it)
        constexpr bool TWO_SIDED = NH > 1;
        const double t = exp_neg_lean(R, TWO_SIDED ? fabs(uv) : uv);
        double q, unused;
        softplus_tail_lean(R, t, q, unused);
        const double sp = fma(q, t, TWO_SIDED ? fmax(uv, 0.0) : uv);
        const double x = sp * ((kTwoPi * fs) * dt);           // loop-invariant factor: one multiplication on the chain
        ok = (TWO_SIDED ? softplus_lane_lean2(uv) : softplus_lane_common(uv)) && fabs(x) <= kPiOver4;
        double s1, c1;
        sincos_reduced(R, x, s1, c1);
        double ck = c1, sk = s1;
        p.c[0] = c1 * rho; p.s[0] = s1 * rho;
        CGP_UNROLL for (int k = 1; k < NH; k++) {
            const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
            ck = cn; sk = sn;
            p.c[k] = ck * rho; p.s[k] = sk * rho;
        }
    }
    // precompute_spec() with the branch-free full-accuracy softplus for ANY |uv| < 700 (cgp_fastmath.hpp: softplus_pair_any) in place of
    // the lean one: the middle tier of the one-wavefront sigma-point kernels (round 5) for records whose frequency state drops below
    // 1.5, where the lean form does not hold -- they used to take precompute() with its two regime branches behind every speculative
    // fan.  ok = |uv| < 700 and a rotation angle within pi / 4.
    CGP_DEV void precompute_any(const FanRegs& R, double uv, Pre& p, bool& ok) const {
        double sp, unused; bool ok1;
        softplus_pair_any(uv, sp, unused, ok1);
        const double x = sp * ((kTwoPi * fs) * dt);
        ok = ok1 && fabs(x) <= kPiOver4;
        double s1, c1;
        sincos_reduced(R, x, s1, c1);
        double ck = c1, sk = s1;
        p.c[0] = c1 * rho; p.s[0] = s1 * rho;
        CGP_UNROLL for (int k = 1; k < NH; k++) {
            const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
            ck = cn; sk = sn;
            p.c[k] = ck * rho; p.s[k] = sk * rho;
        }
    }
    // The sigma points of one prediction spread around the mean, so their rotation angles differ from the mean's by a
    // small d = dt (w - w0): the fan anchors (cos, sin) at the mean once and every point takes the small-angle rotation
    // by d (sin to d^9, cos to d^8: remainders < 3e-19 for |d| <= 2^-4) instead of a full sincos -- 14 instead of ~45
    // instructions per group of points.  If any lane's d is larger (or NaN) the wavefront also evaluates the full
    // sincos and those lanes take it.
    struct Anchor { double w0, c1, s1; };
    CGP_DEV void anchor(double uv0, Anchor& a) const {
        a.w0 = (kTwoPi * (large_batch ? softplus_batch(uv0) : softplus_sel(wide, uv0))) * fs;
        fast_sincos(dt * a.w0, a.s1, a.c1);
    }
    CGP_DEV void precompute(double uv, const Anchor& a, Pre& p) const {
        const double w = (kTwoPi * (large_batch ? softplus_batch(uv) : softplus_sel(wide, uv))) * fs;
        const double d = dt * (w - a.w0), d2 = d * d;
        const double ps = fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
        const double cd = fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
        const double sd = d * ps;
        double c1 = fma(a.c1, cd, -a.s1 * sd), s1 = fma(a.s1, cd, a.c1 * sd);
        const bool small = fabs(d) <= 0.0625;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!small) != 0, 0)) {
            double sf, cf;
            fast_sincos(dt * w, sf, cf);
            c1 = small ? c1 : cf;
            s1 = small ? s1 : sf;
        }
        double ck = c1, sk = s1;
        p.c[0] = c1 * rho; p.s[0] = s1 * rho;
        CGP_UNROLL for (int k = 1; k < NH; k++) {
            const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
            ck = cn; sk = sn;
            p.c[k] = ck * rho; p.s[k] = sk * rho;
        }
    }
    // The anchored form with the FULL-accuracy softplus for finite arguments and no branch (round 5: the large-batch sigma-point lane
    // kernel on records outside the lean regime -- a body without branches is what lets the scheduler interleave two groups' chains):
    // ok = false for uv >= 700 (the naive form overflows from 709.78 on) or a rotation angle beyond pi / 4; the caller repeats the fan
    // with precompute().
    CGP_DEV void precompute_any(double uv, const Anchor&, Pre& p, bool& ok) const {
        const double w = (kTwoPi * fast_log_ge1_finite(fast_exp_core(uv) + 1.0)) * fs;
        const double x = dt * w;
        double s1, c1;
        sincos_reduced(x, s1, c1);                    // (not the anchored small-angle form: on the CRLB records a fan is wider than 1/16 rad)
        ok = uv < 700.0 && fabs(x) <= kPiOver4;
        double ck = c1, sk = s1;
        p.c[0] = c1 * rho; p.s[0] = s1 * rho;
        CGP_UNROLL for (int k = 1; k < NH; k++) {
            const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
            ck = cn; sk = sn;
            p.c[k] = ck * rho; p.s[k] = sk * rho;
        }
    }
    // The anchored form without its two regime branches (lean softplus as is, small-angle rotation as is), for a lane
    // that walks many groups in a loop the scheduler should see as one block: ok = false where precompute(uv, a, p)
    // would have taken a fallback (uv outside [1.5, 700) or |d| > 1/16); the caller then repeats the fan with that.
    CGP_DEV void precompute_spec(double uv, const Anchor& a, Pre& p, bool& ok) const {
        const double t = exp_neg_lean_lane(uv);
        const double w = (kTwoPi * fma(log1p_over_t_lean(t), t, uv)) * fs;
        const double d = dt * (w - a.w0), d2 = d * d;
        const double ps = fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
        const double cd = fma(d2, fma(d2, fma(d2, fma(d2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
        const double sd = d * ps;
        const double c1 = fma(a.c1, cd, -a.s1 * sd), s1 = fma(a.s1, cd, a.c1 * sd);
        ok = softplus_lane_common(uv) && fabs(d) <= 0.0625;
        double ck = c1, sk = s1;
        p.c[0] = c1 * rho; p.s[0] = s1 * rho;
        CGP_UNROLL for (int k = 1; k < NH; k++) {
            const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
            ck = cn; sk = sn;
            p.c[k] = ck * rho; p.s[k] = sk * rho;
        }
    }
    CGP_DEV void mean_pre(const Vec<D>& u, const Pre& p, Vec<D>& f) const {
        CGP_UNROLL for (int k = 0; k < NH; k++) {
            f.v[2 * k] = p.c[k] * u.v[2 * k] - p.s[k] * u.v[2 * k + 1];
            f.v[2 * k + 1] = p.s[k] * u.v[2 * k] + p.c[k] * u.v[2 * k + 1];
        }
        f.v[IV] = M[0] * u.v[IV] + M[1] * u.v[IV + 1];
        f.v[IV + 1] = M[2] * u.v[IV] + M[3] * u.v[IV + 1];
    }
    CGP_DEV void mean(const Vec<D>& u, Vec<D>& f) const {
        Pre p;
        precompute(u.v[IV], p);
        mean_pre(u, p, f);
    }
    CGP_DEV void propagate(const Vec<D>& u, const Sym<D>& P, Vec<D>& f, Mat<D>& T, Sym<D>& Pp) const {
        bool ok;
        propagate_impl<false>(u, P, f, T, Pp, ok);
    }
    // the same without the regime branches of the softplus and the sin / cos (lean softplus as is, reduced sin / cos as is):
    // ok = false where propagate() would have taken a fallback; the caller then repeats with propagate()
    CGP_DEV void propagate_spec(const Vec<D>& u, const Sym<D>& P, Vec<D>& f, Mat<D>& T, Sym<D>& Pp, bool& ok) const {
        propagate_impl<true>(u, P, f, T, Pp, ok);
    }
    template <bool SPEC>
    CGP_DEV void propagate_impl(const Vec<D>& u, const Sym<D>& P, Vec<D>& f, Mat<D>& T, Sym<D>& Pp, bool& ok) const {
        double sp, dsp;
        double c[NH], s[NH], jv[2 * NH];
        double w, dw;
        if constexpr (SPEC) {
            const double x = u.v[IV];
            const double t = exp_neg_lean_lane(x);
            sp = fma(log1p_over_t_lean(t), t, x);
            dsp = rcp_nr1(1.0 + t);
            w = (kTwoPi * sp) * fs; dw = (kTwoPi * dsp) * fs;
            double s1, c1; bool ok2;
            fast_sincos_spec(dt * w, s1, c1, ok2);
            ok = softplus_lane_common(x) && ok2;
            double ck = c1, sk = s1;
            c[0] = c1 * rho; s[0] = s1 * rho;
            CGP_UNROLL for (int k = 1; k < NH; k++) {
                const double cn = fma(ck, c1, -sk * s1), sn = fma(sk, c1, ck * s1);
                ck = cn; sk = sn;
                c[k] = ck * rho; s[k] = sk * rho;
            }
        } else {
            ok = true;
            softplus_pair_sel(uniform, wide, u.v[IV], sp, dsp);
            w = (kTwoPi * sp) * fs; dw = (kTwoPi * dsp) * fs;
            rotations(w, c, s);
        }
        CGP_UNROLL for (int k = 0; k < NH; k++) {
            const double u0 = u.v[2 * k], u1 = u.v[2 * k + 1];
            f.v[2 * k] = c[k] * u0 - s[k] * u1;
            f.v[2 * k + 1] = s[k] * u0 + c[k] * u1;
            const double dth = (dt * (double)(k + 1)) * dw;
            jv[2 * k] = dth * (-s[k] * u0 - c[k] * u1);       // d f_{2k} / d u_v     (N1)
            jv[2 * k + 1] = dth * (c[k] * u0 - s[k] * u1);    // d f_{2k+1} / d u_v
        }
        f.v[IV] = M[0] * u.v[IV] + M[1] * u.v[IV + 1];
        f.v[IV + 1] = M[2] * u.v[IV] + M[3] * u.v[IV + 1];
        // T = J P
        CGP_UNROLL for (int j = 0; j < D; j++) {
            CGP_UNROLL for (int k = 0; k < NH; k++) {
                const double p0 = P(2 * k, j), p1 = P(2 * k + 1, j), pv = P(IV, j);
                T.a[2 * k][j] = fma(jv[2 * k], pv, fma(c[k], p0, -s[k] * p1));
                T.a[2 * k + 1][j] = fma(jv[2 * k + 1], pv, fma(s[k], p0, c[k] * p1));
            }
            T.a[IV][j] = fma(M[0], P(IV, j), M[1] * P(IV + 1, j));
            T.a[IV + 1][j] = fma(M[2], P(IV, j), M[3] * P(IV + 1, j));
        }
        // Pp = T J^T + Sigma, lower triangle: (T J^T)(i, j) = sum_l T(i, l) J(j, l)
        CGP_UNROLL for (int i = 0; i < D; i++)
            CGP_UNROLL for (int j = 0; j <= i; j++) {
                double v;
                if (j < IV) {
                    const int k = j / 2;
                    if (j % 2 == 0) v = fma(jv[j], T.a[i][IV], fma(c[k], T.a[i][2 * k], -s[k] * T.a[i][2 * k + 1]));
                    else            v = fma(jv[j], T.a[i][IV], fma(s[k], T.a[i][2 * k], c[k] * T.a[i][2 * k + 1]));
                    if (i == j) v += q;
                } else if (j == IV) {
                    v = fma(M[0], T.a[i][IV], M[1] * T.a[i][IV + 1]) + (i == IV ? MS[0] : MS[1]);
                } else {
                    v = fma(M[2], T.a[i][IV], M[3] * T.a[i][IV + 1]) + MS[2];
                }
                Pp(i, j) = v;
            }
    }
    // The filter's prediction alone (no cross-covariance out) for ONE harmonic, by the blocks of J = [R g e_1^T; 0 M] and P = [A B; B^T C]
    // (R = rho * rotation, g = d f / d u_v, M the Matern-3/2 transition):
    //     X = R B + g C_0.,   Y = R A + g B_.0^T,   Pp_AA = Y R^T + X_.0 g^T + q I,   Pp_AC = X M^T,   Pp_CC = M C M^T + Sigma_C
    // with g = dth (-f_1, f_0) taken from the rotated mean, M C M^T as three 3-term quadratic forms with precomputed products (MM) and
    // every constant addend as the start of its FMA chain: 50 + 12 float64 operations where propagate() spends 72 + 22 on T = J P and T J^T (round 5: the large-batch lane EKF is bound by its
    // float64 instruction count, cgp_lane4.hpp).  Same quantities up to rounding (sums associate differently: ~1e-16).
    CGP_DEV void propagate_blocks(const Vec<D>& u, const Sym<D>& P, Vec<D>& f, Sym<D>& Pp) const {
        static_assert(NH == 1, "block form written out for one harmonic");
        double sp, dsp, c[1], s[1];
        if (large_batch) softplus_pair_batch(u.v[2], sp, dsp);
        else softplus_pair_sel(uniform, wide, u.v[2], sp, dsp);
        const double wdt = (kTwoPi * fs) * dt;
        double s1, c1;
        if (large_batch) fast_sincos_small(wdt * sp, s1, c1);
        else fast_sincos(wdt * sp, s1, c1);
        c[0] = c1 * rho; s[0] = s1 * rho;
        const double cc = c[0], ss = s[0];
        f.v[0] = fma(cc, u.v[0], -(ss * u.v[1]));
        f.v[1] = fma(ss, u.v[0], cc * u.v[1]);
        f.v[2] = fma(M[0], u.v[2], M[1] * u.v[3]);
        f.v[3] = fma(M[2], u.v[2], M[3] * u.v[3]);
        const double dth = wdt * dsp;
        const double g0 = -(dth * f.v[1]), g1 = dth * f.v[0];
        const double A00 = P(0, 0), A10 = P(1, 0), A11 = P(1, 1);
        const double B00 = P(2, 0), B01 = P(3, 0), B10 = P(2, 1), B11 = P(3, 1);
        const double C00 = P(2, 2), C10 = P(3, 2), C11 = P(3, 3);
        const double X00 = fma(g0, C00, fma(cc, B00, -(ss * B10))), X01 = fma(g0, C10, fma(cc, B01, -(ss * B11)));
        const double X10 = fma(g1, C00, fma(ss, B00, cc * B10)),    X11 = fma(g1, C10, fma(ss, B01, cc * B11));
        const double Y00 = fma(g0, B00, fma(cc, A00, -(ss * A10))), Y01 = fma(g0, B10, fma(cc, A10, -(ss * A11)));
        const double Y10 = fma(g1, B00, fma(ss, A00, cc * A10)),    Y11 = fma(g1, B10, fma(ss, A10, cc * A11));
        Pp(0, 0) = fma(g0, X00, fma(cc, Y00, fma(-ss, Y01, q)));
        Pp(1, 0) = fma(g0, X10, fma(cc, Y10, -(ss * Y11)));
        Pp(1, 1) = fma(g1, X10, fma(ss, Y10, fma(cc, Y11, q)));
        Pp(2, 0) = fma(M[0], X00, M[1] * X01);
        Pp(3, 0) = fma(M[2], X00, M[3] * X01);
        Pp(2, 1) = fma(M[0], X10, M[1] * X11);
        Pp(3, 1) = fma(M[2], X10, M[3] * X11);
        Pp(2, 2) = fma(MM[0], C00, fma(MM[1], C10, fma(MM[2], C11, MS[0])));
        Pp(3, 2) = fma(MM[3], C00, fma(MM[4], C10, fma(MM[5], C11, MS[1])));
        Pp(3, 3) = fma(MM[6], C00, fma(MM[7], C10, fma(MM[8], C11, MS[2])));
    }
    CGP_DEV void add_sigma(Sym<D>& S, double w) const {
        CGP_UNROLL for (int i = 0; i < IV; i++) S(i, i) = fma(w, q, S(i, i));
        S(IV, IV) = fma(w, MS[0], S(IV, IV));
        S(IV + 1, IV) = fma(w, MS[1], S(IV + 1, IV));
        S(IV + 1, IV + 1) = fma(w, MS[2], S(IV + 1, IV + 1));
    }
};

// ------------------------------------------------------------------------------------------------ linear SDE
template <int D_> struct LinearSDE {
    static constexpr int D = D_;
    Mat<D> A;
    bool uniform = false, wide = false;
    CGP_DEV void setup(const double* __restrict__ p, int /*model_id*/) { load_mat<D>(p, A); }
    static constexpr int IVC = 0;
    struct Pre {};
    CGP_DEV void precompute(double, Pre&) const {}
    CGP_DEV void drift_pre(const Vec<D>& u, const Pre&, Vec<D>& a) const { matvec<D>(A, u, a); }
    CGP_DEV void drift(const Vec<D>& u, Vec<D>& a) const { matvec<D>(A, u, a); }
    CGP_DEV void drift_jp(const Vec<D>& u, const Sym<D>& P, Vec<D>& a, Mat<D>& T) const {
        matvec<D>(A, u, a);
        mul_dense_sym<D>(A, P, T);
    }
    CGP_DEV void drift_jac(const Vec<D>& u, Vec<D>& a, Mat<D>& J) const { matvec<D>(A, u, a); J = A; }
};

// ------------------------------------------------------------------------------------------------ harmonic chirp SDE
// model_harmonic_chirp drift (models.py:164-168); NH = 1 is model_chirp (models.py:104-110); lam = 0 is model_lascala.
template <int NH> struct HarmonicSDE {
    static constexpr int D = 2 * NH + 2;
    static constexpr int IV = D - 2;
    double lam, gam, fs;
    bool uniform = false, wide = false;
    CGP_DEV void setup(const double* __restrict__ p, int /*model_id*/) { lam = p[0]; gam = sqrt(3.0) / p[1]; fs = p[2]; }
    static constexpr int IVC = IV;
    struct Pre { double w; };
    CGP_DEV void precompute(double uv, Pre& p) const { p.w = (kTwoPi * softplus_sel(wide, uv)) * fs; }
    // precompute() without the regime branch and with pinned coefficients (see HarmonicLCD::precompute_spec): ok = the
    // lane is where the lean softplus is valid, 1.5 <= uv < 700; the caller re-evaluates with precompute() otherwise.
    CGP_DEV void precompute_spec(const SoftplusRegs& R, double uv, Pre& p, bool& ok) const {
        const double t = exp_neg_lean(R, uv);
        double q, unused;
        softplus_tail_lean(R, t, q, unused);
        p.w = (kTwoPi * fma(q, t, uv)) * fs;
        ok = softplus_lane_common(uv);
    }
    // ... and with the branch-free full-accuracy softplus for any |uv| < 700 (HarmonicLCD::precompute_any): the middle tier
    CGP_DEV void precompute_any(const SoftplusRegs&, double uv, Pre& p, bool& ok) const {
        double sp, unused;
        softplus_pair_any(uv, sp, unused, ok);
        p.w = (kTwoPi * sp) * fs;
    }
    CGP_DEV void drift(const Vec<D>& u, Vec<D>& a) const {
        Pre p;
        precompute(u.v[IV], p);
        drift_pre(u, p, a);
    }
    CGP_DEV void drift_pre(const Vec<D>& u, const Pre& p, Vec<D>& a) const {
        const double w = p.w;
        CGP_UNROLL for (int k = 0; k < NH; k++) {
            const double wk = w * (double)(k + 1);
            a.v[2 * k] = -lam * u.v[2 * k] - wk * u.v[2 * k + 1];
            a.v[2 * k + 1] = wk * u.v[2 * k] - lam * u.v[2 * k + 1];
        }
        a.v[IV] = u.v[IV + 1];
        a.v[IV + 1] = -(gam * gam) * u.v[IV] - 2.0 * gam * u.v[IV + 1];
    }
    // a(u) and the non-trivial Jacobian pieces: wk[k] and the d/du_v column jv (N2)
    CGP_DEV void pieces(const Vec<D>& u, Vec<D>& a, double (&wk)[NH], double (&jv)[2 * NH]) const {
        double sp, dsp;
        softplus_pair_sel(uniform, wide, u.v[IV], sp, dsp);
        const double w = (kTwoPi * sp) * fs, dw = (kTwoPi * dsp) * fs;
        CGP_UNROLL for (int k = 0; k < NH; k++) {
            wk[k] = w * (double)(k + 1);
            const double dwk = dw * (double)(k + 1);
            a.v[2 * k] = -lam * u.v[2 * k] - wk[k] * u.v[2 * k + 1];
            a.v[2 * k + 1] = wk[k] * u.v[2 * k] - lam * u.v[2 * k + 1];
            jv[2 * k] = -dwk * u.v[2 * k + 1];
            jv[2 * k + 1] = dwk * u.v[2 * k];
        }
        a.v[IV] = u.v[IV + 1];
        a.v[IV + 1] = -(gam * gam) * u.v[IV] - 2.0 * gam * u.v[IV + 1];
    }
    CGP_DEV void drift_jp(const Vec<D>& u, const Sym<D>& P, Vec<D>& a, Mat<D>& T) const {
        double wk[NH], jv[2 * NH];
        pieces(u, a, wk, jv);
        CGP_UNROLL for (int j = 0; j < D; j++) {
            CGP_UNROLL for (int k = 0; k < NH; k++) {
                const double p0 = P(2 * k, j), p1 = P(2 * k + 1, j), pv = P(IV, j);
                T.a[2 * k][j] = fma(jv[2 * k], pv, fma(-lam, p0, -wk[k] * p1));
                T.a[2 * k + 1][j] = fma(jv[2 * k + 1], pv, fma(wk[k], p0, -lam * p1));
            }
            T.a[IV][j] = P(IV + 1, j);
            T.a[IV + 1][j] = fma(-(gam * gam), P(IV, j), -2.0 * gam * P(IV + 1, j));
        }
    }
    CGP_DEV void drift_jac(const Vec<D>& u, Vec<D>& a, Mat<D>& J) const {
        double wk[NH], jv[2 * NH];
        pieces(u, a, wk, jv);
        CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) J.a[i][j] = 0.0;
        CGP_UNROLL for (int k = 0; k < NH; k++) {
            J.a[2 * k][2 * k] = -lam; J.a[2 * k][2 * k + 1] = -wk[k];
            J.a[2 * k + 1][2 * k] = wk[k]; J.a[2 * k + 1][2 * k + 1] = -lam;
            J.a[2 * k][IV] = jv[2 * k]; J.a[2 * k + 1][IV] = jv[2 * k + 1];
        }
        J.a[IV][IV + 1] = 1.0;
        J.a[IV + 1][IV] = -(gam * gam);
        J.a[IV + 1][IV + 1] = -2.0 * gam;
    }
};

// ------------------------------------------------------------------------------------------------ KPT measurement
// h(x) = sum_k x[k] sin(k g(x[0] + x[d-1])) and its gradient (models.py:575-578).
// UNIFORM: every lane of the wavefront evaluates the same argument (one wavefront per trial): the wave-uniform softplus and
// sincos take their regime branches on scalars instead of diverging per lane.  The harmonics k = 2 .. NH come from the
// fundamental's pair by the angle-addition recurrence (two dependent operations each) instead of NH range reductions.
template <int NH, bool UNIFORM = false> struct KptMeasurement {
    static constexpr int D = NH + 2;
    CGP_DEV static double eval(const Vec<D>& x, Vec<D>& H) {
        double gs, dgs, s1, c1;
        if constexpr (UNIFORM) { softplus_pair_uniform(x.v[0] + x.v[D - 1], gs, dgs); fast_sincos_uniform(gs, s1, c1); }
        else { softplus_pair(x.v[0] + x.v[D - 1], gs, dgs); fast_sincos(gs, s1, c1); }
        double h = 0.0, dsum = 0.0, sn = s1, cs = c1;
        CGP_UNROLL for (int i = 0; i < D; i++) H.v[i] = 0.0;
        CGP_UNROLL for (int k = 1; k <= NH; k++) {
            if (k > 1) {                                              // (sin, cos)(k g) from ((k - 1) g) and (g)
                const double sk = fma(sn, c1, cs * s1), ck = fma(cs, c1, -(sn * s1));
                sn = sk; cs = ck;
            }
            h = fma(x.v[k], sn, h);
            H.v[k] = sn;
            dsum = fma(x.v[k] * cs, (double)k * dgs, dsum);
        }
        H.v[0] += dsum;
        H.v[D - 1] += dsum;
        return h;
    }
};

}  // namespace cgp
