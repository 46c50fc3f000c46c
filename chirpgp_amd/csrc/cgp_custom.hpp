// cgp_custom.hpp -- a model slot for the generic kernels, filled at run time (hiprtc; cgp_rtc.hip, cgp_model_from_source).
//
// The reference's filters take ANY JAX-traceable callable -- cond_m_cov(u, dt), a(u) -- and differentiate it with jax.jacfwd
// (filters_smoothers.py:255, 342, 382, 425); the enumerated models of cgp_models.hpp cover the reference's own builders and nothing
// else (its Lorenz-63 test, test/test_ekfs.py:11-62, could not run on the HIP path: VERDICT r5 missing #5).  Here the caller hands the
// model as a few lines of device source,
//     template <class T> __device__ void cond_mean(const T* u, const double* p, double dt, T* mean);      // discrete: mean of cond_m_cov
//     __device__ void cond_cov(const double* u, const double* p, double dt, double* cov);                 //   its covariance, [d][d] row-major
//     template <class T> __device__ void drift(const T* u, const double* p, T* a);                        // SDE: a(u)   (b b^T arrives as `gamma`)
// written once for a generic scalar T: the kernels instantiate it with double for values and with Dual<d> -- forward-mode dual numbers,
// the in-kernel counterpart of jacfwd -- for the Jacobian, exact to rounding.  CustomDisc / CustomSDE adapt it to the interface the
// generic lane-per-trial kernels of cgp_kernels.hpp expect of a model (LinearDisc / LinearSDE), and the runtime-compiled program
// instantiates filter_kernel<EkfPredict<..>>, smoother_kernel<EksStep<..>>, filter_kernel<CdEkfPredict<..>>, smoother_kernel<CdEksStep<..>>
// on them -- ekf, eks, cd_ekf, cd_eks for any model of dimension <= 8, same arithmetic as the compiled-in models -- and the sigma-point
// methods: sgp_filter / sgp_smoother through a literal fan (SgpPredictCustom: the covariance evaluated at every point), cd_sgp_filter /
// cd_sgp_smoother through the generic CdSgpPredict / CdSgpsStep.
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {
namespace ad {

// value + N partial derivatives; every operation below propagates them by the chain rule
template <int N> struct Dual {
    double v;
    double d[N];
    __device__ Dual() {}
    __device__ Dual(double c) : v(c) { CGP_UNROLL for (int i = 0; i < N; i++) d[i] = 0.0; }
};
#define CGP_DUAL_LOOP CGP_UNROLL for (int i = 0; i < N; i++)
template <int N> CGP_DEV Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; CGP_DUAL_LOOP r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> CGP_DEV Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; CGP_DUAL_LOOP r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> CGP_DEV Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; CGP_DUAL_LOOP r.d[i] = fma(a.v, b.d[i], a.d[i] * b.v); return r; }
template <int N> CGP_DEV Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; const double ib = 1.0 / b.v; r.v = a.v * ib; CGP_DUAL_LOOP r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r;
}
template <int N> CGP_DEV Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; CGP_DUAL_LOOP r.d[i] = -a.d[i]; return r; }
template <int N> CGP_DEV Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> CGP_DEV Dual<N> operator+(double a, const Dual<N>& b) { return b + a; }
template <int N> CGP_DEV Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> CGP_DEV Dual<N> operator-(double a, const Dual<N>& b) { Dual<N> r; r.v = a - b.v; CGP_DUAL_LOOP r.d[i] = -b.d[i]; return r; }
template <int N> CGP_DEV Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; CGP_DUAL_LOOP r.d[i] = a.d[i] * b; return r; }
template <int N> CGP_DEV Dual<N> operator*(double a, const Dual<N>& b) { return b * a; }
template <int N> CGP_DEV Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> CGP_DEV Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> CGP_DEV Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> CGP_DEV Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> CGP_DEV Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> CGP_DEV Dual<N>& operator+=(Dual<N>& a, double b) { a.v += b; return a; }
template <int N> CGP_DEV Dual<N>& operator*=(Dual<N>& a, double b) { a = a * b; return a; }
// f(a) with derivative df: the chain rule for a function of one argument
template <int N> CGP_DEV Dual<N> chain(const Dual<N>& a, double f, double df) { Dual<N> r; r.v = f; CGP_DUAL_LOOP r.d[i] = df * a.d[i]; return r; }
template <int N> CGP_DEV Dual<N> sin(const Dual<N>& a) { double s, c; ::sincos(a.v, &s, &c); return chain(a, s, c); }
template <int N> CGP_DEV Dual<N> cos(const Dual<N>& a) { double s, c; ::sincos(a.v, &s, &c); return chain(a, c, -s); }
template <int N> CGP_DEV Dual<N> exp(const Dual<N>& a) { const double e = ::exp(a.v); return chain(a, e, e); }
template <int N> CGP_DEV Dual<N> log(const Dual<N>& a) { return chain(a, ::log(a.v), 1.0 / a.v); }
template <int N> CGP_DEV Dual<N> sqrt(const Dual<N>& a) { const double r = ::sqrt(a.v); return chain(a, r, 0.5 / r); }
template <int N> CGP_DEV Dual<N> tanh(const Dual<N>& a) { const double t = ::tanh(a.v); return chain(a, t, 1.0 - t * t); }
template <int N> CGP_DEV Dual<N> pow(const Dual<N>& a, double e) { const double p = ::pow(a.v, e - 1.0); return chain(a, p * a.v, e * p); }
// the reference's positive bijection g(x) = log(exp(x) + 1) (models.py:50, the naive form) and its derivative
CGP_DEV double softplus(double x) { return ::log(::exp(x) + 1.0); }
template <int N> CGP_DEV Dual<N> softplus(const Dual<N>& a) { const double e = ::exp(a.v), z = e + 1.0; return chain(a, ::log(z), e / z); }
#undef CGP_DUAL_LOOP
// the plain functions under the same names, so that a body written for a generic T finds them for T = double
using ::sin; using ::cos; using ::exp; using ::log; using ::sqrt; using ::tanh; using ::pow;

}  // namespace ad

// cond_m_cov(u, dt) -> (mean, cov) supplied as U::mean<T> / U::cov: the discrete-model interface of LinearDisc (cgp_models.hpp)
template <int D_, class U> struct CustomDisc {
    static constexpr int D = D_;
    const double* __restrict__ p = nullptr;
    double dt = 0.0;
    bool uniform = false, wide = false, large_batch = false;
    CGP_DEV void setup(const double* __restrict__ params, double dt_, int /*model_id*/) { p = params; dt = dt_; }
    CGP_DEV void mean(const Vec<D>& u, Vec<D>& f) const { U::template mean<double>(u.v, p, dt, f.v); }
    CGP_DEV void mean_and_cov(const Vec<D>& u, Vec<D>& f, double (&cov)[D_ * D_]) const { U::template mean<double>(u.v, p, dt, f.v); U::cov(u.v, p, dt, cov); }
    // f = mean(u), T = J P with J = d mean / d u by dual numbers (jacfwd, filters_smoothers.py:255), Pp = T J^T + cov(u)
    CGP_DEV void propagate(const Vec<D>& u, const Sym<D>& P, Vec<D>& f, Mat<D>& T, Sym<D>& Pp) const {
        ad::Dual<D> x[D], y[D];
        CGP_UNROLL for (int i = 0; i < D; i++) { x[i] = ad::Dual<D>(u.v[i]); x[i].d[i] = 1.0; }
        U::template mean<ad::Dual<D>>(x, p, dt, y);
        Mat<D> J;
        CGP_UNROLL for (int i = 0; i < D; i++) { f.v[i] = y[i].v; CGP_UNROLL for (int j = 0; j < D; j++) J.a[i][j] = y[i].d[j]; }
        double cov[D * D];
        U::cov(u.v, p, dt, cov);
        Sym<D> Sigma;
        CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j <= i; j++) Sigma(i, j) = cov[i * D + j];      // the lower triangle, like load_sym
        mul_dense_sym<D>(J, P, T);
        mul_nt_sym_add<D>(T, J, Sigma, Pp);
    }
};

// Sigma-point prediction of a custom discrete model, filters_smoothers.py:88-121 as written -- one lane walks the whole point set
// (read from global memory), the model's covariance is evaluated AT EVERY POINT (the enumerated models' is constant; a
// caller's need not be) -- plus (CROSS) the smoother's D^T of :525.
template <class DM, bool CROSS>
CGP_DEV void sgp_prediction_literal(const DM& model, const SigmaSet& sg, const Vec<DM::D>& mf, const Sym<DM::D>& Pf,
                                    Vec<DM::D>& mp, Sym<DM::D>& Pp, Mat<DM::D>& DT) {
    constexpr int D = DM::D;
    Sym<D> L; Vec<D> inv;
    cholesky<D>(Pf, L, inv);
    CGP_UNROLL for (int i = 0; i < D; i++) mp.v[i] = 0.0;
    CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) Pp.a[i] = 0.0;
    Mat<D> X;
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) X.a[i][j] = 0.0;
    for (int p = 0; p < sg.s; p++) {
        Vec<D> chi, f;
        sigma_point<D, false>(mf, L, sg, p, chi);
        double cov[D * D];
        model.mean_and_cov(chi, f, cov);
        const double w = sg.template weight<false>(p);
        CGP_UNROLL for (int i = 0; i < D; i++) {
            const double wf = w * f.v[i];
            mp.v[i] += wf;
            CGP_UNROLL for (int j = 0; j <= i; j++) Pp(i, j) = fma(wf, f.v[j], fma(w, cov[i * D + j], Pp(i, j)));
            if (CROSS) { CGP_UNROLL for (int k = 0; k < D; k++) X.a[k][i] = fma(chi.v[k], wf, X.a[k][i]); }
        }
    }
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j <= i; j++) Pp(i, j) -= mp.v[i] * mp.v[j];
    if (CROSS) { CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) DT.a[j][i] = X.a[i][j] - mf.v[i] * mp.v[j]; }
}
// sgp_filter (filters_smoothers.py:446-490) / sgp_smoother (:493-531) on a custom model, one lane per trial
template <class DM> struct SgpPredictCustom {
    static constexpr bool USES_SIGMA = true;
    static constexpr bool LANE_TWO_WAVES = false;
    static constexpr int D = DM::D; static constexpr bool WAVE = false; static constexpr bool USES_LDS = false;
    DM model; SigmaSet sg;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); sg = a.sg; }
    CGP_DEV void predict(int, double*, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& mp, Sym<D>& Pp) const {
        Mat<D> unused;
        sgp_prediction_literal<DM, false>(model, sg, mf, Pf, mp, Pp, unused);
    }
};
template <class DM> struct SgpsStepCustom {
    static constexpr bool USES_SIGMA = true;
    static constexpr int D = DM::D; static constexpr bool WAVE = false; static constexpr bool USES_LDS = false;
    DM model; SigmaSet sg;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); sg = a.sg; }
    CGP_DEV void step(int, double*, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& ms, Sym<D>& Ps) const {
        Vec<D> mp; Sym<D> Pp; Mat<D> DT, G;
        sgp_prediction_literal<DM, true>(model, sg, mf, Pf, mp, Pp, DT);
        smoother_gain<D>(DT, Pp, G);
        smoother_apply<D>(G, mf, Pf, mp, Pp, ms, Ps);
    }
};

// ekf_for_kpt's measurement function h(u) supplied as U::measure<T> (filters_smoothers.py:298-311: H = jacfwd(h)(mp), pred = h(mp)): the
// interface of KptUpdate (cgp_kernels.hpp).  The vector the linear filters read as H arrives as the body's q.
template <int D_, class U> struct CustomMeasurement {
    static constexpr int D = D_;
    static constexpr bool LINEAR = false;
    CGP_DEV static void update(const Vec<D>& mp, const Sym<D>& Pp, const Vec<D>& q, double Xi, double y, Vec<D>& mf, Sym<D>& Pf,
                               double& S, double& innov) {
        ad::Dual<D> x[D];
        CGP_UNROLL for (int i = 0; i < D; i++) { x[i] = ad::Dual<D>(mp.v[i]); x[i].d[i] = 1.0; }
        const ad::Dual<D> h = U::template measure<ad::Dual<D>>(x, q.v);
        Vec<D> H;
        CGP_UNROLL for (int i = 0; i < D; i++) H.v[i] = h.d[i];
        scalar_update<D>(mp, Pp, H, Xi, y, true, h.v, mf, Pf, S, innov);
    }
};

// SDE drift a(u) supplied as U::drift<T>: the interface of LinearSDE (cgp_models.hpp)
template <int D_, class U> struct CustomSDE {
    static constexpr int D = D_;
    const double* __restrict__ p = nullptr;
    bool uniform = false, wide = false;
    CGP_DEV void setup(const double* __restrict__ params, int /*model_id*/) { p = params; }
    CGP_DEV void drift(const Vec<D>& u, Vec<D>& a) const { U::template drift<double>(u.v, p, a.v); }
    // sigma-point interface (cd_sgp_common, cgp_steps.hpp): nothing to share between the points of a group
    static constexpr int IVC = 0;
    struct Pre {};
    CGP_DEV void precompute(double, Pre&) const {}
    CGP_DEV void drift_pre(const Vec<D>& u, const Pre&, Vec<D>& a) const { drift(u, a); }
    CGP_DEV void drift_jac(const Vec<D>& u, Vec<D>& a, Mat<D>& J) const {
        ad::Dual<D> x[D], y[D];
        CGP_UNROLL for (int i = 0; i < D; i++) { x[i] = ad::Dual<D>(u.v[i]); x[i].d[i] = 1.0; }
        U::template drift<ad::Dual<D>>(x, p, y);
        CGP_UNROLL for (int i = 0; i < D; i++) { a.v[i] = y[i].v; CGP_UNROLL for (int j = 0; j < D; j++) J.a[i][j] = y[i].d[j]; }
    }
    CGP_DEV void drift_jp(const Vec<D>& u, const Sym<D>& P, Vec<D>& a, Mat<D>& T) const {
        Mat<D> J;
        drift_jac(u, a, J);
        mul_dense_sym<D>(J, P, T);
    }
};

}  // namespace cgp
