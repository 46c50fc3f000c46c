/*
 * port.c -- plain-C float64 restatement of the chirpgp filtering / smoothing path (TEST INFRASTRUCTURE).
 *
 * Second, independently written implementation next to the NumPy oracle (oracle/np_*.py): dense runtime-d
 * loops, one trial per OpenMP iteration.  It is (i) diffed against the NumPy oracle (tests/test_oracle_port.py),
 * (ii) the checker of the GPU parity tests at sizes the NumPy oracle is too slow for, and (iii) the timed
 * host-CPU baseline of bench.py ("cpu_baseline", kind "port").  Nothing in chirpgp_amd/ links or calls it.
 *
 * It takes the same argument structures as include/chirpgp_hip.h, with HOST pointers.
 * Follows /root/reference/chirpgp/filters_smoothers.py (fs), quadratures.py (qd), models.py (md).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../../include/chirpgp_hip.h"

#define MAXD 16
#define TWO_PI 6.283185307179586476925286766559

/* The checker build keeps the state dimension a run-time value.  The TIMED build of bench.py's cpu_baseline
 * (oracle/port.py: build_native) is compiled with -DFIXED_D=<d> -march=native so that every d-loop has a constant trip
 * count and is unrolled / vectorised -- a fair host-CPU number instead of generic runtime-d loops.  Same source, same
 * arithmetic; only the rounding differs when the compiler is allowed to contract a*b+c (tests/test_oracle_port.py). */
#ifdef FIXED_D
#define DIM(x) (FIXED_D)
#else
#define DIM(x) (x)
#endif

typedef struct {
    int id, d, nh;
    /* LINEAR / KPT / LINEAR_SDE */
    const double *F, *Sigma, *A;
    /* harmonic family */
    double lam, fscale, rho, q, M[4], MS[4], gam;
    const double *gamma;
} model_t;

/* ------------------------------------------------------------------------------------------------ helpers */
static double softplus(double x) { return log(exp(x) + 1.0); }                     /* md:50, naive form   */
static double dsoftplus(double x) { double e = exp(x); return e / (e + 1.0); }     /* jacfwd of the above */

static void m32_solution(double ell, double sigma, double dt, double *M, double *S) /* md:61-73 */
{
    double gamma = sqrt(3.0) / ell, eta = dt * gamma, beta = sigma * sigma * exp(-2 * eta), e = exp(-eta);
    M[0] = (1 + eta) * e; M[1] = dt * e; M[2] = -dt * gamma * gamma * e; M[3] = (1 - eta) * e;
    S[0] = sigma * sigma - beta * (2 * eta + 2 * eta * eta + 1);
    S[1] = S[2] = 2 * dt * dt * gamma * gamma * gamma * beta;
    S[3] = gamma * gamma * (sigma * sigma + beta * (2 * eta - 2 * eta * eta - 1));
}

static void model_setup(model_t *m, const cgp_model *cm, int64_t trial, double dt)
{
    const double *p = cm->params + trial * cm->param_stride;
    int d = DIM(cm->d);
    memset(m, 0, sizeof(*m));
    m->id = cm->model_id; m->d = d; m->nh = cm->n_harm;
    m->gamma = cm->gamma ? cm->gamma + trial * cm->gamma_stride : NULL;
    switch (cm->model_id) {
    case CGP_M_LINEAR: case CGP_M_KPT: m->F = p; m->Sigma = p + d * d; break;
    case CGP_M_LINEAR_SDE: m->A = p; break;
    case CGP_M_HARMONIC_LCD: {
        double lam = p[0], b = p[1], ell = p[2], sigma = p[3];
        m->lam = lam; m->fscale = p[4];
        m->rho = exp(-lam * dt);
        m->q = (lam == 0.0) ? b * b * dt : b * b / (2 * lam) * (1 - exp(-2 * lam * dt));     /* md:302-308 */
        m32_solution(ell, sigma, dt, m->M, m->MS);
        break; }
    case CGP_M_LASCALA_LCD:
        m->lam = 0; m->fscale = 1; m->rho = 1; m->q = 0; m->nh = 1;
        m32_solution(p[0], p[1], dt, m->M, m->MS);
        break;
    case CGP_M_HARMONIC_SDE:
        m->lam = p[0]; m->gam = sqrt(3.0) / p[1]; m->fscale = p[2];
        break;
    }
}

/* Discrete model: conditional mean f(u), its Jacobian J (row-major d*d) and covariance Sigma.  md:264-311, 332-386. */
static void disc_eval(const model_t *m, const double *u, double dt, double *f, double *J, double *Sig)
{
    int d = DIM(m->d);
    if (m->id == CGP_M_LINEAR || m->id == CGP_M_KPT) {
        for (int i = 0; i < d; i++) { double s = 0; for (int j = 0; j < d; j++) s += m->F[i * d + j] * u[j]; f[i] = s; }
        if (J) memcpy(J, m->F, sizeof(double) * d * d);
        if (Sig) memcpy(Sig, m->Sigma, sizeof(double) * d * d);
        return;
    }
    int nh = m->nh, iv = d - 2;
    double w = TWO_PI * softplus(u[iv]) * m->fscale;
    double dw = TWO_PI * dsoftplus(u[iv]) * m->fscale;
    if (J) memset(J, 0, sizeof(double) * d * d);
    if (Sig) memset(Sig, 0, sizeof(double) * d * d);
    for (int k = 1; k <= nh; k++) {
        double th = dt * k * w, c = cos(th) * m->rho, s = sin(th) * m->rho;
        int i = 2 * (k - 1);
        f[i] = c * u[i] - s * u[i + 1];
        f[i + 1] = s * u[i] + c * u[i + 1];
        if (J) {
            J[i * d + i] = c; J[i * d + i + 1] = -s; J[(i + 1) * d + i] = s; J[(i + 1) * d + i + 1] = c;
            double dth = dt * k * dw;
            J[i * d + iv] = dth * (-s * u[i] - c * u[i + 1]);
            J[(i + 1) * d + iv] = dth * (c * u[i] - s * u[i + 1]);
        }
        if (Sig) { Sig[i * d + i] = m->q; Sig[(i + 1) * d + i + 1] = m->q; }
    }
    f[iv] = m->M[0] * u[iv] + m->M[1] * u[iv + 1];
    f[iv + 1] = m->M[2] * u[iv] + m->M[3] * u[iv + 1];
    if (J) { J[iv * d + iv] = m->M[0]; J[iv * d + iv + 1] = m->M[1]; J[(iv + 1) * d + iv] = m->M[2]; J[(iv + 1) * d + iv + 1] = m->M[3]; }
    if (Sig) { Sig[iv * d + iv] = m->MS[0]; Sig[iv * d + iv + 1] = m->MS[1]; Sig[(iv + 1) * d + iv] = m->MS[2]; Sig[(iv + 1) * d + iv + 1] = m->MS[3]; }
}

/* SDE drift a(u) and its Jacobian.  md:104-110, 164-168. */
static void sde_eval(const model_t *m, const double *u, double *a, double *Ja)
{
    int d = DIM(m->d);
    if (m->id == CGP_M_LINEAR_SDE) {
        for (int i = 0; i < d; i++) { double s = 0; for (int j = 0; j < d; j++) s += m->A[i * d + j] * u[j]; a[i] = s; }
        if (Ja) memcpy(Ja, m->A, sizeof(double) * d * d);
        return;
    }
    int nh = m->nh, iv = d - 2;
    double w = TWO_PI * softplus(u[iv]) * m->fscale, dw = TWO_PI * dsoftplus(u[iv]) * m->fscale, g = m->gam;
    if (Ja) memset(Ja, 0, sizeof(double) * d * d);
    for (int k = 1; k <= nh; k++) {
        int i = 2 * (k - 1);
        a[i] = -m->lam * u[i] - w * k * u[i + 1];
        a[i + 1] = w * k * u[i] - m->lam * u[i + 1];
        if (Ja) {
            Ja[i * d + i] = -m->lam; Ja[i * d + i + 1] = -w * k; Ja[(i + 1) * d + i] = w * k; Ja[(i + 1) * d + i + 1] = -m->lam;
            Ja[i * d + iv] = -dw * k * u[i + 1];
            Ja[(i + 1) * d + iv] = dw * k * u[i];
        }
    }
    a[iv] = u[iv + 1];
    a[iv + 1] = -(g * g) * u[iv] - 2 * g * u[iv + 1];
    if (Ja) { Ja[iv * d + iv + 1] = 1.0; Ja[(iv + 1) * d + iv] = -(g * g); Ja[(iv + 1) * d + iv + 1] = -2 * g; }
}

static void matmul(int d, const double *A, const double *B, double *C)          /* C = A B   */
{
    d = DIM(d);
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) { double s = 0; for (int k = 0; k < d; k++) s += A[i * d + k] * B[k * d + j]; C[i * d + j] = s; }
}
static void matmul_nt(int d, const double *A, const double *B, double *C)       /* C = A B^T */
{
    d = DIM(d);
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) { double s = 0; for (int k = 0; k < d; k++) s += A[i * d + k] * B[j * d + k]; C[i * d + j] = s; }
}
/* Lower Cholesky; an all-NaN factor on failure (JAX semantics; LAPACK potrf fails on a pivot <= 0 or NaN). */
static void chol_lower(int d, const double *P, double *L)
{
    int bad = 0;
    d = DIM(d);
    memset(L, 0, sizeof(double) * d * d);
    for (int j = 0; j < d && !bad; j++) {
        double s = P[j * d + j];
        for (int k = 0; k < j; k++) s -= L[j * d + k] * L[j * d + k];
        if (!(s > 0.0)) { bad = 1; break; }
        double ljj = sqrt(s);
        L[j * d + j] = ljj;
        for (int i = j + 1; i < d; i++) {
            double t = P[i * d + j];
            for (int k = 0; k < j; k++) t -= L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / ljj;
        }
    }
    if (bad) for (int i = 0; i < d * d; i++) L[i] = NAN;
}
/* X = P^{-1} R for n right-hand sides (R, X row-major d*n) via Cholesky; fs:81-82, 429-431, 617-618. */
static void cho_solve(int d, const double *P, const double *R, int n, double *X)
{
    double L[MAXD * MAXD];
    d = DIM(d);
    chol_lower(d, P, L);
    for (int c = 0; c < n; c++) {
        double y[MAXD];
        for (int i = 0; i < d; i++) { double s = R[i * n + c]; for (int k = 0; k < i; k++) s -= L[i * d + k] * y[k]; y[i] = s / L[i * d + i]; }
        for (int i = d - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < d; k++) s -= L[k * d + i] * X[k * n + c]; X[i * n + c] = s / L[i * d + i]; }
    }
}

/* fs:55-68 with the jax.scipy.stats.norm.logpdf arithmetic (oracle/np_filters.py:_neg_log_normal_pdf). */
static double linear_update(int d, const double *mp, const double *Pp, const double *H, double Xi, double y, double pred_override,
                            int use_override, double *mf, double *Pf)
{
    double PH[MAXD], S = 0, pred = 0;
    d = DIM(d);
    for (int i = 0; i < d; i++) { double s = 0; for (int j = 0; j < d; j++) s += Pp[i * d + j] * H[j]; PH[i] = s; }
    for (int i = 0; i < d; i++) { S += H[i] * PH[i]; pred += H[i] * mp[i]; }
    S += Xi;
    if (use_override) pred = pred_override;
    double K[MAXD];
    for (int i = 0; i < d; i++) K[i] = PH[i] / S;
    for (int i = 0; i < d; i++) mf[i] = mp[i] + K[i] * (y - pred);
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) Pf[i * d + j] = Pp[i * d + j] - (K[i] * K[j]) * S;
    double sc = sqrt(S), s2 = sc * sc;
    return (log(TWO_PI * s2) + (y - pred) * (y - pred) / s2) / 2;
}

/* fs:71-85.  DT (d*d) is the transposed cross-covariance. */
static void smoother_common(int d, const double *DT, const double *mf, const double *Pf, const double *mp, const double *Pp,
                            double *ms, double *Ps)
{
    double X[MAXD * MAXD], G[MAXD * MAXD], dm[MAXD], dP[MAXD * MAXD], T1[MAXD * MAXD], T2[MAXD * MAXD];
    d = DIM(d);
    cho_solve(d, Pp, DT, d, X);
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) G[i * d + j] = X[j * d + i];
    for (int i = 0; i < d; i++) dm[i] = ms[i] - mp[i];
    for (int i = 0; i < d * d; i++) dP[i] = Ps[i] - Pp[i];
    matmul(d, G, dP, T1);
    matmul_nt(d, T1, G, T2);
    for (int i = 0; i < d; i++) { double s = 0; for (int k = 0; k < d; k++) s += G[i * d + k] * dm[k]; ms[i] = mf[i] + s; }
    for (int i = 0; i < d * d; i++) Ps[i] = Pf[i] + T2[i];
}

/* fs:88-121: sigma-point prediction; optionally the cross term D^T for the smoother (fs:525). */
static void sgp_prediction(const model_t *m, const cgp_sigma *sg, double dt, const double *mf, const double *Pf,
                           double *mp, double *Pp, double *DT)
{
    int d = DIM(m->d), s = sg->s;
    double L[MAXD * MAXD], Sig[MAXD * MAXD], chi[MAXD], f[MAXD], second[MAXD * MAXD], cross[MAXD * MAXD];
    chol_lower(d, Pf, L);
    memset(second, 0, sizeof(second)); memset(cross, 0, sizeof(cross));
    for (int i = 0; i < d; i++) mp[i] = 0;
    for (int p = 0; p < s; p++) {
        const double *xi = sg->xi + (size_t)p * d;
        double w = sg->w[p];
        for (int i = 0; i < d; i++) { double t = 0; for (int j = 0; j < d; j++) t += L[i * d + j] * xi[j]; chi[i] = mf[i] + t; }
        disc_eval(m, chi, dt, f, NULL, Sig);
        for (int i = 0; i < d; i++) mp[i] += w * f[i];
        for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
            second[i * d + j] += w * (f[i] * f[j] + Sig[i * d + j]);
            cross[i * d + j] += w * (chi[i] * f[j]);
        }
    }
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) Pp[i * d + j] = second[i * d + j] - mp[i] * mp[j];
    if (DT) for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) DT[j * d + i] = cross[i * d + j] - mf[i] * mp[j];
}

/* fs:124-137. */
static void cd_sgp_common(const model_t *m, const cgp_sigma *sg, const double *mm, const double *P, double *dm, double *dP)
{
    int d = DIM(m->d), s = sg->s;
    double L[MAXD * MAXD], chi[MAXD], a[MAXD], acc[MAXD * MAXD];
    chol_lower(d, P, L);
    memset(acc, 0, sizeof(acc));
    for (int i = 0; i < d; i++) dm[i] = 0;
    for (int p = 0; p < s; p++) {
        const double *xi = sg->xi + (size_t)p * d;
        double w = sg->w[p];
        for (int i = 0; i < d; i++) { double t = 0; for (int j = 0; j < d; j++) t += L[i * d + j] * xi[j]; chi[i] = mm[i] + t; }
        sde_eval(m, chi, a, NULL);
        for (int i = 0; i < d; i++) dm[i] += w * a[i];
        for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) acc[i * d + j] += w * ((chi[i] - mm[i]) * a[j]);
    }
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) dP[i * d + j] = acc[i * d + j] + acc[j * d + i] + m->gamma[i * d + j];
}

/* Moment ODE right-hand sides.  kind: 0 cd_ekf (fs:384-385), 1 cd_sgp (fs:569-570),
 * 2 cd_eks (fs:427-432), 3 cd_sgp_smoother (fs:615-621); PinvG = Pf^{-1} gamma (constant over the 4 stages). */
typedef struct { const model_t *m; const cgp_sigma *sg; int kind; const double *mf; const double *PinvG; } ode_t;

static void ode_rhs(const ode_t *o, const double *mm, const double *P, double *dm, double *dP)
{
    const model_t *m = o->m; int d = DIM(m->d);
    double J[MAXD * MAXD], T1[MAXD * MAXD], T2[MAXD * MAXD];
    if (o->kind == 0) {
        sde_eval(m, mm, dm, J);
        matmul_nt(d, P, J, T1); matmul(d, J, P, T2);
        for (int i = 0; i < d * d; i++) dP[i] = T1[i] + T2[i] + m->gamma[i];
    } else if (o->kind == 1) {
        cd_sgp_common(m, o->sg, mm, P, dm, dP);
    } else if (o->kind == 2) {
        double A[MAXD * MAXD], diff[MAXD], v[MAXD];
        sde_eval(m, mm, dm, J);
        /* A = J + (Pf^{-1} gamma^T)^T = J + gamma Pf^{-1};  PinvG = Pf^{-1} gamma (gamma symmetric) */
        for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) A[i * d + j] = J[i * d + j] + o->PinvG[j * d + i];
        for (int i = 0; i < d; i++) diff[i] = mm[i] - o->mf[i];
        for (int i = 0; i < d; i++) { double s = 0; for (int k = 0; k < d; k++) s += o->PinvG[k * d + i] * diff[k]; v[i] = s; }
        for (int i = 0; i < d; i++) dm[i] += v[i];
        matmul(d, A, P, T1); matmul_nt(d, P, A, T2);
        for (int i = 0; i < d * d; i++) dP[i] = T1[i] + T2[i] - m->gamma[i];
    } else {
        double diff[MAXD];
        cd_sgp_common(m, o->sg, mm, P, dm, dP);
        for (int i = 0; i < d; i++) diff[i] = mm[i] - o->mf[i];
        for (int i = 0; i < d; i++) { double s = 0; for (int k = 0; k < d; k++) s += o->PinvG[k * d + i] * diff[k]; dm[i] += s; }
        /* G^T P + P G with G = PinvG */
        for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < d; k++) { s1 += o->PinvG[k * d + i] * P[k * d + j]; s2 += P[i * d + k] * o->PinvG[k * d + j]; }
            dP[i * d + j] = dP[i * d + j] + s1 + s2 - 2 * m->gamma[i * d + j];
        }
    }
}

/* qd:34-54 / 57-81 */
static void rk4(const ode_t *o, double *mm, double *P, double dt)
{
    int d = DIM(o->m->d), n = d * d;
    double k1m[MAXD], k2m[MAXD], k3m[MAXD], k4m[MAXD], tm[MAXD];
    double k1P[MAXD * MAXD], k2P[MAXD * MAXD], k3P[MAXD * MAXD], k4P[MAXD * MAXD], tP[MAXD * MAXD];
    ode_rhs(o, mm, P, k1m, k1P);
    for (int i = 0; i < d; i++) tm[i] = mm[i] + dt * k1m[i] / 2;
    for (int i = 0; i < n; i++) tP[i] = P[i] + dt * k1P[i] / 2;
    ode_rhs(o, tm, tP, k2m, k2P);
    for (int i = 0; i < d; i++) tm[i] = mm[i] + dt * k2m[i] / 2;
    for (int i = 0; i < n; i++) tP[i] = P[i] + dt * k2P[i] / 2;
    ode_rhs(o, tm, tP, k3m, k3P);
    for (int i = 0; i < d; i++) tm[i] = mm[i] + dt * k3m[i];
    for (int i = 0; i < n; i++) tP[i] = P[i] + dt * k3P[i];
    ode_rhs(o, tm, tP, k4m, k4P);
    for (int i = 0; i < d; i++) mm[i] = mm[i] + dt * (k1m[i] + 2 * k2m[i] + 2 * k3m[i] + k4m[i]) / 6;
    for (int i = 0; i < n; i++) P[i] = P[i] + dt * (k1P[i] + 2 * k2P[i] + 2 * k3P[i] + k4P[i]) / 6;
}

/* KPT measurement h and its gradient; md:575-578. */
static double kpt_h(int d, int nh, const double *x, double *H)
{
    double s = x[0] + x[d - 1], gs = softplus(s), dgs = dsoftplus(s), h = 0, dsum = 0;
    for (int i = 0; i < d; i++) H[i] = 0;
    for (int k = 1; k <= nh; k++) {
        double sn = sin(gs * k), cs = cos(gs * k);
        h += x[k] * sn;
        H[k] = sn;
        dsum += x[k] * cs * k * dgs;
    }
    H[0] += dsum; H[d - 1] += dsum;
    return h;
}

/* ------------------------------------------------------------------------------------------------ public */
int port_filter(int method, const cgp_model *cm, const cgp_sigma *sg, const cgp_init *in, double dt,
                const double *ys, int64_t B, int64_t T, double *mfs, double *Pfs, double *nll, uint32_t flags)
{
    int d = DIM(cm->d);
    if (d > MAXD || d != cm->d) return CGP_E_UNSUPPORTED;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; b++) {
        model_t m; model_setup(&m, cm, b, dt);
        double mf[MAXD], Pf[MAXD * MAXD], mp[MAXD], Pp[MAXD * MAXD], J[MAXD * MAXD], Sig[MAXD * MAXD], T1[MAXD * MAXD], Hk[MAXD];
        const double *H = in->H ? in->H + b * in->H_stride : NULL;
        double Xi = in->Xi[b * in->Xi_stride], cum = 0;
        memcpy(mf, in->m0 + b * in->m0_stride, sizeof(double) * d);
        memcpy(Pf, in->P0 + b * in->P0_stride, sizeof(double) * d * d);
        ode_t o = { &m, sg, method == CGP_F_CD_EKF ? 0 : 1, NULL, NULL };
        for (int64_t t = 0; t < T; t++) {
            double y = ys[b * T + t], inc;
            switch (method) {
            case CGP_F_EKF:
                disc_eval(&m, mf, dt, mp, J, Sig);
                matmul(d, J, Pf, T1); matmul_nt(d, T1, J, Pp);
                for (int i = 0; i < d * d; i++) Pp[i] += Sig[i];
                inc = linear_update(d, mp, Pp, H, Xi, y, 0, 0, mf, Pf);
                break;
            case CGP_F_SGP:
                sgp_prediction(&m, sg, dt, mf, Pf, mp, Pp, NULL);
                inc = linear_update(d, mp, Pp, H, Xi, y, 0, 0, mf, Pf);
                break;
            case CGP_F_CD_EKF: case CGP_F_CD_SGP:
                memcpy(mp, mf, sizeof(double) * d); memcpy(Pp, Pf, sizeof(double) * d * d);
                rk4(&o, mp, Pp, dt);
                inc = linear_update(d, mp, Pp, H, Xi, y, 0, 0, mf, Pf);
                break;
            default: { /* CGP_F_EKF_KPT, fs:298-311 */
                disc_eval(&m, mf, dt, mp, J, Sig);
                matmul(d, J, Pf, T1); matmul_nt(d, T1, J, Pp);
                for (int i = 0; i < d * d; i++) Pp[i] += Sig[i];
                double pred = kpt_h(d, m.nh, mp, Hk);
                inc = linear_update(d, mp, Pp, Hk, Xi, y, pred, 1, mf, Pf);
                break; }
            }
            cum += inc;
            if (mfs) memcpy(mfs + (b * T + t) * d, mf, sizeof(double) * d);
            if (Pfs) memcpy(Pfs + (b * T + t) * d * d, Pf, sizeof(double) * d * d);
            if (nll && !(flags & CGP_NLL_FINAL_ONLY)) nll[b * T + t] = cum;
        }
        if (nll && (flags & CGP_NLL_FINAL_ONLY)) nll[b] = cum;
    }
    return CGP_OK;
}

int port_smoother(int method, const cgp_model *cm, const cgp_sigma *sg, double dt,
                  const double *mfs, const double *Pfs, int64_t B, int64_t T, double *mss, double *Pss, uint32_t flags)
{
    int d = DIM(cm->d);
    (void)flags;
    if (d > MAXD || d != cm->d) return CGP_E_UNSUPPORTED;
    if (T <= 0) return CGP_OK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; b++) {
        model_t m; model_setup(&m, cm, b, dt);
        double ms[MAXD], Ps[MAXD * MAXD], mp[MAXD], Pp[MAXD * MAXD], J[MAXD * MAXD], Sig[MAXD * MAXD], DT[MAXD * MAXD], T1[MAXD * MAXD], PinvG[MAXD * MAXD];
        memcpy(ms, mfs + (b * T + T - 1) * d, sizeof(double) * d);
        memcpy(Ps, Pfs + (b * T + T - 1) * d * d, sizeof(double) * d * d);
        memcpy(mss + (b * T + T - 1) * d, ms, sizeof(double) * d);
        memcpy(Pss + (b * T + T - 1) * d * d, Ps, sizeof(double) * d * d);
        for (int64_t t = T - 2; t >= 0; t--) {
            const double *mf = mfs + (b * T + t) * d, *Pf = Pfs + (b * T + t) * d * d;
            if (method == CGP_S_EKS) {
                disc_eval(&m, mf, dt, mp, J, Sig);
                matmul(d, J, Pf, DT); matmul_nt(d, DT, J, Pp);
                for (int i = 0; i < d * d; i++) Pp[i] += Sig[i];
                smoother_common(d, DT, mf, Pf, mp, Pp, ms, Ps);
            } else if (method == CGP_S_SGP) {
                sgp_prediction(&m, sg, dt, mf, Pf, mp, Pp, DT);
                smoother_common(d, DT, mf, Pf, mp, Pp, ms, Ps);
            } else {
                cho_solve(d, Pf, m.gamma, d, PinvG);
                ode_t o = { &m, sg, method == CGP_S_CD_EKS ? 2 : 3, mf, PinvG };
                rk4(&o, ms, Ps, -dt);
                (void)T1;
            }
            memcpy(mss + (b * T + t) * d, ms, sizeof(double) * d);
            memcpy(Pss + (b * T + t) * d * d, Ps, sizeof(double) * d * d);
        }
    }
    return CGP_OK;
}

/* qd:234-274 with func = g, d = 1. */
int port_gaussian_expectation(const double *ms, const double *sd, int64_t n, int64_t in_stride,
                              const double *xi, const double *w, int32_t order, double *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        double acc = 0;
        for (int p = 0; p < order; p++) acc += w[p] * softplus(ms[i * in_stride] + sd[i * in_stride] * xi[p]);
        out[i] = acc;
    }
    return CGP_OK;
}

void port_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int port_fixed_d(void)
{
#ifdef FIXED_D
    return FIXED_D;
#else
    return 0;
#endif
}

int port_num_threads(void)
{
    int n = 1;
#ifdef _OPENMP
#pragma omp parallel
#pragma omp master
    n = omp_get_num_threads();
#endif
    return n;
}
