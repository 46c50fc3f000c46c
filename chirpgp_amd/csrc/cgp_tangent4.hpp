// cgp_tangent4.hpp -- the EKF's negative log-likelihood AND its exact gradient in one launch (cgp_ekf_nll_grad): forward tangents of
// (m, P, nll) carried through the scan.
//
// The reference's drivers minimise obj(theta) = ekf(build_model(g(theta)), ys)[-1][-1] with value_and_grad THROUGH the scan
// (demos/ekfs_mle.py:43-51): an exact derivative.  Rounds 1 - 5 took central differences of 2 P + 1 filter passes (accurate to 1e-4 of
// the gradient).  Here one lane carries ONE tangent direction of one trial: the primal recursion of
// filters_smoothers.py:55-68, 222-264 on the chirp LCD model (models.py:264-311; SURVEY N1 for its Jacobian) and, beside it, the
// derivative of every quantity along the direction:
//     d mp = J dm + (d f)(m)                       d J = (dJ/dm)[dm] + (d J)            (d .) = the model's explicit dependence on the
//     d Pp = X + X^T + J dP J^T + d Sigma,  X = d J (J P)^T                              parameters along the direction
//     d S = h . d Pp h + d Xi      d K = (d Pp h - K d S) / S      d nu = -h . d mp
//     d mf = d mp + d K nu + K d nu        d Pf = d Pp - (d K K^T + K d K^T) S - K K^T d S
//     d nll += (d S / S + 2 nu d nu / S - nu^2 d S / S^2) / 2
// A direction is 24 doubles the HOST computes from the builder (chirpgp_amd/mle.py: complex-step derivatives of the model constants
// along theta_k): d log rho, d q, d M32_F (4), d M32_Sigma (3), d Xi, d m0 (4), d P0 (10, packed lower triangle).  The frequency
// state's chain (softplus, its first and second derivative, sin / cos) is differentiated in the kernel.  Nothing is approximated:
// the gradient agrees with the derivative taken in 100-digit arithmetic to rounding (tests/test_gpu_gradient.py: 1e-8).
//
// Layout: lane g = trial * n_dir + direction -- a trial's directions sit in neighbouring lanes and read the same measurements (one
// 64-byte request per eight steps, served once per trial by the cache).  Every lane repeats the primal (200 of its ~520 instructions a step):
// cheaper than exchanging it.  value = nll[trial] (written by direction 0), grad[trial][direction].
#pragma once
#include "cgp_kernels.hpp"

namespace cgp {

constexpr int kDirDoubles = 24;      // d log rho | d q | d M (4) | d MS (3) | d Xi | d m0 (4) | d P0 (10)

struct TangentIO {
    const double* __restrict__ H;  int64_t H_stride;
    const double* __restrict__ Xi; int64_t Xi_stride;
    const double* __restrict__ m0; int64_t m0_stride;
    const double* __restrict__ P0; int64_t P0_stride;
    const double* __restrict__ ys; int64_t ys_stride, ys_repeat; const int32_t* __restrict__ ys_index;
    const double* __restrict__ dirs;     // [B][n_dir][kDirDoubles]
    int64_t B, T;
    int n_dir;
    double* __restrict__ nll;            // [B]
    double* __restrict__ grad;           // [B][n_dir]
    __device__ __forceinline__ const double* record(int64_t trial) const {
        int64_t g = trial;
        if (ys_repeat > 1) g = (int64_t)((uint64_t)trial / (uint64_t)ys_repeat);
        if (ys_index) g = ys_index[g];
        return ys + g * ys_stride;
    }
};

__global__ void __launch_bounds__(64) ekf4_tangent_kernel(TangentIO io, ModelArgs ma) {
    const int64_t gidx = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t total = io.B * io.n_dir;
    const bool active = gidx < total;
    if (!active) return;                                                  // (a partial wavefront runs with a partial EXEC mask)
    const int64_t gi = gidx;
    const int64_t trial = gi / io.n_dir;
    const int dir = (int)(gi - trial * io.n_dir);

    HarmonicLCD<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    const double rho = model.rho, q = model.q;
    const double M0 = model.M[0], M1 = model.M[1], M2 = model.M[2], M3 = model.M[3];
    const double scale = (kTwoPi * model.fs) * model.dt;                  // rotation angle = scale * softplus(u2)
    const double* __restrict__ dp = io.dirs + gi * kDirDoubles;
    const double dlr = dp[0], dq = dp[1], dM0 = dp[2], dM1 = dp[3], dM2 = dp[4], dM3 = dp[5];
    const double dS0 = dp[6], dS1 = dp[7], dS2 = dp[8], dXi = dp[9];
    double h[4];
    CGP_UNROLL for (int i = 0; i < 4; i++) h[i] = io.H[trial * io.H_stride + i];
    const double Xi = io.Xi[trial * io.Xi_stride];

    // state: m, P (packed lower triangle: 00 10 11 20 21 22 30 31 32 33) and their tangents
    double m[4], dm[4], P[10], dP[10];
    CGP_UNROLL for (int i = 0; i < 4; i++) { m[i] = io.m0[trial * io.m0_stride + i]; dm[i] = dp[10 + i]; }
    {
        const double* __restrict__ p0 = io.P0 + trial * io.P0_stride;
        int k = 0;
        CGP_UNROLL for (int i = 0; i < 4; i++) CGP_UNROLL for (int j = 0; j <= i; j++) { P[k] = p0[i * 4 + j]; dP[k] = dp[14 + k]; k++; }
    }
    double nll = 0.0, dnll = 0.0;
    const double* __restrict__ rec = io.record(trial);
    auto S_ = [](const double (&A)[10], int i, int j) { return i >= j ? A[i * (i + 1) / 2 + j] : A[j * (j + 1) / 2 + i]; };

    for (int64_t t0 = 0; t0 < io.T; t0 += 8) {
        double yb[8];
        CGP_UNROLL for (int k = 0; k < 8; k++) yb[k] = (t0 + k < io.T) ? rec[t0 + k] : 0.0;
        const int n = (io.T - t0 < 8) ? (int)(io.T - t0) : 8;
        for (int k = 0; k < n; k++) {
            const double y = yb[k];
            // ---- the model at m: softplus and its two derivatives, the rotation
            double sp, dsp;
            softplus_pair(m[2], sp, dsp);
            const double th1 = scale * dsp;                               // d theta / d u2
            const double th2 = th1 * (1.0 - dsp);                         // d^2 theta / d u2^2   (sigmoid' = sigmoid (1 - sigmoid))
            double sn, cs;
            fast_sincos(scale * sp, sn, cs);
            const double rc = rho * cs, rs = rho * sn;
            const double mp0 = rc * m[0] - rs * m[1], mp1 = rs * m[0] + rc * m[1];
            const double mp2 = M0 * m[2] + M1 * m[3], mp3 = M2 * m[2] + M3 * m[3];
            const double J02 = -th1 * mp1, J12 = th1 * mp0;
            // ---- tangent of the model along the direction
            const double dth = th1 * dm[2], dth1 = th2 * dm[2];
            const double drc = dlr * rc - rs * dth, drs = dlr * rs + rc * dth;
            const double dmp0 = drc * m[0] - drs * m[1] + rc * dm[0] - rs * dm[1];
            const double dmp1 = drs * m[0] + drc * m[1] + rs * dm[0] + rc * dm[1];
            const double dmp2 = dM0 * m[2] + dM1 * m[3] + M0 * dm[2] + M1 * dm[3];
            const double dmp3 = dM2 * m[2] + dM3 * m[3] + M2 * dm[2] + M3 * dm[3];
            const double dJ02 = -(dth1 * mp1 + th1 * dmp1), dJ12 = dth1 * mp0 + th1 * dmp0;
            // ---- A = J P (rows of J: [rc, -rs, J02, 0], [rs, rc, J12, 0], [0, 0, M0, M1], [0, 0, M2, M3])
            double A[4][4], dA[4][4];      // dA = dJ P + J dP
            CGP_UNROLL for (int j = 0; j < 4; j++) {
                const double p0 = S_(P, 0, j), p1 = S_(P, 1, j), p2 = S_(P, 2, j), p3 = S_(P, 3, j);
                const double e0 = S_(dP, 0, j), e1 = S_(dP, 1, j), e2 = S_(dP, 2, j), e3 = S_(dP, 3, j);
                A[0][j] = rc * p0 - rs * p1 + J02 * p2;
                A[1][j] = rs * p0 + rc * p1 + J12 * p2;
                A[2][j] = M0 * p2 + M1 * p3;
                A[3][j] = M2 * p2 + M3 * p3;
                dA[0][j] = drc * p0 - drs * p1 + dJ02 * p2 + rc * e0 - rs * e1 + J02 * e2;
                dA[1][j] = drs * p0 + drc * p1 + dJ12 * p2 + rs * e0 + rc * e1 + J12 * e2;
                dA[2][j] = dM0 * p2 + dM1 * p3 + M0 * e2 + M1 * e3;
                dA[3][j] = dM2 * p2 + dM3 * p3 + M2 * e2 + M3 * e3;
            }
            // ---- Pp = A J^T + Sigma;  dPp = dA J^T + A dJ^T + dSigma   (lower triangle; column i of J^T = row i of J)
            double Pp[10], dPp[10];
            {
                int kk = 0;
                CGP_UNROLL for (int i = 0; i < 4; i++) CGP_UNROLL for (int j = 0; j <= i; j++) {
                    double v, dv;
                    if (j == 0) { v = A[i][0] * rc - A[i][1] * rs + A[i][2] * J02; dv = dA[i][0] * rc - dA[i][1] * rs + dA[i][2] * J02 + A[i][0] * drc - A[i][1] * drs + A[i][2] * dJ02; }
                    else if (j == 1) { v = A[i][0] * rs + A[i][1] * rc + A[i][2] * J12; dv = dA[i][0] * rs + dA[i][1] * rc + dA[i][2] * J12 + A[i][0] * drs + A[i][1] * drc + A[i][2] * dJ12; }
                    else if (j == 2) { v = A[i][2] * M0 + A[i][3] * M1; dv = dA[i][2] * M0 + dA[i][3] * M1 + A[i][2] * dM0 + A[i][3] * dM1; }
                    else { v = A[i][2] * M2 + A[i][3] * M3; dv = dA[i][2] * M2 + dA[i][3] * M3 + A[i][2] * dM2 + A[i][3] * dM3; }
                    Pp[kk] = v; dPp[kk] = dv; kk++;
                }
                Pp[0] += q; Pp[2] += q; Pp[5] += model.MS[0]; Pp[8] += model.MS[1]; Pp[9] += model.MS[2];          // Sigma = blockdiag(q, q, M32_Sigma)
                dPp[0] += dq; dPp[2] += dq; dPp[5] += dS0; dPp[8] += dS1; dPp[9] += dS2;
            }
            // ---- update (filters_smoothers.py:55-68) and its tangent
            double PH[4], dPH[4];
            CGP_UNROLL for (int i = 0; i < 4; i++) {
                PH[i] = S_(Pp, i, 0) * h[0] + S_(Pp, i, 1) * h[1] + S_(Pp, i, 2) * h[2] + S_(Pp, i, 3) * h[3];
                dPH[i] = S_(dPp, i, 0) * h[0] + S_(dPp, i, 1) * h[1] + S_(dPp, i, 2) * h[2] + S_(dPp, i, 3) * h[3];
            }
            const double S = h[0] * PH[0] + h[1] * PH[1] + h[2] * PH[2] + h[3] * PH[3] + Xi;
            const double dS = h[0] * dPH[0] + h[1] * dPH[1] + h[2] * dPH[2] + h[3] * dPH[3] + dXi;
            const double iS = rcp_nr(S);
            const double pred = h[0] * mp0 + h[1] * mp1 + h[2] * mp2 + h[3] * mp3;
            const double nu = y - pred;
            const double dnu = -(h[0] * dmp0 + h[1] * dmp1 + h[2] * dmp2 + h[3] * dmp3);
            double K[4], dK[4];
            CGP_UNROLL for (int i = 0; i < 4; i++) { K[i] = PH[i] * iS; dK[i] = (dPH[i] - K[i] * dS) * iS; }
            const double mpv[4] = {mp0, mp1, mp2, mp3}, dmpv[4] = {dmp0, dmp1, dmp2, dmp3};
            CGP_UNROLL for (int i = 0; i < 4; i++) { m[i] = mpv[i] + K[i] * nu; dm[i] = dmpv[i] + dK[i] * nu + K[i] * dnu; }
            {
                int kk = 0;
                CGP_UNROLL for (int i = 0; i < 4; i++) CGP_UNROLL for (int j = 0; j <= i; j++) {
                    const double kk_ij = K[i] * K[j];
                    P[kk] = Pp[kk] - kk_ij * S;
                    dP[kk] = dPp[kk] - (dK[i] * K[j] + K[i] * dK[j]) * S - kk_ij * dS;
                    kk++;
                }
            }
            nll += nll_increment(S, nu);
            dnll += 0.5 * (dS * iS + (2.0 * nu * dnu - nu * nu * dS * iS) * iS);
        }
    }
    if (active) {
        if (dir == 0) io.nll[trial] = nll;
        io.grad[gi] = dnll;
    }
}

inline hipError_t launch_ekf4_tangent(const TangentIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0 || io.n_dir <= 0) return hipSuccess;
    const int64_t total = io.B * io.n_dir;
    hipLaunchKernelGGL(ekf4_tangent_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, stream, io, ma);
    return hipGetLastError();
}

}  // namespace cgp
