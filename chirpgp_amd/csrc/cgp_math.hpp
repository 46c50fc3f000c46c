// cgp_math.hpp -- fixed-size float64 linear algebra held entirely in VGPRs (gfx950).
//
// Every container is a plain array indexed only by compile-time constants after full unrolling, so the
// compiler keeps it in registers (runtime-indexed arrays go to scratch: cdna_hip_programming.md rule 20).
// Symmetric matrices are stored packed (lower triangle, D(D+1)/2 doubles): d = 4 -> 10, d = 8 -> 36 doubles.
#pragma once
#ifndef __HIPCC_RTC__          // (a runtime compilation -- cgp_rtc.hip, custom models -- has the device runtime built in and no host headers)
#include <hip/hip_runtime.h>
#include <cstdint>
#else                          // the fixed-width types of <cstdint> as this platform's host compiler defines them (LP64)
typedef long int64_t; typedef unsigned long uint64_t; typedef int int32_t; typedef unsigned int uint32_t;
typedef unsigned long uintptr_t; typedef unsigned long size_t;
#endif

#define CGP_DEV __device__ __forceinline__
#define CGP_UNROLL _Pragma("unroll")

namespace cgp {

constexpr double kTwoPi = 6.283185307179586476925286766559;

template <int D> struct Vec { double v[D]; };
template <int D> struct Mat { double a[D][D]; };

template <int D> struct Sym {
    static constexpr int N = D * (D + 1) / 2;
    double a[N];
    static constexpr int idx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
    CGP_DEV double  operator()(int i, int j) const { return a[idx(i, j)]; }
    CGP_DEV double& operator()(int i, int j) { return a[idx(i, j)]; }
};

// ---------------------------------------------------------------------------------------------- products
// T = A P   (A dense, P symmetric)
template <int D> CGP_DEV void mul_dense_sym(const Mat<D>& A, const Sym<D>& P, Mat<D>& T) {
    CGP_UNROLL for (int i = 0; i < D; i++)
        CGP_UNROLL for (int j = 0; j < D; j++) {
            double s = A.a[i][0] * P(0, j);
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(A.a[i][k], P(k, j), s);
            T.a[i][j] = s;
        }
}
// out = T A^T + C  (lower triangle only; result symmetric by construction of the callers)
template <int D> CGP_DEV void mul_nt_sym_add(const Mat<D>& T, const Mat<D>& A, const Sym<D>& C, Sym<D>& out) {
    CGP_UNROLL for (int i = 0; i < D; i++)
        CGP_UNROLL for (int j = 0; j <= i; j++) {
            double s = T.a[i][0] * A.a[j][0];
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(T.a[i][k], A.a[j][k], s);
            out(i, j) = s + C(i, j);
        }
}
// out = T + T^T + sign * C   (symmetric)
template <int D> CGP_DEV void sym_from_sum(const Mat<D>& T, const Sym<D>& C, double sign, Sym<D>& out) {
    CGP_UNROLL for (int i = 0; i < D; i++)
        CGP_UNROLL for (int j = 0; j <= i; j++) out(i, j) = (T.a[i][j] + T.a[j][i]) + sign * C(i, j);
}
template <int D> CGP_DEV void matvec(const Mat<D>& A, const Vec<D>& x, Vec<D>& y) {
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = A.a[i][0] * x.v[0];
        CGP_UNROLL for (int k = 1; k < D; k++) s = fma(A.a[i][k], x.v[k], s);
        y.v[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------- Cholesky
// Lower Cholesky factor of a packed symmetric matrix.  Like LAPACK potrf under JAX, a pivot that is <= 0
// or NaN makes the whole factor NaN (no trap, no exception): SURVEY.md section 5 "silent NaN propagation".
// Also returns 1/L_ii in inv_diag for the triangular solves.
// sqrt(s) and 1 / sqrt(s) together from v_rsq_f64 (~2^-24) and ONE coupled Newton (Goldschmidt) step: 6 instructions,
// 5e-15 relative (1.5 e^2), instead of sqrt() + divide (107 + 74 cycles of dependent latency, measured).  Round 1 took a
// second step and a residual correction (11 instructions, ~1 ulp): the Cholesky factors feed filters gated at 1e-5 whose
// results sat at 1e-13, and every wave-per-trial sigma-point kernel factorises once per step or RK4 stage.
CGP_DEV void sqrt_rsqrt(double s, double& root, double& inv_root) {
    const double y = __builtin_amdgcn_rsq(s);
    const double g = s * y, h = 0.5 * y;
    const double r = fma(-g, h, 0.5);
    root = fma(g, r, g);
    const double h1 = fma(h, r, h);
    inv_root = h1 + h1;
}

template <int D> CGP_DEV void cholesky(const Sym<D>& P, Sym<D>& L, Vec<D>& inv_diag) {
    bool bad = false;
    CGP_UNROLL for (int j = 0; j < D; j++) {
        double s = P(j, j);
        CGP_UNROLL for (int k = 0; k < j; k++) s = fma(-L(j, k), L(j, k), s);
        bad = bad || !(s > 0.0);
        double ljj, inv;
        sqrt_rsqrt(s, ljj, inv);
        L(j, j) = ljj;
        inv_diag.v[j] = inv;
        CGP_UNROLL for (int i = j + 1; i < D; i++) {
            double t = P(i, j);
            CGP_UNROLL for (int k = 0; k < j; k++) t = fma(-L(i, k), L(j, k), t);
            L(i, j) = t * inv;
        }
    }
    const double poison = bad ? __builtin_nan("") : 0.0;
    CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) L.a[i] += poison;
    CGP_UNROLL for (int i = 0; i < D; i++) inv_diag.v[i] += poison;
}

// Solve (L L^T) x = b for one right-hand side given as D scalars (in place).
template <int D> CGP_DEV void cho_solve_vec(const Sym<D>& L, const Vec<D>& inv_diag, Vec<D>& x) {
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = x.v[i];
        CGP_UNROLL for (int k = 0; k < i; k++) s = fma(-L(i, k), x.v[k], s);
        x.v[i] = s * inv_diag.v[i];
    }
    CGP_UNROLL for (int i = D - 1; i >= 0; i--) {
        double s = x.v[i];
        CGP_UNROLL for (int k = i + 1; k < D; k++) s = fma(-L(k, i), x.v[k], s);
        x.v[i] = s * inv_diag.v[i];
    }
}
// X = (L L^T)^{-1} R, column by column (R and X dense D x D, may alias).
template <int D> CGP_DEV void cho_solve_mat(const Sym<D>& L, const Vec<D>& inv_diag, const Mat<D>& R, Mat<D>& X) {
    CGP_UNROLL for (int c = 0; c < D; c++) {
        Vec<D> col;
        CGP_UNROLL for (int i = 0; i < D; i++) col.v[i] = R.a[i][c];
        cho_solve_vec<D>(L, inv_diag, col);
        CGP_UNROLL for (int i = 0; i < D; i++) X.a[i][c] = col.v[i];
    }
}

// ---------------------------------------------------------------------------------------------- Kalman pieces
// Scalar-measurement update, filters_smoothers.py:55-68: S = H^T Pp H + Xi, K = Pp H / S, mf = mp + K (y - pred),
// Pf = Pp - (K K^T) S (the reference's form, not Joseph).  Here K = (Pp H) * (1 / S) with a Newton-refined reciprocal
// and (K K^T) S is formed as K (Pp H)^T -- the same matrix, one rounding fewer.  Returns S and the innovation; the
// negative log-likelihood increment is nll_increment(S, innov) (cgp_fastmath.hpp).
template <int D>
CGP_DEV void scalar_update(const Vec<D>& mp, const Sym<D>& Pp, const Vec<D>& H, double Xi, double y,
                           bool override_pred, double pred_in, Vec<D>& mf, Sym<D>& Pf, double& S_out, double& innov_out) {
    Vec<D> PH;
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = Pp(i, 0) * H.v[0];
        CGP_UNROLL for (int k = 1; k < D; k++) s = fma(Pp(i, k), H.v[k], s);
        PH.v[i] = s;
    }
    double S = H.v[0] * PH.v[0], pred = H.v[0] * mp.v[0];
    CGP_UNROLL for (int k = 1; k < D; k++) { S = fma(H.v[k], PH.v[k], S); pred = fma(H.v[k], mp.v[k], pred); }
    S += Xi;
    if (override_pred) pred = pred_in;
    const double innov = y - pred;
    double rS = __builtin_amdgcn_rcp(S);
    double e = fma(-S, rS, 1.0);
    rS = fma(rS, e, rS);
    e = fma(-S, rS, 1.0);
    rS = fma(rS, e, rS);
    Vec<D> K;
    CGP_UNROLL for (int i = 0; i < D; i++) K.v[i] = PH.v[i] * rS;
    CGP_UNROLL for (int i = 0; i < D; i++) mf.v[i] = fma(K.v[i], innov, mp.v[i]);
    CGP_UNROLL for (int i = 0; i < D; i++)
        CGP_UNROLL for (int j = 0; j <= i; j++) Pf(i, j) = fma(-K.v[i], PH.v[j], Pp(i, j));
    S_out = S;
    innov_out = innov;
}

// Gaussian smoother step, filters_smoothers.py:71-85: G = (Pp^{-1} DT)^T, ms = mf + G (ms - mp), Ps = Pf + G (Ps - Pp) G^T.
// DT is dense D x D (the transposed cross-covariance).  ms / Ps are updated in place.
template <int D>
CGP_DEV void smoother_gain(const Mat<D>& DT, const Sym<D>& Pp, Mat<D>& G) {
    Sym<D> L; Vec<D> inv;
    cholesky<D>(Pp, L, inv);
    Mat<D> X;
    cho_solve_mat<D>(L, inv, DT, X);
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) G.a[i][j] = X.a[j][i];
}
template <int D>
CGP_DEV void smoother_apply(const Mat<D>& G, const Vec<D>& mf, const Sym<D>& Pf, const Vec<D>& mp, const Sym<D>& Pp,
                            Vec<D>& ms, Sym<D>& Ps) {
    Vec<D> dm; Sym<D> dP;
    CGP_UNROLL for (int i = 0; i < D; i++) dm.v[i] = ms.v[i] - mp.v[i];
    CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) dP.a[i] = Ps.a[i] - Pp.a[i];
    Mat<D> T;
    mul_dense_sym<D>(G, dP, T);
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = mf.v[i];
        CGP_UNROLL for (int k = 0; k < D; k++) s = fma(G.a[i][k], dm.v[k], s);
        ms.v[i] = s;
    }
    mul_nt_sym_add<D>(T, G, Pf, Ps);
}

// ---------------------------------------------------------------------------------------------- memory helpers
template <int D> CGP_DEV void load_vec(const double* __restrict__ p, Vec<D>& v) {
    CGP_UNROLL for (int i = 0; i < D; i++) v.v[i] = p[i];
}
// Reads the lower triangle of a row-major D x D matrix (what LAPACK potrf / the reference's symmetric maths use).
template <int D> CGP_DEV void load_sym(const double* __restrict__ p, Sym<D>& P) {
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j <= i; j++) P(i, j) = p[i * D + j];
}
template <int D> CGP_DEV void load_mat(const double* __restrict__ p, Mat<D>& A) {
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) A.a[i][j] = p[i * D + j];
}
template <int D> CGP_DEV void store_vec(double* __restrict__ p, const Vec<D>& v) {
    if constexpr (D % 2 == 0) {
        CGP_UNROLL for (int i = 0; i < D; i += 2) *reinterpret_cast<double2*>(p + i) = make_double2(v.v[i], v.v[i + 1]);
    } else {
        CGP_UNROLL for (int i = 0; i < D; i++) p[i] = v.v[i];
    }
}
template <int D> CGP_DEV void store_sym_full(double* __restrict__ p, const Sym<D>& P) {
    if constexpr (D % 2 == 0) {
        CGP_UNROLL for (int i = 0; i < D; i++)
            CGP_UNROLL for (int j = 0; j < D; j += 2) *reinterpret_cast<double2*>(p + i * D + j) = make_double2(P(i, j), P(i, j + 1));
    } else {
        CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) p[i * D + j] = P(i, j);
    }
}

// Inclusive prefix sum over the 64 lanes with DPP moves (zeros shifted in): within the rows of 16 by row_shr 1 / 2 / 3 of the
// input, then 4 and 8 of the partial sums on the lanes that have such a neighbour (bank masks), then the row totals forwarded
// with row_bcast 15 / 31 (row masks) -- seven move-pair + add steps of ~ 13 cycles, where six __shfl_up rounds (two
// ds_bpermute_b32 and an LDS round trip each) cost some 700 cycles per flush: ~ 10 cycles of every step of a 64-step chunk.
template <int CTRL, int ROW_MASK, int BANK_MASK> CGP_DEV double dpp_zero_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}
CGP_DEV double wave_inclusive_scan(double v) {
    double s = v + dpp_zero_f64<0x111, 0xF, 0xF>(v);            // row_shr:1
    s += dpp_zero_f64<0x112, 0xF, 0xF>(v);                      // row_shr:2
    s += dpp_zero_f64<0x113, 0xF, 0xF>(v);                      // row_shr:3
    s += dpp_zero_f64<0x114, 0xF, 0xE>(s);                      // row_shr:4 on lanes 4..15 of a row
    s += dpp_zero_f64<0x118, 0xF, 0xC>(s);                      // row_shr:8 on lanes 8..15
    s += dpp_zero_f64<0x142, 0xA, 0xF>(s);                      // row_bcast:15: lane 15 of rows 0, 2 into rows 1, 3
    s += dpp_zero_f64<0x143, 0xC, 0xF>(s);                      // row_bcast:31: lane 31 into rows 2, 3
    return s;
}
// Wave-uniform broadcast of a double held in lane `src` (src may be a runtime, wave-uniform value).
CGP_DEV double readlane_f64(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

}  // namespace cgp
