"""Generates tests/golden/exact_*.npz: the path's formulas evaluated in 100-digit arithmetic (mpmath), rounded to float64 at the end.

Why.  The reference cannot run in this pipeline (no JAX: SURVEY.md 8c), so the float64 oracles (oracle/np_filters.py, oracle/c/port.c)
are pinned by the reference's property tests only.  What separates ANY float64 evaluation of the reference's formulas -- XLA's, NumPy's,
the C port's, the HIP kernels' -- from each other is rounding: operation order, fused multiply-adds, the libm behind exp / log / sin /
cos.  This script evaluates the same recursions with a unit roundoff of 1e-100, i.e. their exact values for float64 inputs, so that the
distance of every float64 implementation from that common reference can be measured in units of 2^-53 (tests/test_exact.py): the
"amplification" of the recursion.  XLA's result is an evaluation of the same kind and sits within the same distance of the exact value --
that bounds what the unrunnable reference could differ by, against the north star's 1e-5 gate.  (Parity stays "partial" by the rules.)

Restated here, independently of oracle/ (plain lists of mpf, no NumPy linear algebra), following the reference line by line:
    ekf / eks                         filters_smoothers.py:55-85, 222-264, 317-349     (jacfwd -> complex step with h = 1e-45 at 100 digits)
    sgp_filter / sgp_smoother         filters_smoothers.py:88-121, 446-531             (Gauss-Hermite order 3: nodes 0, +-sqrt 3; weights 2/3, 1/6, 1/6)
    cd_sgp_filter / cd_sgp_smoother   filters_smoothers.py:124-137, 534-632; quadratures.py:34-81
    chirp model                       models.py:50, 61-73, 76-119, 264-311, 437-459
Inputs: the toy chirp of demos/ekfs_mle.py:16-39 at the MLE start point, T = 500 (cd: 300), dt = 1e-3 -- one record at Xi = 0.1 (tracks)
and one at Xi = 1 on a 5.5 Hz chirp started at frequency state 7 (the estimate slides from 7 to 4.4 through the regime boundaries of the
headline kernel's speculative step; "exact_lost": the filter starts 1.5 Hz off and is ten times noisier).

    python -m tests.golden.make_exact          (about two minutes)
    python -m tests.golden.make_exact --grad   (exact_grad.npz: the MLE objective's exact gradient on the same records; about five minutes)
    python -m tests.golden.make_exact --rest   (exact_rest.npz: kf, rts, cd_ekf, cd_eks, ekf_for_kpt on the first record -- with the three pairs above,
                                                all eleven public functions of filters_smoothers.py; about a minute)
    python -m tests.golden.make_exact --lascala   (exact_lascala.npz: the La Scala model through the six pipelines of the first fixtures)
    python -m tests.golden.make_exact --harmonic  (exact_harmonic.npz: BASELINE C5's model -- three harmonics, d = 8 -- with ekf + eks and the cubature
                                                sgp_filter + sgp_smoother)
"""
import math
import os
import sys

import numpy as np
from mpmath import mp, mpf, mpc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))
mp.dps = 100
H_STEP = mpf(10) ** -45
NAN = mpf('nan')


# ------------------------------------------------------------------------------------------------ small dense algebra on lists
def zeros(n, m):
    return [[mpf(0)] * m for _ in range(n)]


def matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(len(B))) for j in range(len(B[0]))] for i in range(len(A))]


def matvec(A, x):
    return [sum(A[i][k] * x[k] for k in range(len(x))) for i in range(len(A))]


def tr(A):
    return [list(r) for r in zip(*A)]


def madd(A, B, s=1):
    return [[a + s * b for a, b in zip(ra, rb)] for ra, rb in zip(A, B)]


def mscale(A, s):
    return [[a * s for a in r] for r in A]


def vadd(a, b, s=1):
    return [x + s * y for x, y in zip(a, b)]


def outer(a, b):
    return [[x * y for y in b] for x in a]


def chol(P):
    """lower Cholesky factor; all-NaN where the matrix is not positive definite (JAX semantics)"""
    n = len(P)
    L = zeros(n, n)
    for j in range(n):
        s = P[j][j] - sum(L[j][k] ** 2 for k in range(j))
        if not (s > 0):
            return [[NAN] * n for _ in range(n)]
        L[j][j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i][j] = (P[i][j] - sum(L[i][k] * L[j][k] for k in range(j))) / L[j][j]
    return L


def cho_solve(P, R):
    """P^-1 R through the Cholesky factor (R a matrix)"""
    n = len(P)
    L = chol(P)
    cols = []
    for c in range(len(R[0])):
        y = [mpf(0)] * n
        for i in range(n):
            y[i] = (R[i][c] - sum(L[i][k] * y[k] for k in range(i))) / L[i][i]
        x = [mpf(0)] * n
        for i in reversed(range(n)):
            x[i] = (y[i] - sum(L[k][i] * x[k] for k in range(i + 1, n))) / L[i][i]
        cols.append(x)
    return tr(cols)


def blkdiag(*blocks):
    blocks = [b if isinstance(b, list) else [[b]] for b in blocks]
    n = sum(len(b) for b in blocks)
    out = zeros(n, n)
    k = 0
    for b in blocks:
        for i, r in enumerate(b):
            for j, v in enumerate(r):
                out[k + i][k + j] = v
        k += len(b)
    return out


# ------------------------------------------------------------------------------------------------ the chirp model (models.py)
def g(x):
    return mp.log(mp.exp(x) + 1)                                        # models.py:50


def m32_solution(ell, sigma, dt):                                        # models.py:61-73
    gam = mp.sqrt(3) / ell
    eta = dt * gam
    beta = sigma ** 2 * mp.exp(-2 * eta)
    e = mp.exp(-eta)
    F = [[(1 + eta) * e, dt * e], [-dt * gam ** 2 * e, (1 - eta) * e]]
    off = 2 * dt ** 2 * gam ** 3 * beta
    S = [[sigma ** 2 - beta * (2 * eta + 2 * eta ** 2 + 1), off],
         [off, gam ** 2 * (sigma ** 2 + beta * (2 * eta - 2 * eta ** 2 - 1))]]
    return F, S


def build_chirp_model(params):
    """models.py:437-459 with 76-119 and 264-311: (drift, b, cond_m_cov, m0, P0, H); params = lam, b, delta, ell, sigma, m0_v (float64 -> exact)"""
    lam, b, delta, ell, sigma, m0_v = (p if isinstance(p, mpf) else mpf(float(p)) for p in params)
    gam = mp.sqrt(3) / ell

    def drift(u):
        w = 2 * mp.pi * g(u[2])
        return [-lam * u[0] - w * u[1], w * u[0] - lam * u[1], u[3], -(gam ** 2) * u[2] - 2 * gam * u[3]]
    disp = blkdiag(b, b, mpf(0), 2 * sigma * (mp.sqrt(3) / ell) ** mpf('1.5'))

    def cond_m_cov(u, dt):
        w = 2 * mp.pi * g(u[2])
        c, s, e = mp.cos(dt * w), mp.sin(dt * w), mp.exp(-lam * dt)
        Fm, Sm = m32_solution(ell, sigma, dt)
        mean = [e * (c * u[0] - s * u[1]), e * (s * u[0] + c * u[1]), Fm[0][0] * u[2] + Fm[0][1] * u[3], Fm[1][0] * u[2] + Fm[1][1] * u[3]]
        q = b ** 2 * dt if lam == 0 else b ** 2 / (2 * lam) * (1 - mp.exp(-2 * lam * dt))
        return mean, blkdiag(q, q, Sm)
    m0 = [mpf(0), mpf(0), m0_v, mpf(0)]
    P0 = blkdiag(delta, delta, [[sigma ** 2, mpf(0)], [mpf(0), (mp.sqrt(3) / ell) ** 2 * sigma ** 2]])
    H = [mpf(0), mpf(1), mpf(0), mpf(0)]
    return drift, disp, cond_m_cov, m0, P0, H


def jacobian(f, x):
    """jacfwd: the exact derivative -- complex step, error h^2 = 1e-90"""
    cols = []
    for j in range(len(x)):
        xc = [mpc(v) for v in x]
        xc[j] = mpc(x[j], H_STEP)
        cols.append([mp.im(v) / H_STEP for v in f(xc)])
    return tr(cols)


# ------------------------------------------------------------------------------------------------ filters_smoothers.py
def linear_update(mp_, Pp, H, Xi, y):                                   # :55-68
    PH = matvec(Pp, H)
    S = sum(h * v for h, v in zip(H, PH)) + Xi
    K = [v / S for v in PH]
    pred = sum(h * v for h, v in zip(H, mp_))
    scale2 = mp.sqrt(S) ** 2
    nll = (mp.log(2 * mp.pi * scale2) + (y - pred) ** 2 / scale2) / 2      # :44-45
    return vadd(mp_, [k * (y - pred) for k in K]), madd(Pp, mscale(outer(K, K), S), -1), nll


def smoother_common(DT, mf, Pf, mp_, Pp, ms, Ps):                       # :71-85
    G = tr(cho_solve(Pp, DT))
    return vadd(mf, matvec(G, vadd(ms, mp_, -1))), madd(Pf, matmul(matmul(G, madd(Ps, Pp, -1)), tr(G)))


class GH3:
    """SigmaPoints.gauss_hermite(4, 3), quadratures.py:156-196: dimension 0 varies fastest"""
    def __init__(self, d=4):
        nodes = [mpf(0), mp.sqrt(3), -mp.sqrt(3)]                        # sqrt(2) x the roots of H_3, flipped order of np.roots: (0, +, -) ... any order: a sum
        w1 = [mpf(2) / 3, mpf(1) / 6, mpf(1) / 6]
        self.xi, self.w = [], []
        for n in range(3 ** d):
            idx = [(n // 3 ** r) % 3 for r in range(d)]
            self.xi.append([nodes[i] for i in idx])
            wt = mpf(1)
            for i in idx:
                wt *= w1[i]
            self.w.append(wt)

    def points(self, m, L):
        return [vadd(m, matvec(L, x)) for x in self.xi]

    def expect_vec(self, vals):
        return [sum(w * v[i] for w, v in zip(self.w, vals)) for i in range(len(vals[0]))]

    def expect_outer(self, a, b):
        n, m = len(a[0]), len(b[0])
        return [[sum(w * x[i] * y[j] for w, x, y in zip(self.w, a, b)) for j in range(m)] for i in range(n)]


def sgp_prediction(sg, cond_m_cov, dt, mf, Pf):                         # :88-121
    chi = sg.points(mf, chol(Pf))
    ev = [cond_m_cov(c, dt) for c in chi]
    fm = [e[0] for e in ev]
    mp_ = sg.expect_vec(fm)
    Ecov = zeros(len(mf), len(mf))
    for w, e in zip(sg.w, ev):
        Ecov = madd(Ecov, mscale(e[1], w))
    Pp = madd(madd(sg.expect_outer(fm, fm), Ecov), outer(mp_, mp_), -1)
    return mp_, Pp, chi, fm


def cd_sgp_common(sg, drift, b, m, P):                                  # :124-137
    chi = sg.points(m, chol(P))
    fa = [drift(c) for c in chi]
    mp_ = sg.expect_vec(fa)
    _Pp = sg.expect_outer([vadd(c, m, -1) for c in chi], fa)
    return mp_, madd(madd(_Pp, tr(_Pp)), matmul(b, tr(b)))


def rk4(ode, m, P, dt, *fixed):                                         # quadratures.py:34-81
    k1m, k1P = ode(m, P, *fixed)
    k2m, k2P = ode(vadd(m, k1m, dt / 2), madd(P, k1P, dt / 2), *fixed)
    k3m, k3P = ode(vadd(m, k2m, dt / 2), madd(P, k2P, dt / 2), *fixed)
    k4m, k4P = ode(vadd(m, k3m, dt), madd(P, k3P, dt), *fixed)
    km = [a + 2 * b_ + 2 * c + d for a, b_, c, d in zip(k1m, k2m, k3m, k4m)]
    kP = madd(madd(madd(k1P, k2P, 2), k3P, 2), k4P)
    return vadd(m, km, dt / 6), madd(P, kP, dt / 6)


def run_filter(step, m0, P0, ys):
    mf, Pf, nll = m0, P0, mpf(0)
    out = []
    for y in ys:
        mf, Pf, inc = step(mf, Pf, y)
        nll += inc
        out.append((mf, Pf, nll))
    return out


def run_smoother(step, filt):
    ms, Ps = filt[-1][0], filt[-1][1]
    out = [(ms, Ps)]
    for mf, Pf, _ in reversed(filt[:-1]):
        ms, Ps = step(ms, Ps, mf, Pf)
        out.append((ms, Ps))
    return out[::-1]


def pipelines(params, Xi, dt, ys, cd_T):
    drift, b, cond, m0, P0, H = build_chirp_model(params)
    Xi, dt = mpf(float(Xi)), mpf(float(dt))
    ys = [mpf(float(y)) for y in ys]
    sg = GH3()
    gamma = matmul(b, tr(b))
    res = {}

    def ekf_step(mf, Pf, y):                                            # :222-264
        J = jacobian(lambda u: cond(u, dt)[0], mf)
        mp_, Sig = cond(mf, dt)
        return linear_update(mp_, madd(matmul(matmul(J, Pf), tr(J)), Sig), H, Xi, y)

    def eks_step(ms, Ps, mf, Pf):                                       # :317-349
        J = jacobian(lambda u: cond(u, dt)[0], mf)
        mp_, Sig = cond(mf, dt)
        return smoother_common(matmul(J, Pf), mf, Pf, mp_, madd(matmul(matmul(J, Pf), tr(J)), Sig), ms, Ps)
    f = run_filter(ekf_step, m0, P0, ys)
    res['ekf'], res['eks'] = f, run_smoother(eks_step, f)
    print('  ekf + eks done', flush=True)

    def sgpf_step(mf, Pf, y):                                           # :446-490
        mp_, Pp, _, _ = sgp_prediction(sg, cond, dt, mf, Pf)
        return linear_update(mp_, Pp, H, Xi, y)

    def sgps_step(ms, Ps, mf, Pf):                                      # :493-531
        mp_, Pp, chi, fm = sgp_prediction(sg, cond, dt, mf, Pf)
        D = madd(sg.expect_outer(chi, fm), outer(mf, mp_), -1)
        return smoother_common(tr(D), mf, Pf, mp_, Pp, ms, Ps)
    f = run_filter(sgpf_step, m0, P0, ys)
    res['sgp_filter'], res['sgp_smoother'] = f, run_smoother(sgps_step, f)
    print('  sgp_filter + sgp_smoother done', flush=True)

    def cdf_step(mf, Pf, y):                                            # :534-582
        mp_, Pp = rk4(lambda m, P: cd_sgp_common(sg, drift, b, m, P), mf, Pf, dt)
        return linear_update(mp_, Pp, H, Xi, y)

    def cds_ode(m, P, mf, Pf):                                          # :585-632
        G = cho_solve(Pf, gamma)
        _m, _P = cd_sgp_common(sg, drift, b, m, P)
        return vadd(_m, matvec(tr(G), vadd(m, mf, -1))), madd(madd(madd(_P, matmul(tr(G), P)), matmul(P, G)), mscale(gamma, 2), -1)

    def cds_step(ms, Ps, mf, Pf):
        return rk4(cds_ode, ms, Ps, -dt, mf, Pf)
    f = run_filter(cdf_step, m0, P0, ys[:cd_T])
    res['cd_sgp_filter'], res['cd_sgp_smoother'] = f, run_smoother(cds_step, f)
    print('  cd_sgp_filter + cd_sgp_smoother done', flush=True)
    return res


def pipelines_rest(params, Xi, dt, ys, cd_T):
    """The five functions the first fixtures leave out -- kf + rts (:145-219), cd_ekf + cd_eks (:352-443), ekf_for_kpt (:267-314) -- so that every
    public function of filters_smoothers.py has its exact values: --rest writes tests/golden/exact_rest.npz."""
    drift, b, cond, m0, P0, H = build_chirp_model(params)
    Xi, dt = mpf(float(Xi)), mpf(float(dt))
    ys = [mpf(float(y)) for y in ys]
    gamma = matmul(b, tr(b))
    res, extra = {}, {}

    # ---- kf + rts on the chirp model frozen at its initial frequency state (the linear test model of bench.py's C1): F = d mean / d u there
    F = jacobian(lambda u: cond(u, dt)[0], m0)
    Sig = cond(m0, dt)[1]

    def kf_step(mf, Pf, y):                                             # :48-68, 145-184
        return linear_update(matvec(F, mf), madd(matmul(matmul(F, Pf), tr(F)), Sig), H, Xi, y)

    def rts_step(ms, Ps, mf, Pf):                                       # :187-219
        return smoother_common(matmul(F, Pf), mf, Pf, matvec(F, mf), madd(matmul(matmul(F, Pf), tr(F)), Sig), ms, Ps)
    f = run_filter(kf_step, m0, P0, ys)
    res['kf'], res['rts'] = f, run_smoother(rts_step, f)
    extra['kf.F'], extra['kf.Sigma'] = [[float(v) for v in r] for r in F], [[float(v) for v in r] for r in Sig]
    print('  kf + rts done', flush=True)

    # ---- cd_ekf + cd_eks
    def cde_ode(m, P):                                                  # :384-385
        J = jacobian(drift, m)
        return drift(m), madd(madd(matmul(P, tr(J)), matmul(J, P)), gamma)

    def cde_step(mf, Pf, y):
        mp_, Pp = rk4(cde_ode, mf, Pf, dt)
        return linear_update(mp_, Pp, H, Xi, y)

    def cds_ode(m, P, mf, Pf):                                          # :427-432
        A = madd(jacobian(drift, m), tr(cho_solve(Pf, tr(gamma))))
        sol = cho_solve(Pf, [[v] for v in vadd(m, mf, -1)])
        return vadd(drift(m), matvec(gamma, [r[0] for r in sol])), madd(madd(matmul(A, P), matmul(P, tr(A))), gamma, -1)

    def cds_step(ms, Ps, mf, Pf):
        return rk4(cds_ode, ms, Ps, -dt, mf, Pf)
    f = run_filter(cde_step, m0, P0, ys[:cd_T])
    res['cd_ekf'], res['cd_eks'] = f, run_smoother(cds_step, f)
    print('  cd_ekf + cd_eks done', flush=True)

    # ---- ekf_for_kpt on build_kpt_chirp_model((0.5, 1e-4, 0.1, 8, 1), fs = 1 / dt, two harmonics) (models.py:522-580)
    q1, q2, p0, f0, a0 = (mpf(v) for v in ('0.5', '0.0001', '0.1', '8', '1'))
    fs_, nh, d = 1 / dt, 2, 4
    Fk = [[mpf(1 if i == j else 0) for j in range(d)] for i in range(d)]
    Fk[d - 1][0] = mpf(1)
    Sk = zeros(d, d)
    Sk[0][0] = (2 * mp.pi * q1 / fs_) ** 2
    for k in range(1, nh + 1):
        Sk[k][k] = q2
    m0k = [2 * mp.pi * f0 / fs_] + [a0] * nh + [mpf(0)]
    P0k = [[p0 if i == j else mpf(0) for j in range(d)] for i in range(d)]

    def h(x):
        ph = g(x[0] + x[d - 1])
        return sum(x[k] * mp.sin(ph * k) for k in range(1, nh + 1))

    def kpt_step(mf, Pf, y):                                            # :298-311
        mp_, Pp = matvec(Fk, mf), madd(matmul(matmul(Fk, Pf), tr(Fk)), Sk)
        Hk = jacobian(lambda x: [h(x)], mp_)[0]
        PH = matvec(Pp, Hk)
        S = sum(a * v for a, v in zip(Hk, PH)) + Xi
        K = [v / S for v in PH]
        pred = h(mp_)
        scale2 = mp.sqrt(S) ** 2
        nll = (mp.log(2 * mp.pi * scale2) + (y - pred) ** 2 / scale2) / 2
        return vadd(mp_, [k * (y - pred) for k in K]), madd(Pp, mscale(outer(K, K), S), -1), nll
    res['ekf_for_kpt'] = run_filter(kpt_step, m0k, P0k, ys)
    print('  ekf_for_kpt done', flush=True)
    return res, extra


class Cubature(GH3):
    """SigmaPoints.cubature(d), quadratures.py:138-150: +- sqrt(d) e_k, weights 1 / (2 d)"""
    def __init__(self, d):
        r = mp.sqrt(d)
        self.xi = [[(r if j == k else mpf(0)) for j in range(d)] for k in range(d)] + [[(-r if j == k else mpf(0)) for j in range(d)] for k in range(d)]
        self.w = [mpf(1) / (2 * d)] * (2 * d)


def build_harmonic_chirp_model(params, nh):
    """models.py:462-494 with 122-178 and 332-386 (freq_scale = 1): (cond_m_cov, m0, P0, H), d = 2 nh + 2"""
    lam, b, delta, ell, sigma, m0_v = (mpf(float(p)) for p in params)

    def cond_m_cov(u, dt):
        w = 2 * mp.pi * g(u[-2])
        e = mp.exp(-lam * dt)
        Fm, Sm = m32_solution(ell, sigma, dt)
        mean = []
        for k in range(1, nh + 1):
            c, s_ = mp.cos(dt * k * w), mp.sin(dt * k * w)
            a, bb = u[2 * k - 2], u[2 * k - 1]
            mean += [e * (c * a - s_ * bb), e * (s_ * a + c * bb)]
        mean += [Fm[0][0] * u[-2] + Fm[0][1] * u[-1], Fm[1][0] * u[-2] + Fm[1][1] * u[-1]]
        q = b ** 2 * dt if lam == 0 else b ** 2 / (2 * lam) * (1 - mp.exp(-2 * lam * dt))
        return mean, blkdiag(*([q] * (2 * nh)), Sm)
    m0 = [mpf(0), mpf(1)] * nh + [m0_v, mpf(0)]
    P0 = blkdiag(*([delta] * (2 * nh)), [[sigma ** 2, mpf(0)], [mpf(0), (mp.sqrt(3) / ell) ** 2 * sigma ** 2]])
    H = [mpf(0), mpf(1)] * nh + [mpf(0), mpf(0)]
    return cond_m_cov, m0, P0, H


def pipelines_harmonic(params, Xi, dt, ys, nh=3):
    """BASELINE C5's methods -- sgp_filter + sgp_smoother with the cubature rule on the three-harmonic model, d = 8 (demos/ghfs_harmonics_mle.py:25-27)
    -- and ekf + eks on the same model: --harmonic writes tests/golden/exact_harmonic.npz."""
    cond, m0, P0, H = build_harmonic_chirp_model(params, nh)
    Xi, dt = mpf(float(Xi)), mpf(float(dt))
    ys = [mpf(float(y)) for y in ys]
    sg = Cubature(2 * nh + 2)
    res = {}

    def ekf_step(mf, Pf, y):
        J = jacobian(lambda u: cond(u, dt)[0], mf)
        mp_, Sig = cond(mf, dt)
        return linear_update(mp_, madd(matmul(matmul(J, Pf), tr(J)), Sig), H, Xi, y)

    def eks_step(ms, Ps, mf, Pf):
        J = jacobian(lambda u: cond(u, dt)[0], mf)
        mp_, Sig = cond(mf, dt)
        return smoother_common(matmul(J, Pf), mf, Pf, mp_, madd(matmul(matmul(J, Pf), tr(J)), Sig), ms, Ps)
    f = run_filter(ekf_step, m0, P0, ys)
    res['ekf'], res['eks'] = f, run_smoother(eks_step, f)
    print('  ekf + eks (d = 8) done', flush=True)

    def sgpf_step(mf, Pf, y):
        mp_, Pp, _, _ = sgp_prediction(sg, cond, dt, mf, Pf)
        return linear_update(mp_, Pp, H, Xi, y)

    def sgps_step(ms, Ps, mf, Pf):
        mp_, Pp, chi, fm = sgp_prediction(sg, cond, dt, mf, Pf)
        D = madd(sg.expect_outer(chi, fm), outer(mf, mp_), -1)
        return smoother_common(tr(D), mf, Pf, mp_, Pp, ms, Ps)
    f = run_filter(sgpf_step, m0, P0, ys)
    res['sgp_filter'], res['sgp_smoother'] = f, run_smoother(sgps_step, f)
    print('  sgp_filter + sgp_smoother (cubature, d = 8) done', flush=True)
    return res


def ekf_final_nll(params, Xi, dt, ys):
    """ekf(...)[2][-1] (filters_smoothers.py:222-264) for mpf parameters: the MLE objective of demos/ekfs_mle.py:42-47"""
    drift, b, cond, m0, P0, H = build_chirp_model(params)

    def step(mf, Pf, y):
        J = jacobian(lambda u: cond(u, dt)[0], mf)
        mp_, Sig = cond(mf, dt)
        return linear_update(mp_, madd(matmul(matmul(J, Pf), tr(J)), Sig), H, Xi, y)
    return run_filter(step, m0, P0, ys)[-1][2]


def exact_gradient(params, Xi, dt, ys, h=mpf(10) ** -30):
    """d ekf_final_nll(g(theta)) / d theta at theta = g_inv(params): central differences with a step of 1e-30 in 100-digit arithmetic
    (truncation 1e-60 relative) -- the exact derivative the reference takes with jax.value_and_grad."""
    theta = [mp.log(mp.exp(mpf(float(p))) - 1) for p in params]
    Xi, dt = mpf(float(Xi)), mpf(float(dt))
    ys = [mpf(float(y)) for y in ys]
    f0 = ekf_final_nll([g(t) for t in theta], Xi, dt, ys)
    grad = []
    for k in range(len(theta)):
        tp, tm = list(theta), list(theta)
        tp[k] += h
        tm[k] -= h
        grad.append((ekf_final_nll([g(t) for t in tp], Xi, dt, ys) - ekf_final_nll([g(t) for t in tm], Xi, dt, ys)) / (2 * h))
        print(f'  d nll / d theta[{k}] done', flush=True)
    return np.array([float(t) for t in theta]), float(f0), np.array([float(v) for v in grad])


def to_f64(res):
    flat = {}
    for name, rows in res.items():
        flat[name + '.0'] = np.array([[float(v) for v in r[0]] for r in rows])
        flat[name + '.1'] = np.array([[[float(v) for v in row] for row in r[1]] for r in rows])
        if len(rows[0]) == 3:
            flat[name + '.2'] = np.array([float(r[2]) for r in rows])
    return flat


def records():
    """(name, params, Xi, dt, ys): float64 inputs, drawn with NumPy only"""
    sys.path.insert(0, ROOT)
    import bench
    T, dt = 500, 1e-3
    track = bench.chirp_batch(1, T, 4242, dt=dt, Xi=0.1)[0]
    lost = bench.chirp_batch(1, T, 4243, dt=dt, Xi=1.0, offset=5.5)[0]
    p = np.array([0.1, 0.1, 0.1, 1., 1., 7.])
    return (('exact_track', p, 0.1, dt, track), ('exact_lost', p, 1.0, dt, lost))


def main_gradient():
    """tests/golden/exact_grad.npz: value and exact gradient of the MLE objective on the two records (T = 500)"""
    out = {}
    for name, p, Xi, dt, ys in records():
        print(name, 'gradient', flush=True)
        theta, f0, grad = exact_gradient(p, Xi, dt, ys)
        out.update({f'{name}.theta': theta, f'{name}.nll': f0, f'{name}.grad': grad, f'{name}.ys': ys, f'{name}.Xi': Xi, f'{name}.dt': dt, f'{name}.params': p})
    np.savez_compressed(os.path.join(OUT, 'exact_grad.npz'), digits=mp.dps, **out)


def main_rest():
    """tests/golden/exact_rest.npz: kf, rts, cd_ekf, cd_eks, ekf_for_kpt on the tracking record"""
    name, p, Xi, dt, ys = records()[0]
    print(name, 'rest', flush=True)
    res, extra = pipelines_rest(p, Xi, dt, ys, cd_T=300)
    np.savez_compressed(os.path.join(OUT, 'exact_rest.npz'), ys=ys, params=p, Xi=Xi, dt=dt, digits=mp.dps, **{k.replace('.', '_'): np.array(v) for k, v in extra.items()}, **to_f64(res))


def main_harmonic():
    """tests/golden/exact_harmonic.npz: the three-harmonic model (d = 8) on tests/cases.py:harmonic_case's record (T = 300)"""
    sys.path.insert(0, ROOT)
    from tests import cases
    c = cases.harmonic_case(T=300, seed=12, nh=3)
    p = np.array([0.1, 0.1, 0.1, 1., 1., 7.])
    print('exact_harmonic', flush=True)
    res = pipelines_harmonic(p, c.Xi, c.dt, c.ys)
    np.savez_compressed(os.path.join(OUT, 'exact_harmonic.npz'), ys=c.ys, params=p, Xi=c.Xi, dt=c.dt, nh=3, digits=mp.dps, **to_f64(res))


def main_lascala():
    """tests/golden/exact_lascala.npz: the La Scala model (models.py:181-261, 419-434, 497-519) -- the chirp model without damping and without chirp
    noise: its drift, dispersion and discretisation are build_chirp_model's at lam = b = 0 (the lam == 0 branch of :302-308 gives q = 0) -- through
    the same six pipelines on the first record (tetralith/jobs/lascala_ekfs_mle.py, lascala_ghfs_mle.py)"""
    name, _, Xi, dt, ys = records()[0]
    p = np.array([0.1, 1., 1., 7.])                                          # delta, ell, sigma, m0_v
    print('exact_lascala', flush=True)
    res = pipelines([0., 0., p[0], p[1], p[2], p[3]], Xi, dt, ys, cd_T=300)
    np.savez_compressed(os.path.join(OUT, 'exact_lascala.npz'), ys=ys, params=p, Xi=Xi, dt=dt, digits=mp.dps, **to_f64(res))


def main():
    if '--grad' in sys.argv:
        return main_gradient()
    if '--lascala' in sys.argv:
        return main_lascala()
    if '--harmonic' in sys.argv:
        return main_harmonic()
    if '--rest' in sys.argv:
        return main_rest()
    for name, p, Xi, dt, ys in records():
        print(name, flush=True)
        res = pipelines(p, Xi, dt, ys, cd_T=300)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), ys=ys, params=p, Xi=Xi, dt=dt, digits=mp.dps, **to_f64(res))


if __name__ == '__main__':
    main()
