"""NumPy float64 restatement of /root/reference/chirpgp/filters_smoothers.py (TEST INFRASTRUCTURE).

One trial at a time, plain Python loops over T, generic callables exactly like
the reference (``cond_m_cov(u, dt)``, ``a(u)``, ``b(u)``, ``h(x)``).  Batched
helpers at the bottom loop over the leading axis (the restatement of
``jax.vmap(..., in_axes=0)`` over ``ys``: tetralith/jobs/crlb_ekf.py:68-72).

jax.jacfwd   -> complex-step differentiation (exact to rounding; h = 1e-30)
lax.scan     -> for-loop;   reverse=True -> reversed for-loop
cholesky / cho_factor / cho_solve -> scipy.linalg (LAPACK potrf/potrs, as XLA:CPU);
                a failed factorisation yields an all-NaN factor (JAX semantics), never an exception.
"""
import math
import numpy as np
import scipy.linalg
from oracle.np_quadratures import rk4_m_cov, rk4_m_cov_backward

__all__ = ['kf', 'rts', 'ekf', 'ekf_for_kpt', 'eks', 'cd_ekf', 'cd_eks',
           'sgp_filter', 'sgp_smoother', 'cd_sgp_filter', 'cd_sgp_smoother',
           'jacobian', 'batched']


# --------------------------------------------------------------------------- primitives
def jacobian(f, x, h=1e-30):
    """d f / d x by the complex-step method: J[:, j] = Im f(x + i h e_j) / h."""
    x = np.asarray(x, dtype=np.float64)
    cols = []
    for j in range(x.size):
        xc = x.astype(np.complex128)
        xc[j] += 1j * h
        cols.append(np.imag(np.asarray(f(xc))) / h)
    return np.stack(cols, axis=-1)


def _chol_lower(P):
    """jax.scipy.linalg.cholesky(P, lower=True): NaN-filled on failure."""
    try:
        if not np.all(np.isfinite(P)):
            raise np.linalg.LinAlgError
        return scipy.linalg.cholesky(P, lower=True)
    except (np.linalg.LinAlgError, ValueError):
        return np.full_like(P, np.nan)


def _cho_solve(P, rhs):
    """cho_solve(cho_factor(P), rhs) with JAX's NaN-on-failure semantics."""
    try:
        if not np.all(np.isfinite(P)):
            raise np.linalg.LinAlgError
        return scipy.linalg.cho_solve(scipy.linalg.cho_factor(P), rhs)
    except (np.linalg.LinAlgError, ValueError):
        return np.full(np.shape(rhs), np.nan)


def _neg_log_normal_pdf(x, mu, variance):
    """-norm.logpdf(x, mu, sqrt(variance)), filters_smoothers.py:44-45.  Written as jax.scipy.stats.norm.logpdf
    evaluates it: -(log(2 pi scale^2) + (x - mu)^2 / scale^2) / -2 with scale = sqrt(variance)."""
    with np.errstate(invalid='ignore', divide='ignore'):
        scale = np.sqrt(variance)
        s2 = scale * scale
        return (np.log(2 * math.pi * s2) + (x - mu) ** 2 / s2) / 2


def _linear_predict(F, Sigma, m, P):
    """filters_smoothers.py:48-52."""
    return F @ m, F @ P @ F.T + Sigma


def _linear_update(mp, Pp, H, Xi, y):
    """filters_smoothers.py:55-68 (scalar measurement; Pf = Pp - K K^T S, not Joseph form)."""
    S = H @ Pp @ H.T + Xi
    K = Pp @ H.T / S
    pred = H @ mp
    return mp + K * (y - pred), Pp - np.outer(K, K) * S, _neg_log_normal_pdf(y, pred, S)


def _gaussian_smoother_common(DT, mf, Pf, mp, Pp, ms, Ps):
    """filters_smoothers.py:71-85.  DT is the transpose of the cross-covariance D."""
    G = _cho_solve(Pp, DT).T
    return mf + G @ (ms - mp), Pf + G @ (Ps - Pp) @ G.T


def _sgp_prediction(sgps, cond_m_cov, dt, mf, Pf):
    """filters_smoothers.py:88-121."""
    chi = sgps.gen_sigma_points(mf, _chol_lower(Pf))
    evals = [cond_m_cov(c, dt) for c in chi]
    fm = np.stack([e[0] for e in evals])
    fc = np.stack([e[1] for e in evals])
    mp = sgps.expectation(fm)
    Pp = sgps.expectation(fm[:, :, None] * fm[:, None, :] + fc) - np.outer(mp, mp)
    return mp, Pp, chi, fm


def _cd_sgp_common(sgps, drift, dispersion_const, m, P):
    """filters_smoothers.py:124-137."""
    chi = sgps.gen_sigma_points(m, _chol_lower(P))
    fa = np.stack([drift(c) for c in chi])
    mp = sgps.expectation(fa)
    _Pp = sgps.expectation((chi - m)[:, :, None] * fa[:, None, :])
    return mp, _Pp + _Pp.T + dispersion_const @ dispersion_const.T


def _stack(mfs, Pfs, mss, Pss):
    """filters_smoothers.py:140-142: the last smoothing row is the last filtering row."""
    d = mfs.shape[1]
    mss = np.asarray(mss, dtype=np.float64).reshape(-1, d)
    Pss = np.asarray(Pss, dtype=np.float64).reshape(-1, d, d)
    return np.vstack([mss, mfs[-1]]), np.vstack([Pss, Pfs[-1, None]])


def _run_filter(step, m0, P0, ys):
    """lax.scan(scan_body, (m0, P0, 0.), ys) emitting (mf, Pf, cumulative nll)."""
    ys = np.asarray(ys, dtype=np.float64)
    T, d = ys.shape[0], np.size(m0)
    mfs, Pfs, nlls = np.zeros((T, d)), np.zeros((T, d, d)), np.zeros(T)
    mf, Pf, nll = np.asarray(m0, dtype=np.float64), np.asarray(P0, dtype=np.float64), 0.
    with np.errstate(all='ignore'):
        for k in range(T):
            mf, Pf, inc = step(mf, Pf, ys[k])
            nll = nll + inc
            mfs[k], Pfs[k], nlls[k] = mf, Pf, nll
    return mfs, Pfs, nlls


def _run_smoother(step, mfs, Pfs):
    """Reverse scan over (mfs[:-1], Pfs[:-1]) from carry (mfs[-1], Pfs[-1])."""
    mfs, Pfs = np.asarray(mfs, dtype=np.float64), np.asarray(Pfs, dtype=np.float64)
    T = mfs.shape[0]
    mss, Pss = np.zeros_like(mfs[:-1]), np.zeros_like(Pfs[:-1])
    ms, Ps = mfs[-1], Pfs[-1]
    with np.errstate(all='ignore'):
        for k in range(T - 2, -1, -1):
            ms, Ps = step(ms, Ps, mfs[k], Pfs[k])
            mss[k], Pss[k] = ms, Ps
    return _stack(mfs, Pfs, mss, Pss)


# --------------------------------------------------------------------------- the 11 public functions
def kf(F, Sigma, H, Xi, m0, P0, ys):
    """filters_smoothers.py:145-184."""
    def step(mf, Pf, y):
        mp, Pp = _linear_predict(F, Sigma, mf, Pf)
        return _linear_update(mp, Pp, H, Xi, y)
    return _run_filter(step, m0, P0, ys)


def rts(F, Sigma, mfs, Pfs):
    """filters_smoothers.py:187-219."""
    def step(ms, Ps, mf, Pf):
        return _gaussian_smoother_common(F @ Pf, mf, Pf, F @ mf, F @ Pf @ F.T + Sigma, ms, Ps)
    return _run_smoother(step, mfs, Pfs)


def ekf(cond_m_cov, H, Xi, m0, P0, dt, ys):
    """filters_smoothers.py:222-264."""
    def step(mf, Pf, y):
        jac_F = jacobian(lambda u: cond_m_cov(u, dt)[0], mf)
        mp, Sigma = cond_m_cov(mf, dt)
        Pp = jac_F @ Pf @ jac_F.T + Sigma
        return _linear_update(mp, Pp, H, Xi, y)
    return _run_filter(step, m0, P0, ys)


def ekf_for_kpt(F, Sigma, h, Xi, m0, P0, dt, ys):
    """filters_smoothers.py:267-314 (linear dynamics, nonlinear scalar measurement h)."""
    def step(mf, Pf, y):
        mp, Pp = _linear_predict(F, Sigma, mf, Pf)
        H = jacobian(lambda x: np.atleast_1d(h(x)), mp)[0]
        S = H @ Pp @ H.T + Xi
        K = Pp @ H.T / S
        pred = h(mp)
        return mp + K * (y - pred), Pp - np.outer(K, K) * S, _neg_log_normal_pdf(y, pred, S)
    return _run_filter(step, m0, P0, ys)


def eks(cond_m_cov, mfs, Pfs, dt):
    """filters_smoothers.py:317-349."""
    def step(ms, Ps, mf, Pf):
        jac_F = jacobian(lambda u: cond_m_cov(u, dt)[0], mf)
        mp, Sigma = cond_m_cov(mf, dt)
        Pp = jac_F @ Pf @ jac_F.T + Sigma
        return _gaussian_smoother_common(jac_F @ Pf, mf, Pf, mp, Pp, ms, Ps)
    return _run_smoother(step, mfs, Pfs)


def cd_ekf(a, b, H, Xi, m0, P0, dt, ys):
    """filters_smoothers.py:352-397."""
    def odes(m, P):
        J = jacobian(a, m)
        bm = b(m)
        return a(m), P @ J.T + J @ P + bm @ bm.T

    def step(mf, Pf, y):
        mp, Pp = rk4_m_cov(odes, mf, Pf, dt)
        return _linear_update(mp, Pp, H, Xi, y)
    return _run_filter(step, m0, P0, ys)


def cd_eks(a, b, mfs, Pfs, dt):
    """filters_smoothers.py:400-443 (dt negated; (mf, Pf) held fixed over the 4 RK4 stages)."""
    dt = -dt

    def odes(m, P, mf, Pf):
        bm = b(m)
        gamma = bm @ bm.T
        A = jacobian(a, m) + _cho_solve(Pf, gamma.T).T
        return a(m) + gamma @ _cho_solve(Pf, m - mf), A @ P + P @ A.T - gamma

    def step(ms, Ps, mf, Pf):
        return rk4_m_cov_backward(odes, ms, Ps, mf, Pf, dt)
    return _run_smoother(step, mfs, Pfs)


def sgp_filter(cond_m_cov, sgps, H, Xi, m0, P0, dt, ys):
    """filters_smoothers.py:446-490."""
    def step(mf, Pf, y):
        mp, Pp, _, _ = _sgp_prediction(sgps, cond_m_cov, dt, mf, Pf)
        return _linear_update(mp, Pp, H, Xi, y)
    return _run_filter(step, m0, P0, ys)


def sgp_smoother(cond_m_cov, sgps, mfs, Pfs, dt):
    """filters_smoothers.py:493-531."""
    def step(ms, Ps, mf, Pf):
        mp, Pp, chi, fm = _sgp_prediction(sgps, cond_m_cov, dt, mf, Pf)
        D = sgps.expectation(chi[:, :, None] * fm[:, None, :]) - np.outer(mf, mp)
        return _gaussian_smoother_common(D.T, mf, Pf, mp, Pp, ms, Ps)
    return _run_smoother(step, mfs, Pfs)


def cd_sgp_filter(a, b, sgps, H, Xi, m0, P0, dt, ys):
    """filters_smoothers.py:534-582 (b is a constant (d, dw) matrix)."""
    def odes(m, P):
        return _cd_sgp_common(sgps, a, b, m, P)

    def step(mf, Pf, y):
        mp, Pp = rk4_m_cov(odes, mf, Pf, dt)
        return _linear_update(mp, Pp, H, Xi, y)
    return _run_filter(step, m0, P0, ys)


def cd_sgp_smoother(a, b, sgps, mfs, Pfs, dt):
    """filters_smoothers.py:585-632."""
    dt = -dt

    def odes(m, P, mf, Pf):
        gamma = b @ b.T
        G = _cho_solve(Pf, gamma)
        _m, _P = _cd_sgp_common(sgps, a, b, m, P)
        return _m + G.T @ (m - mf), _P + G.T @ P + P @ G - 2 * gamma

    def step(ms, Ps, mf, Pf):
        return rk4_m_cov_backward(odes, ms, Ps, mf, Pf, dt)
    return _run_smoother(step, mfs, Pfs)


# --------------------------------------------------------------------------- vmap restatement
def batched(fn, batch_argnums, *args):
    """jax.vmap(fn, in_axes=0 on batch_argnums) as a Python loop; stacks every output."""
    B = np.shape(args[batch_argnums[0]])[0]
    outs = []
    for i in range(B):
        a = [arg[i] if k in batch_argnums else arg for k, arg in enumerate(args)]
        outs.append(fn(*a))
    return tuple(np.stack([o[j] for o in outs]) for j in range(len(outs[0])))
