"""Maximum-likelihood parameter estimation through the filters -- the step every driver of the reference runs before
filtering and smoothing (demos/ekfs_mle.py:39-51, tetralith/jobs/*_mle.py).

The reference minimises ``obj(theta) = filter(build_model(g(theta)), ys)[-1][-1]`` with jaxopt's L-BFGS-B wrapper and
reverse-mode autodiff THROUGH the scan.  A HIP kernel is not JAX-differentiable; instead the engine's strengths are
used: the filters take one parameter vector per trial and an NLL-only output mode, so the objective and its central
finite-difference gradient -- 2 P + 1 filter passes over the same measurements -- are ONE kernel launch with a batch of
2 P + 1 "trials" (13 for the chirp model's 6 parameters).  The optimiser itself stays SciPy's L-BFGS-B on the host,
exactly the algorithm the reference uses.
"""
import numpy as np

from chirpgp_amd import filters_smoothers as fs
from chirpgp_amd import models as M

__all__ = ['batched_nll', 'make_objective', 'fit']


def batched_nll(method, build, thetas, ys, Xi, dt, sgps=None, **build_kw):
    """Final cumulative NLL of ``method`` for every row of ``thetas`` (unconstrained parameters, g() maps them to the
    positive model parameters as in the reference) on the SAME measurement record ``ys`` (T,).

    method: 'ekf' | 'sgp_filter' | 'cd_ekf' | 'cd_sgp_filter';  build: e.g. models.build_chirp_model."""
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    G = thetas.shape[0]
    drift, disp, disc, m0, P0, H = build(M.g(thetas), **build_kw)
    ysb = np.broadcast_to(np.asarray(ys, dtype=np.float64), (G, np.size(ys)))
    kw = dict(nll_final_only=True, want=(False, False, True))
    if method == 'ekf':
        out = fs.ekf(disc, H, Xi, m0, P0, dt, ysb, **kw)
    elif method == 'sgp_filter':
        out = fs.sgp_filter(disc, sgps, H, Xi, m0, P0, dt, ysb, **kw)
    elif method == 'cd_ekf':
        out = fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ysb, **kw)
    elif method == 'cd_sgp_filter':
        out = fs.cd_sgp_filter(drift, disp, sgps, H, Xi, m0, P0, dt, ysb, **kw)
    else:
        raise ValueError(method)
    return np.asarray(out[2])


def make_objective(method, build, ys, Xi, dt, sgps=None, rel_step=1e-6, **build_kw):
    """-> fun(theta) returning (nll, gradient): value and central differences from one batched launch."""
    def fun(theta):
        theta = np.asarray(theta, dtype=np.float64)
        P = theta.size
        h = rel_step * (1.0 + np.abs(theta))
        batch = np.tile(theta, (2 * P + 1, 1))
        for i in range(P):
            batch[1 + 2 * i, i] += h[i]
            batch[2 + 2 * i, i] -= h[i]
        nll = batched_nll(method, build, batch, ys, Xi, dt, sgps, **build_kw)
        grad = (nll[1::2] - nll[2::2]) / (2 * h)
        f = float(nll[0])
        if not np.isfinite(f):                      # diverged filter: the reference writes NaN results and moves on
            return np.inf, np.zeros(P)
        return f, np.where(np.isfinite(grad), grad, 0.0)
    return fun


def fit(method, build, init_params, ys, Xi, dt, sgps=None, maxiter=200, **build_kw):
    """L-BFGS-B from ``init_params`` (positive model parameters, e.g. [0.1, 0.1, 0.1, 1, 1, 7]).
    Returns (opt_params, scipy OptimizeResult)."""
    from scipy.optimize import minimize
    fun = make_objective(method, build, ys, Xi, dt, sgps, **build_kw)
    res = minimize(fun, M.g_inv(np.asarray(init_params, dtype=np.float64)), jac=True, method='L-BFGS-B',
                   options=dict(maxiter=maxiter))
    return M.g(res.x), res
