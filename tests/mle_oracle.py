"""The MLE objective of the reference's drivers (demos/ekfs_mle.py:42-51, ghfs_mle.py:56-60, cd_ghfs_mle.py:47-50) on the
ORACLE: the final cumulative negative log-likelihood of a filter of oracle/c/port.c as a function of the unconstrained
parameters theta (model parameters = g(theta)), and its gradient by a fourth-order central difference -- the reference
differentiates through the scan with JAX; a difference quotient of the checker's own objective is what is available here
and, at O(h^4) truncation, accurate to ~1e-8 relative.  TEST INFRASTRUCTURE (only tests/ imports it)."""
import copy
import numpy as np

from oracle import port

METHOD = {'ekf': port.F_EKF, 'sgp_filter': port.F_SGP, 'cd_ekf': port.F_CD_EKF, 'cd_sgp_filter': port.F_CD_SGP,
          'ekf_for_kpt': port.F_EKF_KPT}


def g(x):
    return np.log(np.exp(x) + 1.)


def g_inv(x):
    return np.log(np.exp(x) - 1.)


def nll(method, build, thetas, ys, Xi, dt, sgps=None, **build_kw):
    """Final NLL for every row of `thetas` on the same record: one batched call of the C port (per-trial parameters)."""
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    ysb = np.broadcast_to(np.asarray(ys, dtype=np.float64), (thetas.shape[0], len(ys)))
    if method == 'ekf_for_kpt':        # tetralith/jobs/kpt_mle.py:41-44: build -> (F, Sigma, m0, P0, h); the port takes a linear
        F, Sigma, m0, P0, h = build(g(thetas), **build_kw)      # descriptor re-labelled as the KPT model, like the engine
        import copy as _copy
        from chirpgp_amd import models as pm                    # descriptor classes only (no device code behind them)
        spec = _copy.copy(pm.linear_cond_m_cov(F, Sigma))
        spec.model_id, spec.n_harm = port.M_KPT, h.n_harm
        return port.filter(port.F_EKF_KPT, spec, None, None, Xi, m0, P0, dt, ysb, nll_final_only=True)[2]
    drift, disp, disc, m0, P0, H = build(g(thetas), **build_kw)
    if method in ('ekf', 'sgp_filter'):
        model = disc
    else:
        model = copy.copy(drift)
        model.gamma = disp.outer()
    return port.filter(METHOD[method], model, sgps if 'sgp' in method else None, H, Xi, m0, P0, dt, ysb, nll_final_only=True)[2]


def value_and_grad(method, build, theta, ys, Xi, dt, sgps=None, h=1e-4, **build_kw):
    """f(theta) and the five-point central difference (-f(+2h) + 8 f(+h) - 8 f(-h) + f(-2h)) / 12 h per coordinate."""
    theta = np.asarray(theta, dtype=np.float64)
    P = theta.size
    step = h * (1.0 + np.abs(theta))
    batch = np.tile(theta, (4 * P + 1, 1))
    for i in range(P):
        for k, m in enumerate((2., 1., -1., -2.)):
            batch[1 + 4 * i + k, i] += m * step[i]
    f = nll(method, build, batch, ys, Xi, dt, sgps, **build_kw)
    fi = f[1:].reshape(P, 4)
    return float(f[0]), (-fi[:, 0] + 8 * fi[:, 1] - 8 * fi[:, 2] + fi[:, 3]) / (12 * step)


def fit(method, build, init_params, ys, Xi, dt, sgps=None, maxiter=500, **build_kw):
    """SciPy L-BFGS-B on the oracle objective from the same start as chirpgp_amd.mle.fit -> (params, OptimizeResult)."""
    from scipy.optimize import minimize
    res = minimize(lambda th: value_and_grad(method, build, th, ys, Xi, dt, sgps, **build_kw),
                   g_inv(np.asarray(init_params, dtype=np.float64)), jac=True, method='L-BFGS-B',
                   options=dict(maxiter=maxiter, ftol=1e-15, gtol=1e-9, maxls=50))
    return g(res.x), res
