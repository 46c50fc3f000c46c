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

__all__ = ['batched_nll', 'make_objective', 'fit', 'fit_many']


def batched_nll(method, build, thetas, ys, Xi, dt, sgps=None, **build_kw):
    """Final cumulative NLL of ``method`` for every row of ``thetas`` (unconstrained parameters, g() maps them to the
    positive model parameters as in the reference) on the SAME measurement record ``ys`` (T,), or on its own record
    when ``ys`` is (G, T).

    method: 'ekf' | 'sgp_filter' | 'cd_ekf' | 'cd_sgp_filter';  build: e.g. models.build_chirp_model."""
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    G = thetas.shape[0]
    drift, disp, disc, m0, P0, H = build(M.g(thetas), **build_kw)
    if type(ys).__module__.startswith('torch'):                               # records already in HBM (fit_many)
        ysb = ys if ys.ndim == 2 else ys[None, :].expand(G, -1)
    else:
        ys = np.asarray(ys, dtype=np.float64)
        ysb = np.broadcast_to(ys, (G, ys.shape[-1])) if ys.ndim == 1 else ys  # one record for all rows, or one per row
    if ysb.shape[0] != G:
        raise ValueError(f'ys has {ysb.shape[0]} records for {G} parameter vectors')
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
    nll = out[2]
    return nll.cpu().numpy() if type(nll).__module__.startswith('torch') else np.asarray(nll)


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


def _value_and_grad_many(method, build, thetas, yss, Xi, dt, sgps, rel_step, build_kw):
    """NLL and central-difference gradient of R records at R parameter vectors: ONE launch of R (2 P + 1) trials."""
    R, P = thetas.shape
    h = rel_step * (1.0 + np.abs(thetas))                                   # (R, P)
    batch = np.repeat(thetas[:, None, :], 2 * P + 1, axis=1)                # (R, 2P+1, P)
    idx = np.arange(P)
    batch[:, 1 + 2 * idx, idx] += h
    batch[:, 2 + 2 * idx, idx] -= h
    # record r serves its 2P+1 rows; the records live in HBM, so the replication is a device-side copy
    ys_rep = yss.repeat_interleave(2 * P + 1, dim=0) if type(yss).__module__.startswith('torch') else np.repeat(yss, 2 * P + 1, axis=0)
    nll = batched_nll(method, build, batch.reshape(-1, P), ys_rep, Xi, dt, sgps, **build_kw).reshape(R, 2 * P + 1)
    f = nll[:, 0].copy()
    grad = (nll[:, 1::2] - nll[:, 2::2]) / (2 * h)
    f[~np.isfinite(f)] = np.inf
    return f, np.where(np.isfinite(grad), grad, 0.0)


def fit_many(method, build, init_params, yss, Xi, dt, sgps=None, maxiter=200, history=10, gtol=1e-5, ftol=2.2e-9,
             rel_step=1e-6, **build_kw):
    """Maximum likelihood for R measurement records in lock step: limited-memory BFGS with a backtracking (Armijo) line
    search, every record with its own iterate, history and step length, and every probe of every record evaluated in
    the SAME kernel launch (R x 13 trials for the chirp model).  A launch costs T x 0.35 us whatever the batch up to
    ~4000 trials, so R records take the wall time of one -- the reference's Monte-Carlo jobs (tetralith/jobs/*_mle.py)
    run one L-BFGS-B per record.  Same objective, same unconstrained parametrisation g() as :func:`fit`.

    yss (R, T);  init_params (P,) or (R, P) positive model parameters  ->  (opt_params (R, P), info dict)."""
    from chirpgp_amd import _engine as E
    yss = E.dev(yss)                                  # uploaded once; every probe replicates rows on the device
    if yss.ndim == 1:
        yss = yss[None, :]
    R = yss.shape[0]
    x = np.array(np.broadcast_to(M.g_inv(np.asarray(init_params, dtype=np.float64)), (R, np.shape(init_params)[-1])))
    P = x.shape[1]
    f, g = _value_and_grad_many(method, build, x, yss, Xi, dt, sgps, rel_step, build_kw)
    S, Y = [], []                                    # lists of (R, P) pairs, newest last
    done = ~np.isfinite(f)
    nit = np.zeros(R, dtype=int)
    launches = 1
    for _ in range(maxiter):
        if done.all():
            break
        # two-loop recursion, vectorised over the records; pairs with s . y <= 0 carry rho = 0 and drop out
        q = g.copy()
        alphas = []
        for s_, y_ in zip(reversed(S), reversed(Y)):
            sy = np.einsum('rp,rp->r', s_, y_)
            rho = np.where(sy > 1e-300, 1.0 / np.where(sy > 1e-300, sy, 1.0), 0.0)
            a = rho * np.einsum('rp,rp->r', s_, q)
            q -= a[:, None] * y_
            alphas.append((a, rho))
        if S:
            sy = np.einsum('rp,rp->r', S[-1], Y[-1]); yy = np.einsum('rp,rp->r', Y[-1], Y[-1])
            q *= np.where((sy > 1e-300) & (yy > 0), sy / np.where(yy > 0, yy, 1.0), 1.0)[:, None]
        for (a, rho), s_, y_ in zip(reversed(alphas), S, Y):
            b = rho * np.einsum('rp,rp->r', y_, q)
            q += (a - b)[:, None] * s_
        d = -q
        gd = np.einsum('rp,rp->r', g, d)
        bad = ~(gd < 0)                               # not a descent direction (or NaN): steepest descent
        d[bad] = -g[bad]; gd[bad] = -np.einsum('rp,rp->r', g[bad], g[bad])
        step = np.ones(R) if S else np.minimum(1.0, 1.0 / np.maximum(np.abs(g).sum(axis=1), 1e-300))
        searching = ~done
        x_new, f_new, g_new = x.copy(), f.copy(), g.copy()
        for _ls in range(25):
            if not searching.any():
                break
            idx = np.flatnonzero(searching)
            xt = x[idx] + step[idx, None] * d[idx]
            ft, gt = _value_and_grad_many(method, build, xt, yss[E.torch_index(idx, yss)], Xi, dt, sgps, rel_step, build_kw)
            launches += 1
            ok = ft <= f[idx] + 1e-4 * step[idx] * gd[idx]
            acc = idx[ok]
            x_new[acc], f_new[acc], g_new[acc] = xt[ok], ft[ok], gt[ok]
            searching[acc] = False
            step[idx[~ok]] *= 0.5
        failed = searching.copy()                     # line search exhausted: this record stops where it is
        moved = ~done & ~failed
        s_, y_ = x_new - x, g_new - g
        s_[~moved] = 0.0; y_[~moved] = 0.0
        S.append(s_); Y.append(y_)
        if len(S) > history:
            S.pop(0); Y.pop(0)
        small = (f - f_new) <= ftol * np.maximum(np.maximum(np.abs(f), np.abs(f_new)), 1.0)
        x, f, g = x_new, f_new, g_new
        nit[moved] += 1
        done |= failed | (moved & (small | (np.abs(g).max(axis=1) <= gtol)))
    return M.g(x), dict(fun=f, grad=g, nit=nit, launches=launches, converged=done)
