"""Maximum-likelihood parameter estimation through the filters -- the step every driver of the reference runs before
filtering and smoothing (demos/ekfs_mle.py:39-51, tetralith/jobs/*_mle.py).

The reference minimises ``obj(theta) = filter(build_model(g(theta)), ys)[-1][-1]`` with jaxopt's L-BFGS-B wrapper and
reverse-mode autodiff THROUGH the scan.  A HIP kernel is not JAX-differentiable; instead the engine's strengths are
used: the filters take one parameter vector per trial and an NLL-only output mode, so the objective and its central
finite-difference gradient -- 2 P + 1 filter passes over the same measurements -- are ONE kernel launch with a batch of
2 P + 1 "trials" (13 for the chirp model's 6 parameters) that all read the ONE copy of the record in HBM.  The optimiser itself stays SciPy's L-BFGS-B on the host,
exactly the algorithm the reference uses.
"""
import numpy as np

from chirpgp_amd import filters_smoothers as fs
from chirpgp_amd import models as M

__all__ = ['batched_nll', 'make_objective', 'fit', 'fit_many', 'grid_search', 'value_and_grad', 'tangent_directions', 'has_exact_gradient']


def batched_nll(method, build, thetas, ys, Xi, dt, sgps=None, record_index=None, **build_kw):
    """Final cumulative NLL of ``method`` for every row of ``thetas`` (unconstrained parameters, g() maps them to the
    positive model parameters as in the reference).  ``ys`` is ONE record (T,) read by all G rows, or R records (R, T) of
    which each serves G / R consecutive rows (``record_index`` (n,) first picks n of them: G / n rows each).  The records
    are read in place through the C-ABI's shared-record addressing (include/chirpgp_hip.h, cgp_filter): nothing is
    replicated, on the host or on the device.

    method: 'ekf' | 'sgp_filter' | 'cd_ekf' | 'cd_sgp_filter' with a 6-tuple builder (models.build_chirp_model,
    build_harmonic_chirp_model, build_lascala_model), or 'ekf_for_kpt' with models.build_kpt_chirp_model (pass ``fs=``,
    ``num_harmonics=``): tetralith/jobs/kpt_mle.py:41-44."""
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    G = thetas.shape[0]
    n_rec = 1 if np.ndim(ys) == 1 else int(np.shape(ys)[0])
    if record_index is not None:
        n_rec = int(np.size(record_index))
    if n_rec < 1 or G % n_rec:
        raise ValueError(f'{G} parameter vectors cannot be shared out evenly over {n_rec} records')
    kw = dict(nll_final_only=True, want=(False, False, True), trials_per_record=G // n_rec, record_index=record_index)
    with np.errstate(all='ignore'):        # a probe whose parameters under- or overflow yields a NaN objective, which the line search rejects
        built = build(M.g(thetas), **build_kw)
    if method == 'ekf_for_kpt':
        F, Sigma, m0, P0, h = built
        out = fs.ekf_for_kpt(F, Sigma, h, Xi, m0, P0, dt, ys, **kw)
    else:
        drift, disp, disc, m0, P0, H = built
        if method == 'ekf':
            out = fs.ekf(disc, H, Xi, m0, P0, dt, ys, **kw)
        elif method == 'sgp_filter':
            out = fs.sgp_filter(disc, sgps, H, Xi, m0, P0, dt, ys, **kw)
        elif method == 'cd_ekf':
            out = fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ys, **kw)
        elif method == 'cd_sgp_filter':
            out = fs.cd_sgp_filter(drift, disp, sgps, H, Xi, m0, P0, dt, ys, **kw)
        else:
            raise ValueError(method)
    nll = out[2]
    return nll.cpu().numpy() if type(nll).__module__.startswith('torch') else np.asarray(nll)


# ---- exact gradients: forward tangents through the scan (include/chirpgp_hip.h: cgp_ekf_nll_grad) ------------------------------
def _m32_c(ell, sigma, dt):
    """models.py:61-73 for complex arguments (the complex step below)."""
    gamma = np.sqrt(3.) / ell
    eta = dt * gamma
    beta = sigma ** 2 * np.exp(-2 * eta)
    e = np.exp(-eta)
    F = np.array([(1 + eta) * e, dt * e, -dt * gamma ** 2 * e, (1 - eta) * e])
    off = 2 * dt ** 2 * gamma ** 3 * beta
    S = np.array([sigma ** 2 - beta * (2 * eta + 2 * eta ** 2 + 1), off, gamma ** 2 * (sigma ** 2 + beta * (2 * eta - 2 * eta ** 2 - 1))])
    return F, S


def _chirp_constants(p, dt, Xi):
    """The 24 model constants a tangent direction differentiates (include/chirpgp_hip.h: CGP_DIR_DOUBLES), as functions of the chirp
    builder's parameters lam, b, delta, ell, sigma, m0_v (models.py:437-459, 264-311, 56-58) -- complex-safe, vectorised over a
    trailing axis: p (6, G) -> (24, G)."""
    lam, b, delta, ell, sigma, m0_v = p
    safe = np.where(lam == 0., 1., lam)
    q = np.where(lam == 0., b ** 2 * dt, b ** 2 / (2 * safe) * (1 - np.exp(-2 * safe * dt)))
    F, S = _m32_c(ell, sigma, dt)
    zero = 0. * lam
    P0 = [delta, zero, delta, zero, zero, sigma ** 2, zero, zero, zero, (np.sqrt(3.) / ell) ** 2 * sigma ** 2]
    return np.stack([-lam * dt, q, *F, *S, Xi + zero, zero, zero, m0_v, zero, *P0])


def _lascala_constants(p, dt, Xi):
    """models.py:497-519: delta, ell, sigma, m0_v; no damping, no chirp noise."""
    delta, ell, sigma, m0_v = p
    F, S = _m32_c(ell, sigma, dt)
    zero = 0. * delta
    P0 = [delta, zero, delta, zero, zero, sigma ** 2, zero, zero, zero, (np.sqrt(3.) / ell) ** 2 * sigma ** 2]
    return np.stack([zero, zero, *F, *S, Xi + zero, zero, zero, m0_v, zero, *P0])


def _constants_of(build):
    return {M.build_chirp_model: _chirp_constants, M.build_lascala_model: _lascala_constants}.get(build)


def tangent_directions(build, thetas, dt, Xi, h=1e-30):
    """d (model constants) / d theta_k for every row of thetas (G, P) -> (G, P, 24): complex-step derivatives of the builder's constants
    with respect to the positive parameters (exact to rounding), times d g(theta) / d theta = sigmoid(theta) (the reference's
    parametrisation, demos/ekfs_mle.py:39-47).  (lam = 0 exactly takes the branch's own derivative, as jax.lax.cond would.)"""
    consts = _constants_of(build)
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    G, P = thetas.shape
    out = np.empty((G, P, 24))
    with np.errstate(all='ignore'):
        params = M.g(thetas).T                                   # (P, G)
        dg = 1.0 / (1.0 + np.exp(-thetas))
        for k in range(P):
            pc = params.astype(np.complex128)
            pc[k] += 1j * h
            out[:, k, :] = (np.imag(consts(pc, dt, float(Xi))) / h).T * dg[:, k, None]
    return out


def has_exact_gradient(method, build, Xi):
    """The in-kernel tangent gradient exists for the discrete EKF on the d = 4 chirp and La Scala models with a scalar Xi."""
    return method == 'ekf' and _constants_of(build) is not None and np.ndim(Xi) == 0


# Which form is the default (``exact=None``).  Measured on MI355X at T = 3141 (tools/grad_bench.py, profiles/r06_grad_bench.txt): the tangent
# kernel takes 3.7 - 4.4 ms per launch whatever the batch up to ~10 000 records (one lane per record and direction: a step is ~520
# dependent-issue vector instructions, 2800 cycles, and a wavefront issues them at the same rate for 6 lanes as for 64); the difference
# form runs 13 probes per record on the matrix-core EKF, 0.8 ms for one record, 0.94 ms for 64, 3.6 ms for 1000 and linear from there.
# So: exact where it is also the faster one -- from EXACT_FROM_RECORDS records in a launch -- and on request (exact=True) anywhere.
EXACT_FROM_RECORDS = 1500


def _exact_by_default(method, build, Xi, n_records, build_kw):
    return has_exact_gradient(method, build, Xi) and not build_kw and n_records >= EXACT_FROM_RECORDS


def value_and_grad(build, thetas, ys, Xi, dt, record_index=None):
    """EKF objective and its EXACT gradient (forward tangents through the scan, cgp_ekf_nll_grad) at every row of thetas (G, P):
    ONE launch of G P lanes; ys (T,) or (R, T) shared out evenly over the rows as in batched_nll.  -> (nll (G,), grad (G, P))."""
    from chirpgp_amd import _engine as E
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    G = thetas.shape[0]
    n_rec = 1 if np.ndim(ys) == 1 else int(np.shape(ys)[0])
    if record_index is not None:
        n_rec = int(np.size(record_index))
    if n_rec < 1 or G % n_rec:
        raise ValueError(f'{G} parameter vectors cannot be shared out evenly over {n_rec} records')
    with np.errstate(all='ignore'):
        drift, disp, disc, m0, P0, H = build(M.g(thetas))
    dirs = tangent_directions(build, thetas, dt, Xi)
    nll, grad = E.run_ekf_nll_grad(disc, H, Xi, m0, P0, dt, ys, dirs, trials_per_record=G // n_rec, record_index=record_index)
    return nll.cpu().numpy(), grad.cpu().numpy()


def make_objective(method, build, ys, Xi, dt, sgps=None, rel_step=1e-6, exact=None, **build_kw):
    """-> fun(theta) returning (nll, gradient): value and central differences from one batched launch of 2 P + 1 filter passes, or --
    ``exact=True``, the discrete EKF on the chirp / La Scala models -- value and EXACT gradient from one launch of the tangent kernel
    (cgp_ekf_nll_grad: 4e-14 of the gradient's scale against 100-digit arithmetic where the differences carry 2e-7; slower for a single
    record, see EXACT_FROM_RECORDS)."""
    if exact is None:
        exact = _exact_by_default(method, build, Xi, 1, build_kw)
    if exact:
        if not has_exact_gradient(method, build, Xi) or build_kw:
            raise ValueError('exact=True: the tangent kernel is built for the discrete EKF on build_chirp_model / build_lascala_model with a scalar Xi')
        from chirpgp_amd import _engine as E
        ys_dev = E.dev(ys)

        def fun_exact(theta):
            f, g_ = value_and_grad(build, np.asarray(theta, dtype=np.float64)[None, :], ys_dev, Xi, dt)
            if not np.isfinite(f[0]):                   # diverged filter: the reference writes NaN results and moves on
                return np.inf, np.zeros(np.size(theta))
            return float(f[0]), np.where(np.isfinite(g_[0]), g_[0], 0.0)
        return fun_exact

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


def fit(method, build, init_params, ys, Xi, dt, sgps=None, maxiter=200, exact=None, **build_kw):
    """L-BFGS-B from ``init_params`` (positive model parameters, e.g. [0.1, 0.1, 0.1, 1, 1, 7]).
    Returns (opt_params, scipy OptimizeResult)."""
    from scipy.optimize import minimize
    fun = make_objective(method, build, ys, Xi, dt, sgps, exact=exact, **build_kw)
    res = minimize(fun, M.g_inv(np.asarray(init_params, dtype=np.float64)), jac=True, method='L-BFGS-B',
                   options=dict(maxiter=maxiter))
    return M.g(res.x), res


def _value_and_grad_many(method, build, thetas, yss, Xi, dt, sgps, rel_step, build_kw, record_index=None, exact=None):
    """NLL and gradient of R records (the rows ``record_index`` of yss; all of them by default) at R parameter vectors: exact (the tangent
    kernel, R P lanes) for the discrete EKF on the chirp / La Scala models, else central differences -- ONE launch of R (2 P + 1)
    trials, each record read in place by its 2 P + 1 probes."""
    R, P = thetas.shape
    if exact is None:
        exact = _exact_by_default(method, build, Xi, R, build_kw)
    if exact:
        if not has_exact_gradient(method, build, Xi) or build_kw:
            raise ValueError('exact=True: the tangent kernel is built for the discrete EKF on build_chirp_model / build_lascala_model with a scalar Xi')
        f, grad = value_and_grad(build, thetas, yss, Xi, dt, record_index=record_index)
        f = f.copy()
        f[~np.isfinite(f)] = np.inf
        return f, np.where(np.isfinite(grad), grad, 0.0)
    h = rel_step * (1.0 + np.abs(thetas))                                   # (R, P)
    batch = np.repeat(thetas[:, None, :], 2 * P + 1, axis=1)                # (R, 2P+1, P)
    idx = np.arange(P)
    batch[:, 1 + 2 * idx, idx] += h
    batch[:, 2 + 2 * idx, idx] -= h
    nll = batched_nll(method, build, batch.reshape(-1, P), yss, Xi, dt, sgps, record_index=record_index, **build_kw).reshape(R, 2 * P + 1)
    f = nll[:, 0].copy()
    grad = (nll[:, 1::2] - nll[:, 2::2]) / (2 * h)
    f[~np.isfinite(f)] = np.inf
    return f, np.where(np.isfinite(grad), grad, 0.0)


def fit_many(method, build, init_params, yss, Xi, dt, sgps=None, maxiter=200, history=10, gtol=1e-5, ftol=2.2e-9,
             rel_step=1e-6, exact=None, **build_kw):
    """Maximum likelihood for R measurement records in lock step: limited-memory BFGS with a backtracking (Armijo) line
    search, every record with its own iterate, history and step length, and every probe of every record evaluated in
    the SAME kernel launch (R x 13 trials for the chirp model).  A launch costs T x 0.35 us whatever the batch up to
    ~4000 trials, so R records take the wall time of one -- the reference's Monte-Carlo jobs (tetralith/jobs/*_mle.py)
    run one L-BFGS-B per record.  Same objective, same unconstrained parametrisation g() as :func:`fit`.

    yss (R, T);  init_params (P,) or (R, P) positive model parameters  ->  (opt_params (R, P), info dict)."""
    from chirpgp_amd import _engine as E
    yss = E.dev(yss)                                  # uploaded once; every probe of every line search reads it in place
    if yss.ndim == 1:
        yss = yss[None, :]
    R = yss.shape[0]
    x = np.array(np.broadcast_to(M.g_inv(np.asarray(init_params, dtype=np.float64)), (R, np.shape(init_params)[-1])))
    P = x.shape[1]
    f, g = _value_and_grad_many(method, build, x, yss, Xi, dt, sgps, rel_step, build_kw, exact=exact)
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
            ft, gt = _value_and_grad_many(method, build, xt, yss, Xi, dt, sgps, rel_step, build_kw, record_index=idx, exact=exact)
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


def grid_search(method, build, grid, yss, Xi, dt, sgps=None, **build_kw):
    """Parameter-grid MLE sweep (BASELINE config C5: "batch x param-grid MLE sweep"): the final NLL of ``method`` at every
    grid point for every record, in ONE launch of R G trials -- G parameter vectors per record, each record read in place.

    grid (G, P) positive model parameters; yss (R, T) or (T,)  ->  (best (R, P), nll (R, G), argmin (R,)).
    A diverged grid point (NaN / inf NLL) never wins; a record whose every grid point diverges gets NaN parameters."""
    grid = np.atleast_2d(np.asarray(grid, dtype=np.float64))
    G = grid.shape[0]
    R = 1 if np.ndim(yss) == 1 else int(np.shape(yss)[0])
    thetas = np.tile(M.g_inv(grid), (R, 1))                     # record-major: the G rows of a record are consecutive trials
    nll = batched_nll(method, build, thetas, yss, Xi, dt, sgps, **build_kw).reshape(R, G)
    masked = np.where(np.isfinite(nll), nll, np.inf)
    arg = np.argmin(masked, axis=1)
    best = M.g(M.g_inv(grid))[arg]                              # the parameters the filter actually ran with
    best[~np.isfinite(masked[np.arange(R), arg])] = np.nan
    return best, nll, arg
