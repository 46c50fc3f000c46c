"""The five-step script every driver of the reference runs (demos/ekfs_mle.py:39-90, demos/ghfs_mle.py:34-90,
demos/cd_ghfs_mle.py:28-80, demos/ghfs_harmonics_mle.py:27-80, tetralith/jobs/*_mle.py), on the MI355X engine:

    MLE of the model parameters through the filter  ->  filter  ->  smoother  ->  E[g(V)]  ->  RMSE  (-> .npz result file)

Shared by the demo scripts of this directory; the plots of the reference's demos are not reproduced."""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs, mle, results                       # noqa: E402
from chirpgp_amd.models import (g, build_chirp_model, build_harmonic_chirp_model, build_lascala_model,     # noqa: E402
                                build_kpt_chirp_model)
from chirpgp_amd.quadratures import gaussian_expectation                            # noqa: E402
from chirpgp_amd.toymodels import (gen_chirp, gen_harmonic_chirp, meow_freq, constant_mag, damped_exp_mag,   # noqa: E402
                                   random_ou_mag)
from chirpgp_amd.tools import rmse                                                  # noqa: E402

INIT_PARAMS = [0.1, 0.1, 0.1, 1., 1., 7.]       # lam, b, delta, ell, sigma, m0_1 (demos/ekfs_mle.py:39)
FILTER_OF = {'ekfs': 'ekf', 'ghfs': 'sgp_filter', 'cd_ekfs': 'cd_ekf', 'cd_ghfs': 'cd_sgp_filter', 'kpt': 'ekf_for_kpt'}
# model families of the reference's jobs: builder, MLE start point (positive parameters)
FAMILIES = {
    'chirp': (build_chirp_model, INIT_PARAMS),                          # tetralith/jobs/ekfs_mle.py:37
    'harmonic': (build_harmonic_chirp_model, INIT_PARAMS),              # tetralith/jobs/harmonic_ekfs_mle.py:40
    'lascala': (build_lascala_model, [0.1, 1., 1., 7.]),                # tetralith/jobs/lascala_ekfs_mle.py:37: delta, ell, sigma, m0_1
    'kpt': (build_kpt_chirp_model, [0.02, 1e-5, 1e-5, 8., 1.]),         # tetralith/jobs/kpt_mle.py:38: q1, q2, p0, f0, a0
}


def magnitudes(rng):
    """The three magnitude laws of the reference's demos (constant, damped, one Ornstein-Uhlenbeck realisation)."""
    return (('const', constant_mag(1.)), ('damped', damped_exp_mag(0.3)), ('ou', random_ou_mag(1., 1., rng)))


def _family_setup(method, family, num_harmonics, dt, init):
    if family is None:
        family = 'kpt' if method == 'kpt' else ('harmonic' if num_harmonics else 'chirp')
    build, init_default = FAMILIES[family]
    build_kw = {}
    if family == 'harmonic':
        build_kw = dict(num_harmonics=num_harmonics)
    elif family == 'kpt':
        build_kw = dict(fs=1. / dt, num_harmonics=max(num_harmonics, 1))
    return build, (init_default if init is None else init), build_kw


def _filter_and_smooth(method, build, build_kw, params, sgps, Xi, dt, ys, keep_rows=True):
    """Filter + smoother + E[g(V)] at given model parameters; `params` and `ys` may both carry a leading record axis.
    The estimate E[g(V)] of the frequency marginal (demos/ekfs_mle.py:69-77: gaussian_expectation on mss[:, 2], sqrt(Pss[:, 2, 2])) is
    written by the smoother launch itself (`select`: include/chirpgp_hip.h, cgp_smoother_select); with ``keep_rows=False`` the full
    smoothing rows are not written at all (None in their place) -- what a caller that only wants the RMSE needs."""
    if method == 'kpt':      # tetralith/jobs/kpt_mle.py:54-76: ekf_for_kpt, then the LINEAR smoother; frequency = g(x_0) fs / 2 pi
        F, Sigma, m0, P0, h = build(params, **build_kw)
        mfs, Pfs, _ = fs.ekf_for_kpt(F, Sigma, h, Xi, m0, P0, dt, ys)
        mss, Pss, sel = fs.rts(F, Sigma, mfs, Pfs, want=(keep_rows, keep_rows), select=dict(comp=0, mean=True, var=True))
        scale = 1. / dt / 2 / math.pi          # (the marginal is rescaled before the expectation: that one stays on the host)
        ms, sd = sel['mean'] * scale, np.sqrt(sel['var']) * scale
        est = gaussian_expectation(ms=ms.reshape(-1), chol_Ps=sd.reshape(-1), func=g, force_shape=True)[:, 0].reshape(ms.shape)
        return mss, Pss, est
    drift, dispersion, m_and_cov, m0, P0, H = build(params, **build_kw)
    # the frequency state is the last-but-one component (index 2 of the chirp model, -2 of the harmonic one)
    kw = dict(want=(keep_rows, keep_rows), select=dict(comp=-2, expect='softplus'))
    if method == 'ekfs':
        mfs, Pfs, _ = fs.ekf(m_and_cov, H, Xi, m0, P0, dt, ys)
        mss, Pss, sel = fs.eks(m_and_cov, mfs, Pfs, dt, **kw)
    elif method == 'ghfs':
        mfs, Pfs, _ = fs.sgp_filter(m_and_cov, sgps, H, Xi, m0, P0, dt, ys)
        mss, Pss, sel = fs.sgp_smoother(m_and_cov, sgps, mfs, Pfs, dt, **kw)
    elif method == 'cd_ekfs':
        mfs, Pfs, _ = fs.cd_ekf(drift, dispersion, H, Xi, m0, P0, dt, ys)
        mss, Pss, sel = fs.cd_eks(drift, dispersion, mfs, Pfs, dt, **kw)
    else:   # the reference passes the dispersion MATRIX: dispersion(jnp.eye(4)) (demos/cd_ghfs_mle.py:48)
        mfs, Pfs, _ = fs.cd_sgp_filter(drift, dispersion(None), sgps, H, Xi, m0, P0, dt, ys)
        mss, Pss, sel = fs.cd_sgp_smoother(drift, dispersion(None), sgps, mfs, Pfs, dt, **kw)
    return mss, Pss, sel['expect']


def run_records(method, yss, Xi, dt, sgps=None, num_harmonics=0, maxiter=200, init=None, family=None, keep_rows=True):
    """The pipeline for R records in LOCK STEP: one maximum-likelihood fit for all of them (chirpgp_amd.mle.fit_many: every
    line-search probe of every record in one kernel launch), then ONE batched filter and ONE batched smoother launch with a
    parameter vector per record -- R records for about the wall time of one (the reference's jobs loop over them,
    tetralith/jobs/ekfs_mle.py:26-81).  yss (R, T) -> dict(opt_params (R, P), fun (R,), nll0 (R,), mss, Pss, est_freq (R, T))."""
    build, init, build_kw = _family_setup(method, family, num_harmonics, dt, init)
    filt = FILTER_OF[method]
    yss = np.asarray(yss, dtype=np.float64)
    R = yss.shape[0]
    theta0 = np.tile(np.log(np.expm1(np.asarray(init, dtype=np.float64))), (R, 1))
    nll0 = mle.batched_nll(filt, build, theta0, yss, Xi, dt, sgps, **build_kw)
    params, info = mle.fit_many(filt, build, init, yss, Xi, dt, sgps=sgps, maxiter=maxiter, **build_kw)
    mss, Pss, est = _filter_and_smooth(method, build, build_kw, params, sgps, Xi, dt, yss, keep_rows)
    return dict(opt_params=params, fun=info['fun'], nll0=nll0, mss=mss, Pss=Pss, est_freq=est, nit=info['nit'], launches=info['launches'])


def run_record(method, ys, Xi, dt, sgps=None, num_harmonics=0, maxiter=200, init=None, family=None, keep_rows=True):
    """MLE -> filter -> smoother -> E[g(V)] on one measurement record.
    method: 'ekfs' | 'ghfs' | 'cd_ekfs' | 'cd_ghfs' | 'kpt';  family: 'chirp' | 'harmonic' | 'lascala' | 'kpt'
    (default: 'kpt' for method 'kpt', else 'harmonic' when num_harmonics > 0, else 'chirp').
    Returns a dict with opt_params, the scipy result, nll0 (objective at the start), the smoothing results and est_freq."""
    build, init, build_kw = _family_setup(method, family, num_harmonics, dt, init)
    filt = FILTER_OF[method]
    nll0 = float(mle.batched_nll(filt, build, np.log(np.expm1(np.asarray(init, dtype=np.float64))), ys, Xi, dt, sgps, **build_kw)[0])
    opt_params, res = mle.fit(filt, build, init, ys, Xi, dt, sgps=sgps, maxiter=maxiter, **build_kw)
    mss, Pss, est = _filter_and_smooth(method, build, build_kw, opt_params, sgps, Xi, dt, ys, keep_rows)
    return dict(opt_params=opt_params, res=res, nll0=nll0, mss=mss, Pss=Pss, est_freq=est)


def records_of_run(seed, T, dt, Xi, signal_harmonics, mags=None):
    """The measurement records of ONE Monte-Carlo run, one per magnitude law: (position, name, ys) in the order the random
    stream of numpy.random.default_rng(seed) produces them (the Ornstein-Uhlenbeck magnitude draws from the same stream)."""
    ts = np.linspace(dt, dt * T, T)
    rng = np.random.default_rng(seed)
    _, true_phase_func = meow_freq(offset=8.)
    for k, (name, mag) in enumerate(magnitudes(rng)):
        if mags is not None and name not in mags:
            continue
        clean = (gen_chirp(ts, mag, true_phase_func) if signal_harmonics == 0
                 else gen_harmonic_chirp(ts, [mag] * signal_harmonics, true_phase_func))
        yield k, name, clean + math.sqrt(Xi) * rng.standard_normal(T)


def demo(method, sgps=None, num_harmonics=0, T=3141, seed=555, Xi=0.1, dt=0.001, maxiter=200, save_dir=None, mags=None, quiet=False,
         family=None, signal_harmonics=None, result_name=None, mc=None):
    """One run per magnitude law, as the reference's demo scripts do; returns [(name, rmse, nll0, nll_opt), ...].
    ``signal_harmonics``: harmonics of the SIGNAL (default: those of the model); ``result_name``: file stem of the saved
    results (the reference's jobs: 'kpt_mle', 'lascala_ekfs_mle', ...; default: the method name), numbered ``mc`` (default: the
    position of the magnitude law, as the demos have a single run)."""
    sig_h = num_harmonics if signal_harmonics is None else signal_harmonics
    ts = np.linspace(dt, dt * T, T)
    true_freq_func, _ = meow_freq(offset=8.)
    out = []
    for k, name, ys in records_of_run(seed, T, dt, Xi, sig_h, mags):
        t0 = time.time()
        r = run_record(method, ys, Xi, dt, sgps=sgps, num_harmonics=num_harmonics, maxiter=maxiter, family=family, keep_rows=bool(save_dir))
        err = float(rmse(true_freq_func(ts), r['est_freq'])) if r['res'].success or np.isfinite(r['res'].fun) else float('nan')
        if save_dir:       # tetralith/jobs/ekfs_mle.py:75-81: NaN results for a diverged run
            results.save_result(save_dir, result_name or method, name, k if mc is None else mc, r['mss'], r['Pss'], err)
        if not quiet:
            print(f'{method:8s} {name:7s} params {np.array2string(r["opt_params"], precision=3)}  nll {r["nll0"]:.2f} -> {r["res"].fun:.2f}  '
                  f'iters {r["res"].nit} ({r["res"].nfev} launches)  RMSE {err:.3f} Hz  [{time.time() - t0:.2f} s]')
        out.append((name, err, r['nll0'], float(r['res'].fun)))
    return out
