"""Filters and smoothers of chirpgp/filters_smoothers.py, executed by the MI355X engine.

Same names, positional argument order and return tuples as the reference (``__all__`` below is the reference's,
filters_smoothers.py:26-36).  Differences a caller sees:

* model callables are the descriptor objects of :mod:`chirpgp_amd.models` (``cond_m_cov``, ``a``, ``b``, ``h``);
  a plain Python closure raises ``TypeError`` -- it cannot be run inside a HIP kernel and there is no CPU fallback;
* ``ys`` (filters) and ``mfs, Pfs`` (smoothers) may carry a leading batch axis -- the reference's
  ``jax.vmap(..., in_axes=0)`` (tetralith/jobs/crlb_ekf.py:68-72); so may ``m0, P0, H, Xi`` and the model parameters;
* NumPy in -> NumPy out, torch CUDA tensors in -> torch CUDA tensors out (results stay in HBM).
"""
import numpy as np

from chirpgp_amd import _engine as E
from chirpgp_amd import models as M
from chirpgp_amd.quadratures import SigmaPoints      # noqa: F401 (re-exported, as filters_smoothers.py:22 imports it)

__all__ = ['kf', 'rts', 'ekf', 'ekf_for_kpt', 'eks', 'cd_ekf', 'cd_eks',
           'sgp_filter', 'sgp_smoother', 'cd_sgp_filter', 'cd_sgp_smoother']


def _discrete(cond_m_cov, dt=None):
    if isinstance(cond_m_cov, (M.DiscreteModel, M.CustomDiscrete)):
        return cond_m_cov
    if hasattr(cond_m_cov, 'at') and dt is not None:          # dt-dependent linear descriptors (disc_m32)
        return cond_m_cov.at(dt)
    raise TypeError('cond_m_cov must be a chirpgp_amd.models descriptor (e.g. disc_chirp_lcd(...), linear_cond_m_cov(F, Sigma), or '
                    'custom_cond_m_cov(source, ...) for a model of your own); arbitrary Python callables cannot run inside the HIP kernels')


def _drift(a):
    if isinstance(a, (M.DriftModel, M.CustomDrift)):
        return a
    raise TypeError('the drift must be a chirpgp_amd.models descriptor (e.g. model_chirp(...)[0], linear_sde(A, B)[0])')


def _gamma_from_callable(b):
    if isinstance(b, M.Dispersion):
        return b.outer()
    raise TypeError('the dispersion must be a chirpgp_amd.models.Dispersion (e.g. model_chirp(...)[1], linear_sde(A, B)[1])')


def _gamma_from_matrix(b):
    if isinstance(b, M.Dispersion):
        return b.outer()
    b = _np(b)                                      # a (d, dw) constant: formed on the host, no torch op on the path
    return b @ np.swapaxes(b, -1, -2)


def _sgps(sgps, d):
    if not isinstance(sgps, tuple) or not hasattr(sgps, 'xi') or not hasattr(sgps, 'w'):
        raise TypeError('sgps must be a SigmaPoints instance')
    if int(sgps.d) != d:
        raise ValueError(f'sigma points are for d = {sgps.d}, the model has d = {d}')
    return sgps


def kf(F, Sigma, H, Xi, m0, P0, ys, **kw):
    """Kalman filter for a scalar measurement (filters_smoothers.py:145-184) -> (mfs, Pfs, cumulative nll)."""
    return E.run_filter(E.F_EKF, M.linear_cond_m_cov(_np(F), _np(Sigma)), None, None, H, Xi, m0, P0, 0., ys, **kw)


def rts(F, Sigma, mfs, Pfs, **kw):
    """RTS smoother (filters_smoothers.py:187-219) -> (mss, Pss)."""
    return E.run_smoother(E.S_EKS, M.linear_cond_m_cov(_np(F), _np(Sigma)), None, None, 0., mfs, Pfs, **kw)


def ekf(cond_m_cov, H, Xi, m0, P0, dt, ys, **kw):
    """Extended Kalman filter (filters_smoothers.py:222-264)."""
    spec = _discrete(cond_m_cov, dt)
    if isinstance(spec, M.CustomDiscrete):            # a model compiled at run time: the generic kernel on the caller's source
        return E.run_filter_custom(spec, None, H, Xi, m0, P0, dt, ys, **_custom_kw(kw))
    return E.run_filter(E.F_EKF, _discrete(cond_m_cov, dt), None, None, H, Xi, m0, P0, dt, ys, **kw)


def ekf_for_kpt(F, Sigma, h, Xi, m0, P0, dt, ys, **kw):
    """Ad-hoc EKF of the KPT model: linear dynamics, harmonic measurement h (filters_smoothers.py:267-314)."""
    if isinstance(h, M.CustomMeasurement):            # a measurement function compiled at run time over the linear dynamics
        spec = h.with_dynamics(_np(F), _np(Sigma))
        return E.run_filter_custom(spec, None, spec.q, Xi, m0, P0, dt, ys, **_custom_kw(kw))
    if not isinstance(h, M.MeasurementKPT):
        raise TypeError('h must be the MeasurementKPT returned by chirpgp_amd.models.build_kpt_chirp_model, or a custom_measurement(source, d)')
    spec = M.linear_cond_m_cov(_np(F), _np(Sigma))
    if spec.d != h.n_harm + 2:
        raise ValueError('F must be (n_harm + 2) x (n_harm + 2)')
    spec.model_id, spec.n_harm = M.M_KPT, h.n_harm
    return E.run_filter(E.F_EKF_KPT, spec, None, None, None, Xi, m0, P0, dt, ys, **kw)


def eks(cond_m_cov, mfs, Pfs, dt, **kw):
    """Extended Kalman smoother (filters_smoothers.py:317-349)."""
    spec = _discrete(cond_m_cov, dt)
    if isinstance(spec, M.CustomDiscrete):
        return E.run_smoother_custom(spec, None, dt, mfs, Pfs, **_custom_kw(kw, ('flags',)))
    return E.run_smoother(E.S_EKS, _discrete(cond_m_cov, dt), None, None, dt, mfs, Pfs, **kw)


def cd_ekf(a, b, H, Xi, m0, P0, dt, ys, **kw):
    """Continuous-discrete EKF with RK4 moment integration (filters_smoothers.py:352-397)."""
    if isinstance(a, M.CustomDrift):
        return E.run_filter_custom(a, _gamma_from_callable(b), H, Xi, m0, P0, dt, ys, **_custom_kw(kw))
    return E.run_filter(E.F_CD_EKF, _drift(a), None, _gamma_from_callable(b), H, Xi, m0, P0, dt, ys, **kw)


def cd_eks(a, b, mfs, Pfs, dt, **kw):
    """Continuous-discrete EKS (filters_smoothers.py:400-443)."""
    if isinstance(a, M.CustomDrift):
        return E.run_smoother_custom(a, _gamma_from_callable(b), dt, mfs, Pfs, **_custom_kw(kw, ('flags',)))
    return E.run_smoother(E.S_CD_EKS, _drift(a), None, _gamma_from_callable(b), dt, mfs, Pfs, **kw)


def sgp_filter(cond_m_cov, sgps, H, Xi, m0, P0, dt, ys, **kw):
    """Sigma-point (Gauss-Hermite / cubature) filter on a discretised model (filters_smoothers.py:446-490)."""
    spec = _discrete(cond_m_cov, dt)
    if isinstance(spec, M.CustomDiscrete):            # a model compiled at run time: the literal fan, covariance at every point
        return E.run_filter_custom(spec, None, H, Xi, m0, P0, dt, ys, sgps=_sgps(sgps, spec.d), **_custom_kw(kw))
    return E.run_filter(E.F_SGP, spec, _sgps(sgps, spec.d), None, H, Xi, m0, P0, dt, ys, **kw)


def sgp_smoother(cond_m_cov, sgps, mfs, Pfs, dt, **kw):
    """Sigma-point smoother (filters_smoothers.py:493-531)."""
    spec = _discrete(cond_m_cov, dt)
    if isinstance(spec, M.CustomDiscrete):
        return E.run_smoother_custom(spec, None, dt, mfs, Pfs, sgps=_sgps(sgps, spec.d), **_custom_kw(kw, ('flags',)))
    return E.run_smoother(E.S_SGP, spec, _sgps(sgps, spec.d), None, dt, mfs, Pfs, **kw)


def cd_sgp_filter(a, b, sgps, H, Xi, m0, P0, dt, ys, **kw):
    """Continuous-discrete sigma-point filter; b is the constant (d, dw) dispersion matrix (filters_smoothers.py:534-582)."""
    spec = _drift(a)
    if isinstance(spec, M.CustomDrift):
        return E.run_filter_custom(spec, _gamma_from_matrix(b), H, Xi, m0, P0, dt, ys, sgps=_sgps(sgps, spec.d), **_custom_kw(kw))
    return E.run_filter(E.F_CD_SGP, spec, _sgps(sgps, spec.d), _gamma_from_matrix(b), H, Xi, m0, P0, dt, ys, **kw)


def cd_sgp_smoother(a, b, sgps, mfs, Pfs, dt, **kw):
    """Continuous-discrete sigma-point smoother (filters_smoothers.py:585-632)."""
    spec = _drift(a)
    if isinstance(spec, M.CustomDrift):
        return E.run_smoother_custom(spec, _gamma_from_matrix(b), dt, mfs, Pfs, sgps=_sgps(sgps, spec.d), **_custom_kw(kw, ('flags',)))
    return E.run_smoother(E.S_CD_SGP, spec, _sgps(sgps, spec.d), _gamma_from_matrix(b), dt, mfs, Pfs, **kw)


def _custom_kw(kw, allowed=('nll_final_only', 'want', 'flags')):
    """Keywords a runtime-compiled model understands (one launch shape, dense records); anything else is refused by name."""
    extra = set(kw) - set(allowed)
    if extra:
        raise TypeError(f'custom models take {" / ".join(allowed)} only, not {sorted(extra)}')
    return kw


def _np(x):
    return x.detach().cpu().numpy() if E._is_torch(x) else np.asarray(x, dtype=np.float64)
