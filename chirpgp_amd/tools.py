"""Helpers around the hot path (chirpgp/tools.py): rmse, the Van Loan discretisation (host side) and the Monte-Carlo
simulators, which run on the device (include/chirpgp_hip.h: cgp_simulate).

The simulators take an integer ``key`` (seed) where the reference takes a jax.random key: jax's streams cannot be
reproduced without JAX, so the draws differ from the reference's while the simulated law is the same.  ``batch`` /
``trial0`` are additions: ``batch = B`` returns B independent trajectories (the reference's ``jax.vmap`` over split
keys, tetralith/jobs/crlb_ekf.py:39-66) as CUDA tensors; trial ``trial0 + i`` of any call draws the same numbers,
whatever the batch or the number of ranks."""
import numpy as np
import scipy.linalg

__all__ = ['rmse', 'lti_sde_to_disc', 'simulate_lgssm', 'simulate_sde', 'simulate_sde_init', 'simulate_measurements']


def rmse(x1, x2, reduce_sum=True):
    """Root mean square error over axis 0 (tools.py:279-293)."""
    val = np.sqrt(np.mean((np.asarray(x1) - np.asarray(x2)) ** 2, axis=0))
    return np.sum(val) if reduce_sum else val


def lti_sde_to_disc(A, B, dt):
    """dX = A X dt + B dW  ->  X_k = F X_{k-1} + N(0, Sigma)  (tools.py:44-78)."""
    A, B = np.atleast_2d(np.asarray(A, dtype=np.float64)), np.asarray(B, dtype=np.float64)
    d = A.shape[0]
    BBt = np.outer(B, B) if B.ndim == 1 else B @ B.T
    F = scipy.linalg.expm(A * dt)
    phi = np.block([[A, BBt], [np.zeros_like(A), -A.T]])
    AB = scipy.linalg.expm(phi * dt) @ np.vstack([np.zeros_like(A), np.eye(d)])
    return F, AB[:d] @ F.T


def _sim(spec, H, Xi, m0, P0, dt, T, key, batch, trial0, want):
    from chirpgp_amd import _engine as E
    from chirpgp_amd import models as M
    if hasattr(spec, 'at') and not isinstance(spec, M.DiscreteModel):
        spec = spec.at(dt)
    if not isinstance(spec, M.DiscreteModel):
        raise TypeError('m_and_cov must be a chirpgp_amd.models discrete descriptor (e.g. disc_chirp_lcd(...), '
                        'linear_cond_m_cov(F, Sigma)); arbitrary Python callables cannot run inside the HIP kernels')
    B = 1 if batch is None else int(batch)
    xs, ys = E.run_simulate(spec, H, Xi, m0, P0, dt, T, key, B, trial0=trial0, want=want)
    if batch is None:
        return (None if xs is None else xs[0].cpu().numpy()), (None if ys is None else ys[0].cpu().numpy())
    return xs, ys


def simulate_lgssm(F, Sigma, x0, T, key, batch=None, trial0=0):
    """x_k = F x_{k-1} + N(0, Sigma) from the given x0 (tools.py:81-116) -> (T, d), or (B, T, d) on the device."""
    from chirpgp_amd import models as M
    return _sim(M.linear_cond_m_cov(np.asarray(F, dtype=np.float64), np.asarray(Sigma, dtype=np.float64)), None, None,
                x0, None, 0., T, key, batch, trial0, (True, False))[0]


def simulate_sde(m_and_cov, m0, P0, dt, T, key, const_diag_cov=False, batch=None, trial0=0):
    """x0 ~ N(m0, P0), x_k = m(x_{k-1}) + chol(cov) dw_k (tools.py:119-170) -> (T, d), or (B, T, d) on the device.
    ``const_diag_cov`` is accepted for signature parity: the Cholesky factor of a diagonal matrix is its square root."""
    return _sim(m_and_cov, None, None, m0, P0, dt, T, key, batch, trial0, (True, False))[0]


def simulate_sde_init(m_and_cov, x0, dt, T, key, const_diag_cov=False, batch=None, trial0=0):
    """simulate_sde from a given x0 (tools.py:173-194)."""
    return _sim(m_and_cov, None, None, x0, None, dt, T, key, batch, trial0, (True, False))[0]


def simulate_measurements(m_and_cov, H, Xi, m0, P0, dt, T, key, batch=None, trial0=0, states=True):
    """States and scalar measurements y_k = H . x_k + sqrt(Xi) e_k, the Monte-Carlo input of tetralith/jobs/crlb_ekf.py:39-66
    -> (xs, ys); ``states=False`` skips the (B, T, d) state output."""
    return _sim(m_and_cov, H, Xi, m0, P0, dt, T, key, batch, trial0, (bool(states), True))
