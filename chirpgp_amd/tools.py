"""Small host-side helpers around the hot path (chirpgp/tools.py): rmse and the Van Loan discretisation."""
import numpy as np
import scipy.linalg

__all__ = ['rmse', 'lti_sde_to_disc']


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
