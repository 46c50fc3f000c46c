"""NumPy float64 restatement of the model functions the scan bodies evaluate.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/chirpgp/models.py;
every callable accepts complex arrays so that np_filters.jacobian() can
differentiate it by the complex-step method (the restatement of jax.jacfwd).
"""
import math
import numpy as np

__all__ = ['g', 'g_inv', 'm32_solution', 'stationary_cov_m32', 'blkdiag',
           'model_chirp', 'model_harmonic_chirp', 'model_lascala',
           'disc_chirp_lcd', 'disc_chirp_lcd_cond_v', 'disc_harmonic_chirp_lcd',
           'disc_model_lascala_lcd', 'disc_m32',
           'build_chirp_model', 'build_harmonic_chirp_model', 'build_lascala_model',
           'build_kpt_chirp_model']


def g(x):
    """softplus, naive form exactly as models.py:50 (overflows to inf for x > ~709)."""
    with np.errstate(over='ignore'):
        return np.log(np.exp(x) + 1.)


def g_inv(x):
    """models.py:53."""
    return np.log(np.exp(x) - 1.)


def blkdiag(*blocks):
    """Dense block-diagonal of scalars / 2-D blocks (jax.scipy.linalg.block_diag)."""
    blocks = [np.atleast_2d(b) for b in blocks]
    n = sum(b.shape[0] for b in blocks)
    dtype = np.result_type(*[b.dtype for b in blocks], np.float64)
    out = np.zeros((n, n), dtype=dtype)
    k = 0
    for b in blocks:
        r = b.shape[0]
        out[k:k + r, k:k + r] = b
        k += r
    return out


def stationary_cov_m32(ell, sigma):
    """models.py:56-58."""
    return np.array([[sigma ** 2, 0.],
                     [0., (math.sqrt(3) / ell) ** 2 * sigma ** 2]])


def m32_solution(ell, sigma, dt):
    """Closed-form Matern-3/2 transition matrix and covariance, models.py:61-73."""
    gamma = math.sqrt(3) / ell
    eta = dt * gamma
    beta = sigma ** 2 * np.exp(-2 * eta)
    transition = np.array([[1 + eta, dt],
                           [-dt * gamma ** 2, 1 - eta]]) * np.exp(-eta)
    off = 2 * dt ** 2 * gamma ** 3 * beta
    Sigma = np.array([[sigma ** 2 - beta * (2 * eta + 2 * eta ** 2 + 1), off],
                      [off, gamma ** 2 * (sigma ** 2 + beta * (2 * eta - 2 * eta ** 2 - 1))]])
    return transition, Sigma


def _rot(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s], [s, c]])


# --------------------------------------------------------------------------- SDE models
def model_chirp(lam, b, ell, sigma, delta):
    """models.py:76-119 -> (drift, dispersion, m0, P0, H)."""
    gamma = math.sqrt(3) / ell

    def drift(u):
        w = 2 * math.pi * g(u[2])
        A = np.array([[-lam, -w, 0., 0.],
                      [w, -lam, 0., 0.],
                      [0., 0., 0., 1.],
                      [0., 0., -(gamma ** 2), -2 * gamma]])
        return A @ u

    def dispersion(_):
        return np.diag(np.array([b, b, 0., 2 * sigma * (math.sqrt(3) / ell) ** 1.5]))

    m0 = np.array([0., 1., 0., 0.])
    P0 = blkdiag(delta, delta, stationary_cov_m32(ell, sigma))
    H = np.array([0., 1., 0., 0.])
    return drift, dispersion, m0, P0, H


def model_harmonic_chirp(lam, b, ell, sigma, delta, num_harmonics=1, freq_scale=1.):
    """models.py:122-178."""
    gamma = math.sqrt(3) / ell
    m32_drift = np.array([[0., 1.], [-(gamma ** 2), -2 * gamma]])

    def drift(u):
        w = 2 * math.pi * g(u[-2]) * freq_scale
        blocks = [np.array([[-lam, -w * k], [w * k, -lam]]) for k in range(1, num_harmonics + 1)]
        return blkdiag(*blocks, m32_drift) @ u

    def dispersion(_):
        return np.diag(np.array([b, b] * num_harmonics + [0., 2 * sigma * (math.sqrt(3) / ell) ** 1.5]))

    m0 = np.array([0., 1.] * num_harmonics + [0., 0.])
    P0 = blkdiag(*([delta, delta] * num_harmonics), stationary_cov_m32(ell, sigma))
    H = np.array([0., 1.] * num_harmonics + [0., 0.])
    return drift, dispersion, m0, P0, H


def model_lascala(ell, sigma, delta):
    """models.py:181-261 (chirp model without damping and chirp dispersion)."""
    gamma = math.sqrt(3) / ell

    def drift(u):
        w = 2 * math.pi * g(u[2])
        A = np.array([[0., -w, 0., 0.],
                      [w, 0., 0., 0.],
                      [0., 0., 0., 1.],
                      [0., 0., -(gamma ** 2), -2 * gamma]])
        return A @ u

    def dispersion(_):
        return np.diag(np.array([0., 0., 0., 2 * sigma * (math.sqrt(3) / ell) ** 1.5]))

    m0 = np.array([0., 1., 0., 0.])
    P0 = blkdiag(delta, delta, stationary_cov_m32(ell, sigma))
    H = np.array([0., 1., 0., 0.])
    return drift, dispersion, m0, P0, H


# --------------------------------------------------------------------------- discretisations
def _chirp_noise_var(lam, b, dt):
    """models.py:302-308 -- the lax.cond on lam == 0."""
    if lam == 0.:
        return b ** 2 * dt
    return b ** 2 / (2 * lam) * (1 - np.exp(-2 * lam * dt))


def disc_chirp_lcd(lam, b, ell, sigma):
    """Locally conditional discretisation, models.py:264-311."""

    def m_and_cov(u, dt):
        w = 2 * math.pi * g(u[2])
        blk_harmonic = _rot(dt * w) * np.exp(-lam * dt)
        blk_m32_m, blk_m32_Sigma = m32_solution(ell, sigma, dt)
        cond_m = blkdiag(blk_harmonic, blk_m32_m) @ u
        q = _chirp_noise_var(lam, b, dt)
        return cond_m, blkdiag(q, q, blk_m32_Sigma)

    return m_and_cov


def disc_chirp_lcd_cond_v(lam, b):
    """models.py:314-329."""

    def m_and_cov(u, v, dt):
        w = 2 * math.pi * g(v)
        cond_m = (_rot(dt * w) * np.exp(-lam * dt)) @ u
        return cond_m, np.eye(2) * _chirp_noise_var(lam, b, dt)

    return m_and_cov


def disc_harmonic_chirp_lcd(lam, b, ell, sigma, num_harmonics=1, freq_scale=1.):
    """models.py:332-386."""

    def m_and_cov(u, dt):
        w = 2 * math.pi * g(u[-2]) * freq_scale
        blocks = [_rot(dt * k * w) * np.exp(-lam * dt) for k in range(1, num_harmonics + 1)]
        blk_m32_m, blk_m32_Sigma = m32_solution(ell, sigma, dt)
        cond_m = blkdiag(*blocks, blk_m32_m) @ u
        q = _chirp_noise_var(lam, b, dt)
        return cond_m, blkdiag(*([q] * (2 * num_harmonics)), blk_m32_Sigma)

    return m_and_cov


def disc_m32(ell, sigma):
    """models.py:408-416."""

    def m_and_cov(u, dt):
        transition, Sigma = m32_solution(ell, sigma, dt)
        return transition @ u, Sigma

    return m_and_cov


def disc_model_lascala_lcd(ell, sigma):
    """models.py:419-434."""

    def m_and_cov(u, dt):
        w = 2 * math.pi * g(u[2])
        blk_m32_m, blk_m32_Sigma = m32_solution(ell, sigma, dt)
        cond_m = blkdiag(_rot(dt * w), blk_m32_m) @ u
        return cond_m, blkdiag(0., 0., blk_m32_Sigma)

    return m_and_cov


# --------------------------------------------------------------------------- builders
def build_chirp_model(params):
    """models.py:437-459.  params = lam, b, delta, ell, sigma, m0_v."""
    lam, b, delta, ell, sigma, m0_v = [float(p) for p in params]
    drift, dispersion, _, P0, H = model_chirp(lam, b, ell, sigma, delta)
    m0 = np.array([0., 0., m0_v, 0.])
    return drift, dispersion, disc_chirp_lcd(lam, b, ell, sigma), m0, P0, H


def build_harmonic_chirp_model(params, num_harmonics=1, freq_scale=1.):
    """models.py:462-494."""
    lam, b, delta, ell, sigma, m0_v = [float(p) for p in params]
    drift, dispersion, _, P0, H = model_harmonic_chirp(lam, b, ell, sigma, delta,
                                                       num_harmonics=num_harmonics, freq_scale=freq_scale)
    m0 = np.array([0., 1.] * num_harmonics + [m0_v, 0.])
    m_and_cov = disc_harmonic_chirp_lcd(lam, b, ell, sigma, num_harmonics=num_harmonics, freq_scale=freq_scale)
    return drift, dispersion, m_and_cov, m0, P0, H


def build_lascala_model(params):
    """models.py:497-519.  params = delta, ell, sigma, m0_v."""
    delta, ell, sigma, m0_v = [float(p) for p in params]
    drift, dispersion, _, P0, H = model_lascala(ell, sigma, delta)
    m0 = np.array([0., 0., m0_v, 0.])
    return drift, dispersion, disc_model_lascala_lcd(ell, sigma), m0, P0, H


def build_kpt_chirp_model(params, fs, num_harmonics=1):
    """models.py:522-580 -> F, Sigma, m0, P0, h.  params = q1, q2, p0, f0, a0."""
    q1, q2, p0, f0, a0 = [float(p) for p in params]
    dim_x = num_harmonics + 2
    P0 = p0 * np.eye(dim_x)
    m0 = np.array([2 * math.pi * f0 / fs] + [a0] * num_harmonics + [0.])
    F = np.eye(dim_x)
    F[-1, 0] = 1.
    Gamma = np.eye(dim_x)[:, :-1]
    Sigma = Gamma @ np.diag(np.array([(2 * math.pi * q1 / fs) ** 2] + [q2] * num_harmonics)) @ Gamma.T
    G = np.eye(dim_x)[1:-1, :]
    ks = np.arange(1, num_harmonics + 1)

    def h(x):
        return np.dot(G @ x, np.sin(g(x[0] + x[-1]) * ks))

    return F, Sigma, m0, P0, h
