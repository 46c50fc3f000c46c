"""NumPy restatement of the pieces of /root/reference/chirpgp/tools.py and toymodels.py that sit
either side of the hot path (TEST INFRASTRUCTURE).  Random streams are NumPy Generators: the
reference's jax.random threefry streams cannot be reproduced without JAX.
"""
import math
import numpy as np
import scipy.linalg

__all__ = ['lti_sde_to_disc', 'rmse', 'simulate_lgssm', 'simulate_sde',
           'gen_chirp', 'gen_harmonic_chirp', 'constant_mag', 'damped_exp_mag', 'random_ou_mag',
           'affine_freq', 'meow_freq', 'tiled_meow']


def lti_sde_to_disc(A, B, dt):
    """Van Loan / Axelsson-Gustafsson discretisation of dX = A X dt + B dW; tools.py:44-78."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    d = A.shape[0]
    BBt = (B ** 2).reshape(1, 1) if B.ndim == 0 else (np.outer(B, B) if B.ndim == 1 else B @ B.T)
    F = scipy.linalg.expm(A * dt)
    phi = np.vstack([np.hstack([A, BBt]), np.hstack([np.zeros_like(A), -A.T])])
    AB = scipy.linalg.expm(phi * dt) @ np.vstack([np.zeros_like(A), np.eye(d)])
    return F, AB[0:d, :] @ F.T


def rmse(x1, x2, reduce_sum=True):
    """tools.py:279-293."""
    val = np.sqrt(np.mean((np.asarray(x1) - np.asarray(x2)) ** 2, axis=0))
    return np.sum(val) if reduce_sum else val


def simulate_lgssm(F, Sigma, x0, T, rng):
    """tools.py:81-116 with a NumPy Generator."""
    chol = np.linalg.cholesky(Sigma)
    xs = np.zeros((T, np.size(x0)))
    x = np.asarray(x0, dtype=np.float64)
    for k in range(T):
        x = F @ x + chol @ rng.standard_normal(x.size)
        xs[k] = x
    return xs


def simulate_sde(m_and_cov, m0, P0, dt, T, rng):
    """tools.py:119-170: x0 ~ N(m0, P0), x_k ~ N(mean(x_{k-1}), cov(x_{k-1})); PSD-safe factor."""
    def factor(C):
        w, V = np.linalg.eigh(C)
        return V * np.sqrt(np.clip(w, 0., None))
    x = np.asarray(m0, dtype=np.float64) + factor(np.asarray(P0)) @ rng.standard_normal(np.size(m0))
    xs = np.zeros((T, x.size))
    for k in range(T):
        m, cov = m_and_cov(x, dt)
        x = m + factor(cov) @ rng.standard_normal(x.size)
        xs[k] = x
    return xs


# --------------------------------------------------------------------------- toy chirps (toymodels.py)
def gen_chirp(ts, magnitude_func, phase_func, base_phase=0.):
    """toymodels.py:37-70."""
    return magnitude_func(ts) * np.sin(base_phase + 2 * math.pi * phase_func(ts))


def gen_harmonic_chirp(ts, magnitude_funcs, fundamental_phase_func, base_phase=0.):
    """toymodels.py:73-104."""
    ys = np.zeros_like(ts)
    for i, mag in enumerate(magnitude_funcs):
        ys = ys + mag(ts) * np.sin(base_phase + (i + 1) * 2 * math.pi * fundamental_phase_func(ts))
    return ys


def constant_mag(b):
    """toymodels.py:122-130."""
    return lambda ts: np.ones_like(ts) * b


def damped_exp_mag(damp_rate):
    """toymodels.py:133-141."""
    return lambda ts: np.exp(-damp_rate * ts)


def random_ou_mag(ell, sigma, rng):
    """toymodels.py:144-167: one OU realisation, x0 ~ N(0, sigma^2)."""
    def generate(ts):
        dt = np.diff(ts)[0]
        a = math.exp(-dt / ell)
        q = math.sqrt(sigma ** 2 * (1 - math.exp(-2 * dt / ell)))
        x = sigma * rng.standard_normal()
        out = np.zeros(ts.size)
        for k in range(ts.size):
            x = a * x + q * rng.standard_normal()
            out[k] = x
        return out
    return generate


def affine_freq(a, b):
    """toymodels.py:170-190."""
    return (lambda ts: a * ts + b), (lambda ts: 0.5 * a * ts ** 2 + b * ts)


def meow_freq(mag=500., scale=5., offset=5.5):
    """toymodels.py:226-268; valid on (0, pi)."""
    def freq(ts):
        return mag * scale * np.cos(ts) / (np.sin(ts) ** 2) * np.exp(-scale / np.sin(ts)) + offset

    def phase(ts):
        return mag * np.exp(-scale / np.sin(ts)) + offset * ts
    return freq, phase


def tiled_meow(T, dt=1e-3, mag=500., scale=5., offset=8., window=3141):
    """Benchmark input: the meow frequency law tiled in `window`-step windows so that
    it stays valid for T > 3141.  At both ends of a window exp(-scale/sin t) -> 0, so phase and frequency
    are continuous across windows.  Returns (ts, freq(ts), phase(ts))."""
    k = np.arange(T)
    local = (k % window + 1) * dt
    base = (k // window) * (offset * window * dt)
    f, p = meow_freq(mag, scale, offset)
    ts = (k + 1) * dt
    return ts, f(local), base + p(local)
