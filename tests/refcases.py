"""Deterministic inputs of the reference's own tests for the hot path, re-created without JAX.

test/test_filters_smoothers.py:14, 19-56 draws its data with ``np.random.seed(666)`` and
``np.random.randn`` only, the two parametrisations in order (RNG state carries over) -- so the
very same numbers can be drawn here.
"""
import math
import functools
import numpy as np


@functools.lru_cache(maxsize=None)
def linear_ou_cases():
    """The two linear OU cases (a, b) in ((1, 1), (2.1, 0.4)) of test_filters_smoothers.py:19-56."""
    state = np.random.get_state()
    np.random.seed(666)
    cases = []
    try:
        for a, b in ((1., 1.), (2.1, 0.4)):
            dim_x, dt = 3, 0.01
            A = -a * np.eye(dim_x)
            B = b * np.eye(dim_x)
            F = math.exp(-a * dt) * np.eye(dim_x)
            Sigma = b ** 2 / (2 * a) * (1 - math.exp(-2 * a * dt)) * np.eye(dim_x)
            Xi = 0.1
            H = np.ones((dim_x,))
            m0 = np.zeros((dim_x,))
            P0 = 0.1 * np.eye(dim_x)
            T = 1000
            xx, yy = np.zeros((T, dim_x)), np.zeros((T,))
            x = m0.copy()
            for i in range(T):
                x = F @ x + np.sqrt(Sigma) @ np.random.randn(dim_x)
                y = H @ x + np.sqrt(Xi) * np.random.randn()
                xx[i], yy[i] = x, y
            cases.append(dict(a=a, b=b, dim_x=dim_x, dt=dt, A=A, B=B, F=F, Sigma=Sigma, Xi=Xi, H=H,
                              m0=m0, P0=P0, xs=xx, ys=yy))
    finally:
        np.random.set_state(state)
    return tuple(cases)


def chirp_measurements(T, seed, dt=1e-3, Xi=0.1, num_harmonics=0, mag='const'):
    """Synthetic toy-chirp measurements of SURVEY.md section 8d (meow frequency law tiled past T = 3141)."""
    from oracle import np_tools as tl
    ts, freq, phase = tl.tiled_meow(T, dt=dt, offset=8.)
    rng = np.random.default_rng(seed)
    if mag == 'const':
        amp = np.ones(T)
    elif mag == 'damped':
        amp = np.exp(-0.3 * ts)
    else:
        raise ValueError(mag)
    if num_harmonics == 0:
        clean = amp * np.sin(2 * math.pi * phase)
    else:
        clean = sum(amp * np.sin((k + 1) * 2 * math.pi * phase) for k in range(num_harmonics))
    return ts, freq, clean + math.sqrt(Xi) * rng.standard_normal(T)
