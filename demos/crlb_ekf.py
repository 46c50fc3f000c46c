"""Counterpart of the reference's tetralith/jobs/crlb_ekf.py on the MI355X engine: simulate many chirp-SDE trajectories,
run the EKF on all of them in one batched launch (the reference's jax.vmap over ys), and report the mean and standard
deviation of the squared filtering errors of the chirp and of the frequency state per time step.

    python demos/crlb_ekf.py [-lam 0.1 -b 0.1 -delta 0.1 -ell 1 -sigma 1 -Xi 0.1] [--num-mcs 100000] [--T 500]
"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs                    # noqa: E402
from chirpgp_amd.models import model_chirp, disc_chirp_lcd          # noqa: E402


def simulate(m_and_cov, m0, P0, H, Xi, dt, T, n, rng):
    """Vectorised simulation of n trajectories of the discretised chirp SDE (crlb_ekf.py:39-57): the transition
    covariance does not depend on the state, so only the conditional mean is evaluated per step (NumPy, host)."""
    _, state_cov = m_and_cov(np.zeros(4), dt)
    chol_cov = np.linalg.cholesky(state_cov + 1e-300 * np.eye(4))
    x = m0[:, None] + np.linalg.cholesky(P0) @ rng.standard_normal((4, n))
    p = m_and_cov.params
    lam, ell, sigma = p[0], p[2], p[3]
    from chirpgp_amd.models import _m32, g
    M, _ = _m32(ell, sigma, dt)
    xs, ys = np.empty((n, T, 4)), np.empty((n, T))
    for k in range(T):
        th = dt * 2 * math.pi * g(x[2])
        c, s = np.cos(th) * math.exp(-lam * dt), np.sin(th) * math.exp(-lam * dt)
        mean = np.stack([c * x[0] - s * x[1], s * x[0] + c * x[1], M[0, 0] * x[2] + M[0, 1] * x[3], M[1, 0] * x[2] + M[1, 1] * x[3]])
        x = mean + chol_cov @ rng.standard_normal((4, n))
        xs[:, k] = x.T
        ys[:, k] = H @ x + math.sqrt(Xi) * rng.standard_normal(n)
    return xs, ys


def main():
    ap = argparse.ArgumentParser()
    for name, default in (('-lam', 0.1), ('-b', 0.1), ('-delta', 0.1), ('-ell', 1.0), ('-sigma', 1.0), ('-Xi', 0.1)):
        ap.add_argument(name, type=float, default=default)
    ap.add_argument('--num-mcs', type=int, default=100000)
    ap.add_argument('--T', type=int, default=500)
    args = ap.parse_args()

    _, _, m0, P0, H = model_chirp(args.lam, args.b, args.ell, args.sigma, args.delta)
    m_and_cov = disc_chirp_lcd(args.lam, args.b, args.ell, args.sigma)
    dt, T = 0.01, args.T
    rng = np.random.default_rng(666)
    t0 = time.time()
    xss, yss = simulate(m_and_cov, m0, P0, H, args.Xi, dt, T, args.num_mcs, rng)
    t1 = time.time()
    mfs, _, _ = fs.ekf(m_and_cov, H, args.Xi, m0, P0, dt, yss, want=(True, False, False))
    t2 = time.time()
    err_chirp = (mfs[:, :, 1] - xss[:, :, 1]) ** 2
    err_v = (mfs[:, :, 2] - xss[:, :, 2]) ** 2
    print(f'{args.num_mcs} trials x {T} steps: simulate {t1 - t0:.2f} s (host), EKF {t2 - t1:.3f} s (GPU incl. transfers)')
    for k in (0, T // 2, T - 1):
        print(f'  t = {dt * (k + 1):5.2f}  chirp err {err_chirp[:, k].mean():.4e} +- {err_chirp[:, k].std():.2e}   '
              f'v err {err_v[:, k].mean():.4e} +- {err_v[:, k].std():.2e}')


if __name__ == '__main__':
    main()
