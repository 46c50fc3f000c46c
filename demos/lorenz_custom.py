"""A model the library does not enumerate -- the Lorenz-63 system of the reference's test/test_ekfs.py:11-62 -- on the engine: the drift and
its discretisation are handed over as device SOURCE (chirpgp_amd.models.custom_sde / custom_cond_m_cov), compiled at run time into the
generic kernels, differentiated with dual numbers in the kernel (the reference: any traceable callable + jax.jacfwd).

    python demos/lorenz_custom.py [--T 2000] [--batch 64]
"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs, models                             # noqa: E402
from chirpgp_amd.quadratures import SigmaPoints                                     # noqa: E402

SOURCE = r'''
// p = [kappa, lam, mu, gamma]: drift of test/test_ekfs.py:20-23; the discretisation is one RK4 step for the mean and the second-order
// expansion Gamma dt + (J Gamma + Gamma J^T) dt^2 / 2 for the covariance (the reference's order-2 TME agrees with both to O(dt^3))
template <class T> __device__ void drift(const T* u, const double* p, T* a) {
    a[0] = p[0] * (u[1] - u[0]);
    a[1] = u[0] * (p[1] - u[2]) - u[1];
    a[2] = u[0] * u[1] - p[2] * u[2];
}
template <class T> __device__ void cond_mean(const T* u, const double* p, double dt, T* m) {
    T k1[3], k2[3], k3[3], k4[3], x[3];
    drift(u, p, k1);
    for (int i = 0; i < 3; i++) x[i] = u[i] + 0.5 * dt * k1[i];
    drift(x, p, k2);
    for (int i = 0; i < 3; i++) x[i] = u[i] + 0.5 * dt * k2[i];
    drift(x, p, k3);
    for (int i = 0; i < 3; i++) x[i] = u[i] + dt * k3[i];
    drift(x, p, k4);
    for (int i = 0; i < 3; i++) m[i] = u[i] + dt * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6;
}
__device__ void cond_cov(const double* u, const double* p, double dt, double* cov) {
    const double G = p[3];
    const double J[3][3] = {{-p[0], p[0], 0.0}, {p[1] - u[2], -1.0, -u[0]}, {u[1], u[0], -p[2]}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) cov[i * 3 + j] = (i == j ? G * dt : 0.0) + (J[i][j] * G + G * J[j][i]) * dt * dt / 2;
}
'''


def simulate(T, dt, Xi, seed, p):
    kappa, lam, mu, G = p

    def drift(u):
        return np.array([kappa * (u[1] - u[0]), u[0] * (lam - u[2]) - u[1], u[0] * u[1] - mu * u[2]])
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(3)
    xs = np.empty((T, 3))
    for k in range(T):
        k1 = drift(x); k2 = drift(x + 0.5 * dt * k1); k3 = drift(x + 0.5 * dt * k2); k4 = drift(x + dt * k3)
        x = x + dt * (k1 + 2 * k2 + 2 * k3 + k4) / 6 + math.sqrt(G * dt) * rng.standard_normal(3)
        xs[k] = x
    return xs, xs[:, 0] + math.sqrt(Xi) * rng.standard_normal(T)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=2000)
    ap.add_argument('--batch', type=int, default=64)
    a = ap.parse_args()
    dt, Xi = 1e-3, 2.
    p = np.array([10., 28., 2., 25.])
    H, m0, P0 = np.array([1., 0., 0.]), np.zeros(3), np.eye(3)
    sims = [simulate(a.T, dt, Xi, 666 + i, p) for i in range(a.batch)]
    xs, ys = np.stack([s[0] for s in sims]), np.stack([s[1] for s in sims])
    disc = models.custom_cond_m_cov(SOURCE, 3, p)
    sde, b = models.custom_sde(SOURCE, 3, p[:3], 5. * np.eye(3))
    sg = SigmaPoints.gauss_hermite(3, 3)
    runs = (('ekf + eks', lambda: fs.eks(disc, *fs.ekf(disc, H, Xi, m0, P0, dt, ys)[:2], dt)),
            ('cd_ekf + cd_eks', lambda: fs.cd_eks(sde, b, *fs.cd_ekf(sde, b, H, Xi, m0, P0, dt, ys)[:2], dt)),
            ('sgp_filter + sgp_smoother (GH-3)', lambda: fs.sgp_smoother(disc, sg, *fs.sgp_filter(disc, sg, H, Xi, m0, P0, dt, ys)[:2], dt)),
            ('cd_sgp_filter + cd_sgp_smoother (GH-3)', lambda: fs.cd_sgp_smoother(sde, 5. * np.eye(3), sg, *fs.cd_sgp_filter(sde, 5. * np.eye(3), sg, H, Xi, m0, P0, dt, ys)[:2], dt)))
    for name, run in runs:
        t0 = time.time()
        mss, Pss = run()                                   # the first call of a model compiles it (about a second), later ones are cached
        t1 = time.time()
        mss, Pss = run()
        t2 = time.time()
        err = np.sqrt(np.mean((mss - xs) ** 2, axis=(0, 1)))
        print(f'{name:40s} RMSE of the smoothed state {np.array2string(err, precision=3)}   first call {t1 - t0:.2f} s, then {t2 - t1:.3f} s '
              f'for {a.batch} records x {a.T} steps')


if __name__ == '__main__':
    main()
