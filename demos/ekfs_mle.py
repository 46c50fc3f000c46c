"""Counterpart of the reference's demos/ekfs_mle.py on the MI355X engine: MLE -> EKF -> EKS -> E[g(V)] -> RMSE.

    python demos/ekfs_mle.py [--method ekf|sgp_filter|cd_ekf] [--T 3141] [--seed 555] [--exact]

--exact: the EKF objective's gradient from the tangent kernel (forward tangents through the scan, what jax.value_and_grad gives the
reference, demos/ekfs_mle.py:43-48) instead of 13-probe central differences.  The estimate E[g(V)] rides in the smoother launch.
"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs, mle                                   # noqa: E402
from chirpgp_amd.models import g, build_chirp_model                                   # noqa: E402
from chirpgp_amd.quadratures import SigmaPoints, gaussian_expectation                 # noqa: E402
from chirpgp_amd.toymodels import gen_chirp, meow_freq, constant_mag, damped_exp_mag, random_ou_mag   # noqa: E402
from chirpgp_amd.tools import rmse                                                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--method', default='ekf', choices=['ekf', 'sgp_filter', 'cd_ekf'])
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--seed', type=int, default=555)
    ap.add_argument('--exact', action='store_true', help='exact gradients (method ekf only)')
    args = ap.parse_args()

    dt, T = 0.001, args.T
    ts = np.linspace(dt, dt * T, T)
    rng = np.random.default_rng(args.seed)
    true_freq_func, true_phase_func = meow_freq(offset=8.)
    sgps = SigmaPoints.gauss_hermite(d=4, order=3)
    Xi = 0.1

    for name, mag in (('constant', constant_mag(1.)), ('damped', damped_exp_mag(0.3)), ('random OU', random_ou_mag(1., 1., rng))):
        ys = gen_chirp(ts, mag, true_phase_func) + math.sqrt(Xi) * rng.standard_normal(T)
        t0 = time.time()
        opt_params, res = mle.fit(args.method, build_chirp_model, [0.1, 0.1, 0.1, 1., 1., 7.], ys, Xi, dt, sgps=sgps,
                                  exact=True if (args.exact and args.method == 'ekf') else None)
        drift, dispersion, m_and_cov, m0, P0, H = build_chirp_model(opt_params)
        if args.method == 'ekf':
            mfs, Pfs, _ = fs.ekf(m_and_cov, H, Xi, m0, P0, dt, ys)
            # gaussian_expectation(ms=mss[:, 2], chol_Ps=sqrt(Pss[:, 2, 2]), func=g) of the reference, written by the smoother launch itself
            mss, Pss, sel = fs.eks(m_and_cov, mfs, Pfs, dt, select=dict(comp=2, expect='softplus'))
        elif args.method == 'sgp_filter':
            mfs, Pfs, _ = fs.sgp_filter(m_and_cov, sgps, H, Xi, m0, P0, dt, ys)
            mss, Pss, sel = fs.sgp_smoother(m_and_cov, sgps, mfs, Pfs, dt, select=dict(comp=2, expect='softplus'))
        else:
            mfs, Pfs, _ = fs.cd_ekf(drift, dispersion, H, Xi, m0, P0, dt, ys)
            mss, Pss, sel = fs.cd_eks(drift, dispersion, mfs, Pfs, dt, select=dict(comp=2, expect='softplus'))
        est = sel['expect']
        assert np.allclose(est, gaussian_expectation(ms=mss[:, 2], chol_Ps=np.sqrt(Pss[:, 2, 2]), func=g, force_shape=True)[:, 0], rtol=1e-12)
        print(f'{name:10s} params {np.array2string(opt_params, precision=3)}  nll {res.fun:.2f}  iters {res.nit} '
              f'({res.nfev} launches)  RMSE {rmse(true_freq_func(ts), est):.3f} Hz  [{time.time() - t0:.2f} s]')


if __name__ == '__main__':
    main()
