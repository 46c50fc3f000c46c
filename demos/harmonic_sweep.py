"""BASELINE config C5 as one driver: harmonic chirp model (3 harmonics, d = 8, cubature rule), a grid of parameter vectors
over (lam, b, ell, sigma) evaluated for every record in ONE launch (NLL only, every record read in place by its G grid
points), arg-min per record, then the full sigma-point filter + smoother at the arg-min, E[g(V)], RMSE and the result files
of the reference's jobs (tetralith/jobs/harmonic_ckfs_mle.py; model and rule of demos/ghfs_harmonics_mle.py:25-64).

    python demos/harmonic_sweep.py [--records 8] [--T 3141] [--points 4] [--save DIR]
"""
import argparse
import itertools
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs, mle, results                         # noqa: E402
from chirpgp_amd.models import g, build_harmonic_chirp_model                          # noqa: E402
from chirpgp_amd.quadratures import SigmaPoints, gaussian_expectation                 # noqa: E402
from chirpgp_amd.toymodels import gen_harmonic_chirp, meow_freq, constant_mag         # noqa: E402
from chirpgp_amd.tools import rmse                                                    # noqa: E402


def parameter_grid(points, delta=0.1, m0_v=7.):
    """`points` log-spaced values of each of lam, b, ell, sigma around the demos' start point -> (points^4, 6) rows
    [lam, b, delta, ell, sigma, m0_1] (demos/ghfs_harmonics_mle.py:50)."""
    lam = np.geomspace(0.03, 1.0, points)
    b = np.geomspace(0.03, 1.0, points)
    ell = np.geomspace(0.3, 3.0, points)
    sigma = np.geomspace(0.3, 3.0, points)
    return np.array([[l_, b_, delta, e_, s_, m0_v] for l_, b_, e_, s_ in itertools.product(lam, b, ell, sigma)])


def sweep_and_smooth(yss, grid, Xi, dt, num_harmonics=3, sgps=None):
    """-> dict(best (R, 6), nll (R, G), argmin (R,), mss (R, T, d), Pss (R, T, d, d), est_freq (R, T))."""
    sgps = sgps or SigmaPoints.cubature(d=2 * num_harmonics + 2)
    best, nll, arg = mle.grid_search('sgp_filter', build_harmonic_chirp_model, grid, yss, Xi, dt, sgps=sgps, num_harmonics=num_harmonics)
    _, _, m_and_cov, m0, P0, H = build_harmonic_chirp_model(best, num_harmonics)     # one parameter vector per record
    mfs, Pfs, _ = fs.sgp_filter(m_and_cov, sgps, H, Xi, m0, P0, dt, yss)
    mss, Pss = fs.sgp_smoother(m_and_cov, sgps, mfs, Pfs, dt)
    R, T = mss.shape[0], mss.shape[1]
    est = gaussian_expectation(ms=mss[:, :, -2].reshape(-1), chol_Ps=np.sqrt(Pss[:, :, -2, -2]).reshape(-1), func=g,
                               force_shape=True)[:, 0].reshape(R, T)
    return dict(best=best, nll=nll, argmin=arg, mss=mss, Pss=Pss, est_freq=est)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--records', type=int, default=8)
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--points', type=int, default=4, help='grid points per parameter (the grid has points^4 vectors)')
    ap.add_argument('--harmonics', type=int, default=3)
    ap.add_argument('--seed', type=int, default=777)
    ap.add_argument('--save', default=None)
    a = ap.parse_args(argv)
    dt, Xi, T = 1e-3, 0.1, a.T
    ts = np.linspace(dt, dt * T, T)
    true_freq, true_phase = meow_freq(offset=8.)
    clean = gen_harmonic_chirp(ts, [constant_mag(1.)] * a.harmonics, true_phase)
    yss = np.stack([clean + math.sqrt(Xi) * np.random.default_rng(a.seed + r).standard_normal(T) for r in range(a.records)])
    grid = parameter_grid(a.points)
    t0 = time.time()
    out = sweep_and_smooth(yss, grid, Xi, dt, a.harmonics)
    wall = time.time() - t0
    errs = [float(rmse(true_freq(ts), out['est_freq'][r])) for r in range(a.records)]
    print(f'{a.records} records x {grid.shape[0]} grid points x T = {T}: sweep + filter + smoother in {wall:.2f} s '
          f'({a.records * grid.shape[0] * T / wall:.3g} trial-steps/s incl. host)')
    for r in range(a.records):
        print(f'  record {r}: grid point {out["argmin"][r]}  params {np.array2string(out["best"][r], precision=3)}  '
              f'nll {out["nll"][r, out["argmin"][r]]:.2f}  RMSE {errs[r]:.3f} Hz')
        if a.save:
            results.save_result(a.save, 'harmonic_sweep', 'const', r, out['mss'][r], out['Pss'][r], errs[r])
    return out, errs


if __name__ == '__main__':
    main()
