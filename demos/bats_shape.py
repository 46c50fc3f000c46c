"""The shape of the reference's bat-call analyses (real_applications/bats/myotis_myotis_analysis.py:50-85,
eptesicus_nilssonii_analysis.py:49-84) on the MI355X engine: harmonic chirp model with 4 (or 5) harmonics -- d = 10 (12) --
frequency state scaled by 1e4, Xi = 1e-4, cubature filter + smoother on ONE record of T = 25 334 samples at 250 kHz, timed like
the reference times it ("Our method takes ... seconds", after a warm-up pass).  The recordings are not part of the reference:
the record here is a synthetic downward sweep with the same number of harmonics, normalised the same way.

    python demos/bats_shape.py [--harmonics 4] [--T 25334]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs                     # noqa: E402
from chirpgp_amd.models import g, build_harmonic_chirp_model        # noqa: E402
from chirpgp_amd.quadratures import SigmaPoints, gaussian_expectation   # noqa: E402


def sweep(T, nh, fs_hz=250000., seed=5):
    rng = np.random.default_rng(seed)
    t = np.arange(1, T + 1) / fs_hz
    f = 2.2e4 * (1 + 0.4 * np.exp(-3 * t / t[-1]))
    phase = 2 * np.pi * np.cumsum(f) / fs_hz
    ys = sum(0.5 ** k * np.sin((k + 1) * phase) for k in range(nh))
    return ys / np.max(np.abs(ys)) + 1e-2 * rng.standard_normal(T), f


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--harmonics', type=int, default=4)
    ap.add_argument('--T', type=int, default=25334)
    a = ap.parse_args(argv)
    import torch
    nh, freq_scale, Xi, dt = a.harmonics, 10000., 1e-4, 1. / 250000
    ys, truth = sweep(a.T, nh)
    params = np.array([0.1, 1., 1., 0.2, 10., 2.])                   # myotis_myotis_analysis.py:60
    _, _, m_and_cov, m0, P0, H = build_harmonic_chirp_model(params, nh, freq_scale)
    sgps = SigmaPoints.cubature(d=2 * nh + 2)
    ys_dev = torch.from_numpy(ys).cuda()

    def one_pass():
        mfs, Pfs, _ = fs.sgp_filter(m_and_cov, sgps, H, Xi, m0, P0, dt, ys_dev)
        return fs.sgp_smoother(m_and_cov, sgps, mfs, Pfs, dt)

    one_pass()
    torch.cuda.synchronize()
    tic = time.time()
    mss, Pss = one_pass()
    torch.cuda.synchronize()
    elapsed = time.time() - tic
    print(f'Our method takes {elapsed} seconds.')
    mss, Pss = mss.cpu().numpy(), Pss.cpu().numpy()
    est = gaussian_expectation(ms=mss[:, -2], chol_Ps=np.sqrt(Pss[:, -2, -2]), func=g, force_shape=True)[:, 0] * freq_scale
    tail = slice(a.T // 10, None)
    err = float(np.sqrt(np.mean((est[tail] - truth[tail]) ** 2)))
    print(f'd = {2 * nh + 2}, T = {a.T}: RMSE of the smoothed frequency {err:.1f} Hz on a {truth[-1]:.0f} - {truth[0]:.0f} Hz sweep')
    return elapsed, err


if __name__ == '__main__':
    main()
