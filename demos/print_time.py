"""Counterpart of the reference's timing script paper_plots_tables/print_time.py:14-61 on the MI355X engine: MLE of the chirp
model on one toy record (T = 3141), then ONE timed EKF + EKS pass at the estimated parameters (after a warm-up pass, as the
reference triggers its jit first).  Prints the elapsed wall time like the reference does.

    python demos/print_time.py [--T 3141] [--maxiter 200]
"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from _pipeline import INIT_PARAMS                                   # noqa: E402
from chirpgp_amd import filters_smoothers as fs, mle                # noqa: E402
from chirpgp_amd.models import build_chirp_model                    # noqa: E402
from chirpgp_amd.toymodels import gen_chirp, meow_freq, constant_mag  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--maxiter', type=int, default=200)
    a = ap.parse_args(argv)
    import torch
    dt, Xi = 0.001, 0.1
    ts = np.linspace(dt, dt * a.T, a.T)
    _, true_phase_func = meow_freq(offset=8.)
    rng = np.random.default_rng(666)
    ys = gen_chirp(ts, constant_mag(1.), true_phase_func) + math.sqrt(Xi) * rng.standard_normal(a.T)
    opt_params, res = mle.fit('ekf', build_chirp_model, INIT_PARAMS, ys, Xi, dt, maxiter=a.maxiter)
    print(f'Parameter learnt: {opt_params}. Convergence: {res.success} ({res.nit} iterations)')
    _, _, m_and_cov, m0, P0, H = build_chirp_model(opt_params)
    ys_dev = torch.from_numpy(ys).cuda()

    def one_pass():
        mfs, Pfs, _ = fs.ekf(m_and_cov, H, Xi, m0, P0, dt, ys_dev)
        return fs.eks(m_and_cov, mfs, Pfs, dt)

    one_pass()
    torch.cuda.synchronize()
    tic = time.time()
    one_pass()
    torch.cuda.synchronize()
    elapsed = time.time() - tic
    print(f'Elapsed {elapsed} seconds.')
    return elapsed


if __name__ == '__main__':
    main()
