"""Exact (tangent kernel) against difference-quotient gradients: time per objective evaluation and per fit, single record and R records in
lock step.   python tools/grad_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chirpgp_amd import mle, models as pm                                  # noqa: E402
from chirpgp_amd.toymodels import gen_chirp, meow_freq, constant_mag      # noqa: E402

INIT = np.array([0.1, 0.1, 0.1, 1., 1., 7.])


def record(T, seed, dt=1e-3, Xi=0.1):
    ts = np.linspace(dt, dt * T, T)
    _, phase = meow_freq(offset=8.)
    return gen_chirp(ts, constant_mag(1.), phase) + np.sqrt(Xi) * np.random.default_rng(seed).standard_normal(T)


def main():
    ys = record(3141, 555)
    th = pm.g_inv(INIT)
    for exact in (True, False):
        fun = mle.make_objective('ekf', pm.build_chirp_model, ys, 0.1, 1e-3, exact=exact)
        fun(th); fun(th)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fun(th)
        dt_eval = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        opt, res = mle.fit('ekf', pm.build_chirp_model, INIT, ys, 0.1, 1e-3, maxiter=200, exact=exact)
        dt_fit = time.perf_counter() - t0
        print(f'one record T = 3141, exact = {exact}: {dt_eval * 1e3:.2f} ms per value + gradient, fit {dt_fit:.2f} s ({res.nit} iterations, {res.nfev} evaluations, nll {res.fun:.6f})', flush=True)
    for R in (64, 1000):
        recs = np.stack([record(3141, 1000 + r) for r in range(R)])
        yd = torch.from_numpy(recs).cuda()
        ths = np.tile(th, (R, 1))
        for exact in (True, False):
            mle._value_and_grad_many('ekf', pm.build_chirp_model, ths, yd, 0.1, 1e-3, None, 1e-6, {}, exact=exact)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                mle._value_and_grad_many('ekf', pm.build_chirp_model, ths, yd, 0.1, 1e-3, None, 1e-6, {}, exact=exact)
            print(f'{R} records T = 3141, exact = {exact}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per value + gradient of all records', flush=True)


if __name__ == '__main__':
    main()
