"""Counterpart of the reference's tetralith/jobs/crlb_ekf.py (and, with --filter ghf, tetralith/jobs/crlb_ghf.py) on the MI355X engine: simulate many chirp-SDE trajectories
on the device (cgp_simulate), run the EKF on all of them in one batched launch (the reference's jax.vmap over ys), and
report the mean and standard deviation of the squared filtering errors of the chirp and of the frequency state per
time step.  Nothing but the per-step error statistics (T doubles each) leaves HBM.

    python demos/crlb_ekf.py [-lam 0.1 -b 0.1 -delta 0.1 -ell 1 -sigma 1 -Xi 0.1] [--num-mcs 1000000] [--T 500] [--chunk 250000]

With torch.distributed initialised (python -m torch.distributed.run --nproc-per-node N demos/crlb_ekf.py ...) every rank
takes a contiguous block of the trials -- the counter-based generator gives trial i the same draws on any rank -- and
the per-step error sums are all-reduced (the natural collective of this job, 4 T doubles).
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs, tools, _engine              # noqa: E402
from chirpgp_amd.models import model_chirp, disc_chirp_lcd                    # noqa: E402
from chirpgp_amd.parallel import shard_bounds                                 # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    for name, default in (('-lam', 0.1), ('-b', 0.1), ('-delta', 0.1), ('-ell', 1.0), ('-sigma', 1.0), ('-Xi', 0.1)):
        ap.add_argument(name, type=float, default=default)
    ap.add_argument('--num-mcs', type=int, default=1000000)
    ap.add_argument('--T', type=int, default=500)
    ap.add_argument('--chunk', type=int, default=250000, help='trials per launch (bounds the HBM held at once)')
    ap.add_argument('--seed', type=int, default=666)
    ap.add_argument('--filter', default='ekf', choices=['ekf', 'ghf'], help='ghf: Gauss-Hermite order-3 sigma-point filter (crlb_ghf.py:64-75)')
    ap.add_argument('--save', default=None, help='write the error statistics to this .npz (keys of crlb_ekf.py:92-95)')
    args = ap.parse_args(argv)

    rank, world = 0, 1
    if 'RANK' in os.environ:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        dist.init_process_group('nccl')
        rank, world = dist.get_rank(), dist.get_world_size()

    _, _, m0, P0, H = model_chirp(args.lam, args.b, args.ell, args.sigma, args.delta)
    m_and_cov = disc_chirp_lcd(args.lam, args.b, args.ell, args.sigma)
    dt, T = 0.01, args.T
    if args.filter == 'ghf':
        from chirpgp_amd.quadratures import SigmaPoints
        sgps = SigmaPoints.gauss_hermite(d=4, order=3)

        def filtering(yss):
            return fs.sgp_filter(m_and_cov, sgps, H, args.Xi, m0, P0, dt, yss, want=(True, False, False))
    else:
        def filtering(yss):
            return fs.ekf(m_and_cov, H, args.Xi, m0, P0, dt, yss, want=(True, False, False))
    lo, hi = shard_bounds(args.num_mcs, rank, world)
    sums = torch.zeros((2, 2, T), dtype=torch.float64, device='cuda')     # [chirp | v][sum e, sum e^2][step], e = squared error
    # warm-up: library load, code-object load and the first allocations are one-off costs of the process
    _, yw = tools.simulate_measurements(m_and_cov, H, args.Xi, m0, P0, dt, 8, args.seed, batch=64)
    filtering(yw)
    torch.cuda.synchronize()
    t0 = time.time()
    for first in range(lo, hi, args.chunk):
        n = min(args.chunk, hi - first)
        xss, yss = tools.simulate_measurements(m_and_cov, H, args.Xi, m0, P0, dt, T, args.seed, batch=n, trial0=first)
        mfs, _, _ = filtering(yss)
        _engine.squared_error_sums(mfs, xss, (1, 2), sums)               # reduced over the trials on the device: no torch arithmetic
        del xss, yss, mfs
    if world > 1:
        dist.all_reduce(sums)
    torch.cuda.synchronize()
    t1 = time.time()
    if rank == 0:
        n = args.num_mcs
        mean_c, mean_v = sums[0, 0] / n, sums[1, 0] / n
        std_c = (sums[0, 1] / n - mean_c ** 2).clamp_min(0).sqrt()
        std_v = (sums[1, 1] / n - mean_v ** 2).clamp_min(0).sqrt()
        print(f'{n} trials x {T} steps on {world} GPU(s): simulate + {args.filter.upper()} + error statistics {t1 - t0:.3f} s')
        if args.save:
            import numpy as np
            np.savez(args.save, ts=dt * np.arange(1, T + 1), err_mean_chirps=mean_c.cpu().numpy(), err_std_chirps=std_c.cpu().numpy(),
                     err_mean_vs=mean_v.cpu().numpy(), err_std_vs=std_v.cpu().numpy())
        for k in (0, T // 2, T - 1):
            print(f'  t = {dt * (k + 1):5.2f}  chirp err {mean_c[k].item():.4e} +- {std_c[k].item():.2e}   '
                  f'v err {mean_v[k].item():.4e} +- {std_v[k].item():.2e}')
    if world > 1:
        dist.destroy_process_group()
    return (mean_c, std_c, mean_v, std_v) if rank == 0 else None


if __name__ == '__main__':
    main()
