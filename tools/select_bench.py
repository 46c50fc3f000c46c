"""Smoothers with selected outputs against the full-row launch: kernel times (HIP events) at the CRLB shape (262 144 x 500, one lane per
trial), C2's (1000 x 10^4, the walk) and C5's (1000 x 10^4, d = 8 tile layout).   python tools/select_bench.py [--only crlb|c2|c5] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                   # noqa: E402
from chirpgp_amd import filters_smoothers as fs, _engine, tools                # noqa: E402
from chirpgp_amd.models import model_chirp, disc_chirp_lcd                     # noqa: E402


def timed(run, reps):
    run(); run()
    torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(reps):
        r = run()
    torch.cuda.synchronize()
    _engine.kernel_events = None
    del r
    return float(np.mean([a.elapsed_time(b) for n, a, b in ev if n == 'smoother']))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--buffers', type=int, default=0, help='CGP_DBG_LANE_BUFFERS: 0 / 2 = covariance rows one step ahead (default), 3 = two where the LDS allows')
    a = ap.parse_args()
    _engine.debug_set(_engine.DBG_LANE_BUFFERS, a.buffers)
    variants = (('full rows', dict()),
                ('full + E[g]', dict(select=dict(comp=-2, expect='softplus'))),
                ('E[g] only', dict(want=(False, False), select=dict(comp=-2, expect='softplus'))),
                ('mean + var only', dict(want=(False, False), select=dict(comp=-2, mean=True, var=True))),
                ('mean + var + E[g] only', dict(want=(False, False), select=dict(comp=-2, mean=True, var=True, expect='softplus'))))
    if a.only in (None, 'crlb'):
        B, T = 262144, 500
        _, _, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
        mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
        _, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666, batch=B, states=False)
        f = fs.ekf(mc, H, 0.1, m0, P0, 0.01, yss, want=(True, True, False))
        del yss
        for name, kw in variants:
            ms = timed(lambda: fs.eks(mc, f[0], f[1], 0.01, **kw), a.reps)
            nsel = sum(1 for k in ('mean', 'var', 'expect') if kw.get('select', {}).get(k))
            nbytes = 160 + (160 if kw.get('want', (True, True))[0] else 0) + 8 * nsel
            print(f'eks 262144 x 500 (lane)   {name:24s} {ms:7.3f} ms   {nbytes} B/step  {nbytes * B * T / ms / 1e6:7.0f} GB/s', flush=True)
        del f
        torch.cuda.empty_cache()
    for tag, kind in (('c2', 'ekf'), ('c5', 'harmonic')):
        if a.only not in (None, tag):
            continue
        wl = bench.make_workload(1000, 10000, kind=kind)
        ys = torch.from_numpy(wl['ys']).cuda()
        if kind == 'ekf':
            f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
            run = lambda **kw: fs.eks(wl['disc'], f[0], f[1], wl['dt'], **kw)
        else:
            f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
            run = lambda **kw: fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'], **kw)
        for name, kw in variants:
            ms = timed(lambda: run(**kw), a.reps)
            print(f'{tag} smoother 1000 x 10000      {name:24s} {ms:7.3f} ms', flush=True)
        del f, ys
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
