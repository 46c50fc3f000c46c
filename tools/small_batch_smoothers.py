"""Discrete smoothers at the per-GPU batch of BASELINE C3 / C5 sharded over 8 GPUs (125 trials, T = 10 000): the default launch
(time-split form: several wavefronts per trial) against one wavefront per trial (CGP_NO_TIME_SPLIT).
    python tools/small_batch_smoothers.py [B ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs, _engine

NO_TIME_SPLIT = 0x1000


def timed(call, reps=5):
    call(); torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(reps):
        call()
    torch.cuda.synchronize(); _engine.kernel_events = None
    return min(a.elapsed_time(b) for _, a, b in ev)


T = 10000
for B in [int(x) for x in sys.argv[1:]] or [125, 250, 500]:
    for kind in ('kf', 'ekf', 'sgp', 'harmonic_ekf', 'harmonic'):
        wl = bench.make_workload(B, T, kind=kind)
        ys = torch.from_numpy(wl['ys']).cuda()
        if kind == 'kf':
            f = fs.kf(wl['F'], wl['Sigma'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], ys)
            sm = lambda **kw: fs.rts(wl['F'], wl['Sigma'], f[0], f[1], **kw)
            name = 'rts d=4'
        elif kind in ('ekf', 'harmonic_ekf'):
            f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
            sm = lambda **kw: fs.eks(wl['disc'], f[0], f[1], wl['dt'], **kw)
            name = f'eks d={wl["d"]}'
        else:
            f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
            sm = lambda **kw: fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'], **kw)
            name = f'sgp_smoother d={wl["d"]}'
        t_def = timed(lambda: sm())
        t_one = timed(lambda: sm(flags=NO_TIME_SPLIT))
        print(f'B={B:5d} T={T} {name:20s} default {t_def:7.3f} ms   one wave per trial {t_one:7.3f} ms   x{t_one / t_def:.2f}', flush=True)
