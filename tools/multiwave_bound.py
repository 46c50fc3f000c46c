"""tools/multiwave_bound.py: what several wavefronts per trial could buy the d = 4 sigma-point filters at small batch.

Splitting the sigma-point GROUPS of a prediction over W wavefronts can only remove the work a wavefront does per extra pass over its
groups: every lane evaluates ONE group's softplus -> sin / cos chain whatever the set (16 groups fill the 64 lanes of a pass), and
the factorisation, the moment sums' tail and the Kalman update are per trial, not per group.  So the filter's time with a ONE-pass
set (cubature, 7 groups) is a lower bound for any W >= 2 split of the 27-group Gauss-Hermite set -- before the LDS exchange and the
barrier such a split needs (two per step for sgp_filter, eight for cd_sgp_filter's RK4 stages) cost anything."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chirpgp_amd import filters_smoothers as fs, _engine
from chirpgp_amd.quadratures import SigmaPoints


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    _engine.kernel_events = None
    return min(a.elapsed_time(b) for n, a, b in ev if n == 'filter')


for kind, T in (('sgp', 10000), ('cd_sgp', 10000)):
    for B in (125, 512, 1000):
        wl = bench.make_workload(B, T, kind=kind)
        ys = torch.from_numpy(wl['ys']).cuda()
        a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
        row = []
        for name, sg in (('GH-3, 81 points in 27 groups (two passes)', SigmaPoints.gauss_hermite(4, 3)), ('cubature, 8 points in 7 groups (one pass)', SigmaPoints.cubature(4))):
            if kind == 'sgp':
                t = timed(lambda: fs.sgp_filter(wl['disc'], sg, *a))
            else:
                t = timed(lambda: fs.cd_sgp_filter(wl['drift'], wl['disp'](None), sg, *a))
            row.append(t)
            print(f'{kind:7s} B = {B:5d} T = {T}: {name:45s} {t:8.3f} ms')
        print(f'        upper bound of any split of the groups over wavefronts: {row[0] / row[1]:.3f} x')
