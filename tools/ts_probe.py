"""eks d = 4 at B = 125, T = 10 000 for rocprofv3 --kernel-trace --stats: time-split passes seen separately."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chirpgp_amd import filters_smoothers as fs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 125
kind = sys.argv[2] if len(sys.argv) > 2 else 'ekf'
if len(sys.argv) > 3:                 # cap on the time-split segments of this context (cgp_debug_set)
    from chirpgp_amd import _engine
    _engine.debug_set(_engine.DBG_WALK_SEGMENTS, int(sys.argv[3]))
wl = bench.make_workload(B, 10000, kind=kind)
ys = torch.from_numpy(wl['ys']).cuda()
if kind in ('ekf', 'harmonic_ekf'):
    f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
    for _ in range(10):
        s = fs.eks(wl['disc'], f[0], f[1], wl['dt'])
else:
    f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
    for _ in range(10):
        s = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'])
torch.cuda.synchronize()
