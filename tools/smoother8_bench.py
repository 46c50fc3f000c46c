"""Times the d = 8 smoothers (rts on a linear model, eks and sgp_smoother on the 3-harmonic chirp) at B = 1000, T = 10^4:
separates the cost of the per-lane gain computation (cheap for rts) from the cooperative walk (tools script, GPU box)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from chirpgp_amd import filters_smoothers as fs, _engine
from chirpgp_amd.tools import lti_sde_to_disc

B, T = 1000, 10000
wl = bench.make_workload(B, T, kind='harmonic')
ys = torch.from_numpy(wl['ys']).cuda()
f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
rng = np.random.default_rng(0)
A = -np.eye(8) + 0.2 * rng.standard_normal((8, 8)); A -= np.eye(8) * max(0, np.max(np.real(np.linalg.eigvals(A))) + 0.2)
F, S = lti_sde_to_disc(A, 0.5 * np.eye(8), 1e-3); S = 0.5 * (S + S.T)
fl = fs.kf(F, S, wl['H'], wl['Xi'], wl['m0'], wl['P0'], ys)


def timeit(name, fn, n=5, **kw):
    fn(**kw); torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(n): fn(**kw)
    torch.cuda.synchronize(); _engine.kernel_events = None
    print(f'{name:40s} {np.mean([a.elapsed_time(b) for _, a, b in ev]):8.3f} ms')


for flags, tag in ((0, 'tile layout'), (0x2 | 0x10, 'generic')):
    kw = dict(flags=flags) if flags else {}
    timeit(f'ekf nh=3 filter [{tag}]', lambda **k: fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys, **k), **kw)
    timeit(f'sgp_filter nh=3 [{tag}]', lambda **k: fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys, **k), **kw)
    kw = dict(flags=flags) if flags else {}
    timeit(f'rts d=8 [{tag}]', lambda **k: fs.rts(F, S, fl[0], fl[1], **k), **kw)
    timeit(f'eks nh=3 [{tag}]', lambda **k: fs.eks(wl['disc'], f[0], f[1], wl['dt'], **k), **kw)
    timeit(f'sgp_smoother nh=3 [{tag}]', lambda **k: fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'], **k), **kw)
