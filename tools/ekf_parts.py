"""Headline filter (ekf4_mfma_kernel, C2: 1000 x 10^4) with outputs switched off one by one: what the stores, the NLL flush and the chain cost."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from chirpgp_amd import filters_smoothers as fs, _engine

wl = bench.make_workload(1000, 10000, kind='ekf')
ys = torch.from_numpy(wl['ys']).cuda()
a = (wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
for name, kw in (('mfs + Pfs + nll', {}), ('mfs + nll', dict(want=(True, False, True))), ('Pfs + nll', dict(want=(False, True, True))), ('nll rows only', dict(want=(False, False, True))),
                 ('mfs + Pfs', dict(want=(True, True, False))), ('mfs only', dict(want=(True, False, False))),
                 ('final nll only', dict(want=(False, False, True), nll_final_only=True))):
    for _ in range(3):
        r = fs.ekf(*a, **kw)
    torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(10):
        r = fs.ekf(*a, **kw)
    torch.cuda.synchronize()
    _engine.kernel_events = None
    ms = np.array([x.elapsed_time(y) for n, x, y in ev])
    print(f'{name:20s} {ms.mean():.3f} ms (min {ms.min():.3f})  = {ms.mean() * 1e-3 * 2.4e9 / 1e4:.0f} cycles a step at 2.4 GHz', flush=True)
