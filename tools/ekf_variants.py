"""C2 filter timing with parts of the output switched off (where does a step's time go?)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from chirpgp_amd import _engine as E, filters_smoothers as fs

wl = bench.make_workload(1000, 10000, seed=0, kind='ekf')
ys = torch.from_numpy(wl['ys']).cuda()


def timed(**kw):
    for _ in range(2):
        fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys, **kw)
    torch.cuda.synchronize()
    E.kernel_events = []
    for _ in range(5):
        fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys, **kw)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for _, a, b in E.kernel_events]
    E.kernel_events = None
    return sum(ms) / len(ms)


for name, kw in [('all outputs', {}), ('no Pfs', dict(want=(True, False, True))), ('nll only', dict(want=(False, False, True))),
                 ('final nll only', dict(want=(False, False, True), nll_final_only=True)), ('no nll', dict(want=(True, True, False))),
                 ('dpp kernel, all outputs', dict(flags=0x80)), ('dpp kernel, final nll only', dict(flags=0x80, want=(False, False, True), nll_final_only=True)),
                 ('generic kernel', dict(flags=0x10))]:
    print(f'{name:32s} {timed(**kw):7.3f} ms', flush=True)
