import os, sys
sys.path.insert(0, '/root/repo')
import torch, bench
from chirpgp_amd import filters_smoothers as fs, _engine
wl = bench.make_workload(1000, 10000, kind='ekf')
ys = torch.from_numpy(wl['ys']).cuda()
a = (wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'])
def timed(reps=10, **kw):
    fs.ekf(*a, ys, **kw); torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(reps): fs.ekf(*a, ys, **kw)
    torch.cuda.synchronize(); _engine.kernel_events = None
    return min(x.elapsed_time(y) for n, x, y in ev if n == 'filter')
print('full outputs         ', timed())
print('no nll               ', timed(want=(True, True, False)))
print('nll only (final)     ', timed(want=(False, False, True), nll_final_only=True))
print('means + nll          ', timed(want=(True, False, True)))
print('no outputs but nll   ', timed(want=(False, False, True)))
