"""tools/split_filters.py: the time-split filters with burn-in (cgp_filter_time_split) at the per-GPU shard sizes of BASELINE C2 / C3 / C5
(125 x 10 000) and C4 (512 x 50 000): kernel time, the launch's own junction mismatch and the worst difference from the sequential
launch's outputs, for several (segments, burn-in)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chirpgp_amd import filters_smoothers as fs, _engine


def timed(fn, reps=3):
    out = fn(); torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize(); _engine.kernel_events = None
    return out, min(a.elapsed_time(b) for n, a, b in ev if n == 'filter')


cases = [('ekf', 125, 10000), ('sgp', 125, 10000), ('harmonic', 125, 10000), ('cd_sgp', 512, 50000)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for kind, B, T in cases:
    wl = bench.make_workload(B, T, kind=kind)
    ys = torch.from_numpy(wl['ys']).cuda()
    a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
    if kind == 'ekf':
        run = lambda **kw: fs.ekf(wl['disc'], *a, **kw)
    elif kind == 'cd_sgp':
        run = lambda **kw: fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], *a, **kw)
    else:
        run = lambda **kw: fs.sgp_filter(wl['disc'], wl['sgps'], *a, **kw)
    seq, t_seq = timed(run)
    print(f'{kind:9s} {B} x {T}: sequential {t_seq:8.3f} ms')
    for segs, burn in ((2, 3008), (4, 3008), (8, 3008), (8, 2048), (8, 4096), (16, 3008)):
        if kind == 'cd_sgp' and segs > 4:
            continue
        got, t = timed(lambda: run(time_split=(segs, burn)))
        err = float(_engine.last_junction_error.max())
        worst = max(float((g - s).abs().max() / s.abs().max()) for g, s in zip(got, seq))
        print(f'          {segs:2d} segments, burn-in {burn}: {t:8.3f} ms ({t_seq / t:4.2f} x)   junction mismatch {err:.1e}   worst output difference {worst:.1e}')
