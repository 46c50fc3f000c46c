"""tools/spread_probe_sgp.py [kind ...]: the C3 / C5 / C4 filters (sgp_filter, cd_sgp_filter; 1000 x 10^4) on record sets away from the bench's: kernel time by (Xi, offset)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
from chirpgp_amd.quadratures import SigmaPoints
for kind in (sys.argv[1:] or ('sgp', 'harmonic', 'cd_sgp')):
    for Xi, off in ((0.1, 8.0), (0.1, 20.0), (1.0, 8.0)):
        wl = bench.make_workload(4, 100, kind=kind)
        nh = 3 if kind == 'harmonic' else 0
        ys = torch.from_numpy(bench.chirp_batch(1000, 10000, 0, Xi=Xi, offset=off, num_harmonics=nh)).cuda()
        if kind == 'cd_sgp':
            run = lambda: fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], wl['H'], Xi, wl['m0'], wl['P0'], 1e-3, ys)
        else:
            run = lambda: fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], Xi, wl['m0'], wl['P0'], 1e-3, ys)
        run(); torch.cuda.synchronize()
        ev = _engine.kernel_events = []
        for _ in range(3): r = run()
        torch.cuda.synchronize(); _engine.kernel_events = None
        print(f'{kind} Xi={Xi} offset={off}: filter {min(a.elapsed_time(b) for _, a, b in ev):.3f} ms', flush=True)
