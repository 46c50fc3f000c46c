"""Config C5's sweep: 3-harmonic model (d = 8), cubature, B = grid points on ONE record (read in place by every trial:
cgp_filter's shared-record addressing), one parameter vector per trial, NLL-only output.  Prints trial-steps/s of the filter launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
from chirpgp_amd.quadratures import SigmaPoints

T = 1000
sg = SigmaPoints.cubature(8)
for B in (1000, 8192, 65536, 262144):
    rng = np.random.default_rng(0)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * rng.uniform(0.8, 1.2, size=(B, 6))
    drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, 3)
    ys = torch.from_numpy(bench.chirp_batch(1, T, 0, num_harmonics=3)[0]).cuda()        # ONE record of T doubles
    disc.params = torch.from_numpy(disc.params).cuda(); m0 = torch.from_numpy(m0).cuda(); P0 = torch.from_numpy(P0).cuda()
    kw = dict(nll_final_only=True, want=(False, False, True), trials_per_record=B)
    for meth in ('sgp_filter', 'ekf'):
        call = (lambda: fs.sgp_filter(disc, sg, H, 0.1, m0, P0, 1e-3, ys, **kw)) if meth == 'sgp_filter' else (lambda: fs.ekf(disc, H, 0.1, m0, P0, 1e-3, ys, **kw))
        call(); torch.cuda.synchronize()
        ev = _engine.kernel_events = []
        for _ in range(3): call()
        torch.cuda.synchronize(); _engine.kernel_events = None
        ms = np.mean([a.elapsed_time(b) for _, a, b in ev])
        print(f'B={B:7d} T={T} {meth:10s} NLL-only: {ms:8.2f} ms  {B * T / (ms * 1e-3):.3e} trial-steps/s')
