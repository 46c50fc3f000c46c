"""tools/kpt_time.py [--no-mle]: ekf_for_kpt (tetralith/jobs/kpt_mle.py, harmonic_kpt_mle.py shapes: T = 3141, 1 and 3 harmonics), ns per step of
the tile-layout kernels (cgp_kpt8.hpp, the default one-wavefront-per-trial launch) beside the generic kernel they replace (flags = wave per
trial | generic kernel), then the wall time of the kpt_mle driver on one record."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'demos'))
import numpy as np, torch
from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
from tests import cases as cs
for nh in (1, 2, 3):
    for B in (1, 11, 125, 1000):
        T = 3141
        prm = np.tile([0.02, 1e-5, 1e-5, 8., 1.], (B, 1))
        F, Sigma, m0, P0, h = pm.build_kpt_chirp_model(prm, 1000., nh)
        ys = torch.from_numpy(np.tile(cs.chirp_measurements(T, 1, num_harmonics=nh if nh > 1 else 0)[2], (B, 1))).cuda()
        res = []
        for flags in (0x2, 0x12):
            call = lambda: fs.ekf_for_kpt(F, Sigma, h, 0.1, m0, P0, 1e-3, ys, nll_final_only=True, want=(False, False, True), flags=flags)
            call(); torch.cuda.synchronize()
            ev = _engine.kernel_events = []
            for _ in range(5): call()
            torch.cuda.synchronize(); _engine.kernel_events = None
            res.append(min(a.elapsed_time(b) for _, a, b in ev))
        print(f'kpt nh={nh} B={B:5d} T={T}: tile layout {res[0]:.3f} ms = {res[0] * 1e6 / T:.0f} ns/step | generic kernel {res[1]:.3f} ms = {res[1] * 1e6 / T:.0f} ns/step', flush=True)
if '--no-mle' not in sys.argv:
    import _pipeline
    for name, kw in (('kpt_mle', dict(num_harmonics=1, signal_harmonics=0)), ('harmonic_kpt_mle', dict(num_harmonics=3, signal_harmonics=3))):
        _pipeline.demo('kpt', family='kpt', T=3141, seed=5, mags=('const',), quiet=True, **kw)          # warm (code objects, allocator)
        t0 = time.time(); rows = _pipeline.demo('kpt', family='kpt', T=3141, seed=5, mags=('const',), quiet=True, **kw)
        print(f'{name} one record: {time.time() - t0:.3f} s', rows)
