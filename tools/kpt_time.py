import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/demos')
from chirpgp_amd import filters_smoothers as fs, models as pm, _engine, mle
from tests import cases as cs
for nh in (1, 3):
    for B in (1, 11, 125, 1000):
        T = 3141
        prm = np.tile([0.02, 1e-5, 1e-5, 8., 1.], (B, 1))
        F, Sigma, m0, P0, h = pm.build_kpt_chirp_model(prm, 1000., nh)
        ys = torch.from_numpy(np.tile(cs.chirp_measurements(T, 1, num_harmonics=nh if nh > 1 else 0)[2], (B, 1))).cuda()
        call = lambda: fs.ekf_for_kpt(F, Sigma, h, 0.1, m0, P0, 1e-3, ys, nll_final_only=True, want=(False, False, True))
        call(); torch.cuda.synchronize()
        ev = _engine.kernel_events = []
        for _ in range(5): call()
        torch.cuda.synchronize(); _engine.kernel_events = None
        ms = min(a.elapsed_time(b) for _, a, b in ev)
        print(f'kpt nh={nh} B={B:5d} T={T}: {ms:.3f} ms  = {ms * 1e6 / T:.0f} ns/step', flush=True)
import _pipeline
t0 = time.time(); rows = _pipeline.demo('kpt', num_harmonics=1, signal_harmonics=0, family='kpt', T=3141, seed=5, mags=('const',), quiet=True); print('kpt_mle one record', time.time() - t0, rows)
t0 = time.time(); rows = _pipeline.demo('ekfs', T=3141, seed=5, mags=('const',), quiet=True); print('ekfs_mle one record', time.time() - t0, rows)
