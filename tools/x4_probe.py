import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
from oracle import port
from tests import cases as cs
drift, disp, disc, m0, P0, H = pm.build_chirp_model(np.array([0.1, 0.1, 0.1, 1., 1., 7.]))
for Xi, off in ((0.1, 8.0), (0.1, 20.0), (1.0, 8.0)):
    ys_h = bench.chirp_batch(64, 2000, 0, Xi=Xi, offset=off)
    ys = torch.from_numpy(ys_h).cuda().repeat(64, 1)
    run = lambda: fs.ekf(disc, H, Xi, m0, P0, 1e-3, ys, flags=0x202)
    run(); torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(4): r = run()
    torch.cuda.synchronize(); _engine.kernel_events = None
    ms = min(a.elapsed_time(b) for _, a, b in ev)
    want = port.filter(port.F_EKF, disc, None, H, Xi, m0, P0, 1e-3, ys_h[:16])
    err = max(cs.max_rel_err(g[:16].cpu().numpy(), w) for g, w in zip(r, want))
    print(f'x4 kernel 4096 x 2000, Xi={Xi} offset={off}: {ms:.3f} ms, worst error vs port (16 records) {err:.1e}', flush=True)
