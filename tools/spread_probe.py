"""tools/spread_probe.py: the C2 filter (ekf, 1000 x 10^4) on record sets away from the headline's (measurement noise Xi, frequency offset):
kernel time, worst error against the port on 16 records, regime counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
from oracle import port
from tests import cases as cs
drift, disp, disc, m0, P0, H = pm.build_chirp_model(np.array([0.1, 0.1, 0.1, 1., 1., 7.]))
for Xi, off in ((0.1, 8.0), (0.1, 5.5), (0.1, 20.0), (1.0, 8.0), (1.0, 20.0), (0.01, 5.5)):
    ys_h = bench.chirp_batch(1000, 10000, 0, Xi=Xi, offset=off)
    ys = torch.from_numpy(ys_h).cuda()
    run = lambda: fs.ekf(disc, H, Xi, m0, P0, 1e-3, ys)
    run(); torch.cuda.synchronize()
    ev = _engine.kernel_events = []
    for _ in range(4): r = run()
    torch.cuda.synchronize(); _engine.kernel_events = None
    ms = min(a.elapsed_time(b) for _, a, b in ev)
    _engine.debug_set(_engine.DBG_COUNT_REGIMES, 1); _engine.debug_counters(reset=True); run(); rg = _engine.debug_counters(reset=True); _engine.debug_set(_engine.DBG_COUNT_REGIMES, 0)
    want = port.filter(port.F_EKF, disc, None, H, Xi, m0, P0, 1e-3, ys_h[:16])
    err = max(cs.max_rel_err(g[:16].cpu().numpy(), w) for g, w in zip(r, want))
    print(f'Xi={Xi} offset={off}: filter {ms:.3f} ms, worst error vs port (16 records) {err:.1e}, {rg}', flush=True)
