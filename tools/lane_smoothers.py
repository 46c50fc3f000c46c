"""tools/lane_smoothers.py [B]: the one-lane-per-trial smoothers (flags = CGP_THREAD_PER_TRIAL) at a large batch x 500 steps: eks, sgp_smoother,
cd_eks, cd_sgp_smoother on filtering results of the CRLB-shaped records; time and algorithmic GB/s (320 B per trial-step at d = 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirpgp_amd import filters_smoothers as fs, tools, _engine
from chirpgp_amd.quadratures import SigmaPoints
from chirpgp_amd.models import model_chirp, disc_chirp_lcd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
drift, disp, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
gh3 = SigmaPoints.gauss_hermite(4, 3)
_, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666, batch=B, states=False)
f = fs.ekf(mc, H, 0.1, m0, P0, 0.01, yss)
runs = {'eks': lambda fl: fs.eks(mc, f[0], f[1], 0.01, flags=fl),
        'sgp_smoother': lambda fl: fs.sgp_smoother(mc, gh3, f[0], f[1], 0.01, flags=fl),
        'cd_eks': lambda fl: fs.cd_eks(drift, disp, f[0], f[1], 0.01, flags=fl),
        'cd_sgp_smoother': lambda fl: fs.cd_sgp_smoother(drift, disp(None), gh3, f[0], f[1], 0.01, flags=fl)}
for name, run in runs.items():
    for tag, fl in (('lane', 0x4), ('default', 0)):
        if tag == 'default' and name.startswith('cd_sgp') and B > 16384:
            continue
        r = run(fl); torch.cuda.synchronize()
        ev = _engine.kernel_events = []
        for _ in range(3): r = run(fl)
        torch.cuda.synchronize(); _engine.kernel_events = None
        ms = min(a.elapsed_time(b) for _, a, b in ev)
        print(f'{name:16s} {tag:8s} {B} x {T}: {ms:8.3f} ms  {320 * B * T / ms / 1e9:6.2f} TB/s algorithmic', flush=True)
        del r
