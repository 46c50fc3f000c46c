"""One variant of the large-batch eks with selected outputs, for the profiler:  python tools/select_probe.py full|meanvar|expect [reps] [B] [T]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chirpgp_amd import filters_smoothers as fs, _engine, tools
from chirpgp_amd.models import model_chirp, disc_chirp_lcd

variant = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
T = int(sys.argv[4]) if len(sys.argv) > 4 else 500
kw = {'full': {}, 'meanvar': dict(want=(False, False), select=dict(comp=-2, mean=True, var=True)),
      'expect': dict(want=(False, False), select=dict(comp=-2, expect='softplus'))}[variant]
_, _, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
_, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666, batch=B, states=False)
f = fs.ekf(mc, H, 0.1, m0, P0, 0.01, yss, want=(True, True, False))
del yss
for _ in range(2):
    r = fs.eks(mc, f[0], f[1], 0.01, **kw)
torch.cuda.synchronize()
ev = _engine.kernel_events = []
for _ in range(reps):
    r = fs.eks(mc, f[0], f[1], 0.01, **kw)
torch.cuda.synchronize()
_engine.kernel_events = None
ms = float(np.mean([a.elapsed_time(b) for n, a, b in ev if n == 'smoother']))
nsel = sum(1 for k in ('mean', 'var', 'expect') if kw.get('select', {}).get(k))
nbytes = 160 + (160 if kw.get('want', (True, True))[0] else 0) + 8 * nsel
print(f'eks {B} x {T}, one lane per trial, {variant}: {ms:.3f} ms, {nbytes} algorithmic B per trial-step = {nbytes * B * T / 1e9:.2f} GB, {nbytes * B * T / ms / 1e6:.0f} GB/s')
