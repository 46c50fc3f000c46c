import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from chirpgp_amd import filters_smoothers as fs, tools, _engine
from chirpgp_amd.models import model_chirp, disc_chirp_lcd
_, _, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
B, T = 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
_, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666, batch=B, states=False)
run = lambda: fs.ekf(mc, H, 0.1, m0, P0, 0.01, yss, flags=0x2)
run(); torch.cuda.synchronize()
ev = _engine.kernel_events = []
for _ in range(4): r = run()
torch.cuda.synchronize(); _engine.kernel_events = None
ms = min(a.elapsed_time(b) for _, a, b in ev)
_engine.debug_set(_engine.DBG_COUNT_REGIMES, 1); _engine.debug_counters(reset=True); run(); rg = _engine.debug_counters(reset=True); _engine.debug_set(_engine.DBG_COUNT_REGIMES, 0)
print(f'one-trial kernel on CRLB-like records {B} x {T}, dt = 0.01: {ms:.3f} ms = {ms*1e6/T:.0f} ns/step', rg)
