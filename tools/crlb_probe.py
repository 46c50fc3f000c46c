"""tools/crlb_probe.py B T flags want [reps] [ekf|ghf]: the filter of the CRLB jobs' shape (tetralith/jobs/crlb_ekf.py:59-79; ghf = the
Gauss-Hermite order-3 sigma-point filter of crlb_ghf.py:64-75) a few times, for rocprofv3.
want = means | full | a 0/1 mask of (mfs, Pfs, nll);  flags = CGP_* launch-shape bits (0 = default, 4 = one lane per trial, 0x200 = four trials per wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirpgp_amd import filters_smoothers as fs, tools, _engine
from chirpgp_amd.quadratures import SigmaPoints
from chirpgp_amd.models import model_chirp, disc_chirp_lcd
B, T, flags = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 0)
w = {'means': '100', 'full': '111'}.get(sys.argv[4], sys.argv[4])      # or a mask: mfs, Pfs, nll wanted ('110' = no NLL rows)
want = tuple(c == '1' for c in w)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
method = sys.argv[6] if len(sys.argv) > 6 else 'ekf'
gh3 = SigmaPoints.gauss_hermite(4, 3)
_, _, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
_, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666, batch=B, states=False)
kw = dict(flags=flags) if flags else {}


def run():
    if method == 'ghf':
        return fs.sgp_filter(mc, gh3, H, 0.1, m0, P0, 0.01, yss, want=want, **kw)
    return fs.ekf(mc, H, 0.1, m0, P0, 0.01, yss, want=want, **kw)


r = run()
torch.cuda.synchronize()
ev = _engine.kernel_events = []
for _ in range(reps):
    r = run()
torch.cuda.synchronize()
_engine.kernel_events = None
ms = min(a.elapsed_time(b) for _, a, b in ev)
nb = (8 + 32 * want[0] + 128 * want[1] + 8 * want[2]) * B * T
print(f'{method} B={B} T={T} flags={flags:#x} {sys.argv[4]}: {ms:.3f} ms  {nb / ms / 1e6:.0f} GB/s algorithmic')
