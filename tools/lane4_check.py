"""tools/lane4_check.py [method]: the large-batch lane-per-trial kernels (cgp_lane4.hpp) against the C port on a few thousand trials
(ragged batch, record lengths with and without a tail block, every output combination), then timed at the CRLB job's shape
(262 144 x 500) beside the round-4 lane kernel (flags = CGP_THREAD_PER_TRIAL | CGP_GENERIC_KERNEL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chirpgp_amd import filters_smoothers as fs, tools, _engine, quadratures
from chirpgp_amd.models import model_chirp, disc_chirp_lcd
from oracle import port
from tests import cases as cs

method = sys.argv[1] if len(sys.argv) > 1 else 'ekf'
big = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
_, _, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
sg = quadratures.SigmaPoints.gauss_hermite(4, 3) if method == 'sgp' else None
OLD = _engine.THREAD_PER_TRIAL | _engine.GENERIC_KERNEL
NEW = _engine.THREAD_PER_TRIAL


def run(ys, flags, want, **kw):
    if method == 'ekf':
        return fs.ekf(mc, H, 0.1, m0, P0, 0.01, ys, flags=flags, want=want, **kw)
    return fs.sgp_filter(mc, sg, H, 0.1, m0, P0, 0.01, ys, flags=flags, want=want, **kw)


worst = 0.0
for B, T in ((1000 + 37, 500), (130, 506), (64, 16), (70, 14), (129, 48)):
    _, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666, batch=B, states=False)
    ref = port.filter(port.F_EKF if method == 'ekf' else port.F_SGP, mc, sg, H, 0.1, m0, P0, 0.01, yss.cpu().numpy())
    for want in ((True, True, True), (True, False, False), (False, True, False), (False, False, True), (True, False, True)):
        out = run(yss, NEW, want)
        for o, r, name in zip(out, ref, ('mfs', 'Pfs', 'nll')):
            if o is not None:
                cs.assert_close(o.cpu().numpy(), r, 1e-9, f'{method} B={B} T={T} want={want} {name}')
                worst = max(worst, float(np.max(np.abs(o.cpu().numpy() - r) / (np.abs(r) + 1e-3 * np.abs(r).max()))))
    fin = run(yss, NEW, (False, False, True), nll_final_only=True)[2]
    cs.assert_close(fin.cpu().numpy(), ref[2][:, -1], 1e-9, 'final nll')
    print(f'parity B={B} T={T}: ok', flush=True)
print(f'worst scaled error vs port {worst:.2e}')

B, T = big, 500
_, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666, batch=B, states=False)
for name, want, nb in (('full', (True, True, True), 176), ('means', (True, False, False), 40)):
    for tag, flags in (('r04 lane kernel', OLD), ('lane4', NEW)):
        r = run(yss, flags, want)
        torch.cuda.synchronize()
        ev = _engine.kernel_events = []
        for _ in range(5):
            r = run(yss, flags, want)
        torch.cuda.synchronize()
        _engine.kernel_events = None
        ms = min(a.elapsed_time(b) for _, a, b in ev)
        print(f'{method} {B}x{T} {name:5s} {tag:16s}: {ms:7.3f} ms  {nb * B * T / ms / 1e9:6.2f} TB/s algorithmic', flush=True)
        del r
