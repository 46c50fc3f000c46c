"""tools/lane4_smoother_check.py: the one-lane-per-trial smoothers of cgp_lane4.hpp (eks, cd_eks; flags = CGP_THREAD_PER_TRIAL) against the C port
on the port's own filtering results: ragged batches, record lengths of every line phase, with and without whole quads."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chirpgp_amd import filters_smoothers as fs, tools
from chirpgp_amd.models import model_chirp, disc_chirp_lcd
from oracle import port
from tests import cases as cs
drift, disp, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
dg = copy.copy(drift); dg.gamma = disp.outer()
for B, T in ((1037, 500), (130, 506), (64, 16), (70, 14), (129, 49), (65, 2), (200, 7), (3, 1001), (257, 511), (100, 5), (64, 3)):
    _, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 11 + B, batch=B, states=False)
    ys = yss.cpu().numpy()
    f = port.filter(port.F_EKF, mc, None, H, 0.1, m0, P0, 0.01, ys)
    for name, got, want in (('eks', lambda: fs.eks(mc, f[0], f[1], 0.01, flags=0x4), lambda: port.smoother(port.S_EKS, mc, None, 0.01, f[0], f[1])),
                            ('cd_eks', lambda: fs.cd_eks(drift, disp, f[0], f[1], 0.01, flags=0x4), lambda: port.smoother(port.S_CD_EKS, dg, None, 0.01, f[0], f[1]))):
        g, w = got(), want()
        for a, b, n in zip(g, w, ('mss', 'Pss')):
            cs.assert_close(a, b, 1e-8, f'{name} B={B} T={T} {n}')
            assert cs.max_rel_err(a, b) <= 1e-9, (name, B, T, n, cs.max_rel_err(a, b))
        assert np.array_equal(g[0][:, -1], f[0][:, -1]) and np.array_equal(g[1][:, -1], f[1][:, -1])
    print(f'smoother parity B={B} T={T}: ok', flush=True)
