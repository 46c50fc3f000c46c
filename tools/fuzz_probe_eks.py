"""tools/fuzz_probe_eks.py <seed>: cd_eks of one record set of tests/test_gpu_fuzz.py (port's cd_ekf rows), every launch shape against the C port, per trial."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_fuzz import make_set
from chirpgp_amd import filters_smoothers as fs, models as pm
from oracle import port
seed = int(sys.argv[1])
B, T, dt, Xi, params, ys, tracks = make_set(seed)
T = min(T, 800); ys = np.ascontiguousarray(ys[:, :T])
drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
dg = copy.copy(drift); dg.gamma = disp.outer()
f = port.filter(port.F_CD_EKF, dg, None, H, Xi, m0, P0, dt, ys)
want = port.smoother(port.S_CD_EKS, dg, None, dt, f[0], f[1])
print('B', B, 'T', T, 'dt', dt)
for name, fl in (('wave', 0x2), ('wave generic', 0x12), ('lane', 0x4), ('dpp', 0x82)):
    got = fs.cd_eks(drift, disp, f[0], f[1], dt, flags=fl)
    for b in range(B):
        g, w = np.asarray(got[0])[b], want[0][b]
        gP, wP = np.asarray(got[1])[b], want[1][b]
        fin_w, fin_g = np.isfinite(w).all(axis=1) & np.isfinite(wP).all(axis=(1, 2)), np.isfinite(g).all(axis=1) & np.isfinite(gP).all(axis=(1, 2))
        both = fin_w & fin_g
        e = np.abs(g[both] - w[both]).max() / max(np.abs(w[both]).max(), 1e-300) if both.any() else float('nan')
        first_w = int(np.argmin(fin_w[::-1])) if not fin_w.all() else -1
        first_g = int(np.argmin(fin_g[::-1])) if not fin_g.all() else -1
        Pd = f[1][b][:, np.arange(4), np.arange(4)]
        print(f'{name:13s} trial {b}: rel err on rows finite in both {e:.2e}; port non-finite rows {int((~fin_w).sum())} (first, counted from the end: {first_w}), kernel {int((~fin_g).sum())} ({first_g}); '
              f'filter rows: min diag P {Pd.min():.2e} max |m| {np.abs(f[0][b]).max():.2e}; smoothed max |m| {np.nanmax(np.abs(w)):.2e} min diag Ps {np.nanmin(wP[:, np.arange(4), np.arange(4)]):.2e}')
