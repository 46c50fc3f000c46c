"""tools/fuzz_probe.py <seed> [method]: one record set of tests/test_gpu_fuzz.py through every launch shape of cd_ekf / cd_eks against the C port, per trial."""
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
want = port.filter(port.F_CD_EKF, dg, None, H, Xi, m0, P0, dt, ys)
print('B', B, 'T', T, 'dt', dt, 'Xi', Xi)
print('params', params)
for name, fl in (('default', 0), ('wave', 0x2), ('wave generic', 0x12), ('lane', 0x4), ('lane generic', 0x14), ('dpp', 0x2 | 0x80)):
    try:
        got = fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ys, flags=fl)
    except Exception as e:
        print(name, 'raised', str(e)[:100]); continue
    for b in range(B):
        e = np.abs(np.asarray(got[0])[b] - want[0][b]).max(axis=1) / np.abs(want[0][b]).max()
        first = int(np.argmax(e > 1e-6)) if (e > 1e-6).any() else -1
        print(f'{name:14s} trial {b}: max rel err {np.nanmax(e):.2e} first step above 1e-6: {first}; port |m| max {np.abs(want[0][b]).max():.2e} finite {np.isfinite(want[0][b]).all()} got finite {np.isfinite(np.asarray(got[0])[b]).all()}; track {tracks[b,0]:.1f}..{tracks[b,:T].max():.1f}; ys finite {np.isfinite(ys[b]).all()}')
if len(sys.argv) > 2:
    b = int(sys.argv[2])
    gw = np.asarray(fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ys, flags=0x2)[0])[b]
    gd = np.asarray(fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ys, flags=0x82)[0])[b]
    w = want[0][b]
    for t in list(range(0, 40, 4)) + list(range(40, T, 40)):
        sc = np.abs(w[t]).max()
        print(f't {t:4d} |m| {sc:.3e} m2 {w[t,2]:.3e} P22 {want[1][b][t,2,2]:.3e} mfma err {np.abs(gw[t]-w[t]).max()/sc:.2e} dpp err {np.abs(gd[t]-w[t]).max()/sc:.2e}')
    gP = np.asarray(fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ys, flags=0x2)[1])[b]
    gN = np.asarray(fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ys, flags=0x2)[2])[b]
    wP, wN = want[1][b], want[2][b]
    print('step-by-step 78..130: |m| max, |P| max, P11, S-ish, errors of m, P, nll increments')
    for t in range(78, 131):
        print(f't {t} |m| {np.abs(w[t]).max():.3e} |P| {np.abs(wP[t]).max():.3e} P11 {wP[t,1,1]:.3e} P00 {wP[t,0,0]:.3e} y {ys[b,t]:.2f} m err {np.abs(gw[t]-w[t]).max()/np.abs(w[t]).max():.2e} P err {np.abs(gP[t]-wP[t]).max()/np.abs(wP[t]).max():.2e} nll {wN[t]:.6e} got {gN[t]:.6e}')
