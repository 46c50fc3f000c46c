"""tools/ab_check.py: the library now installed against the C port on 48 bench records (EKF, full T): worst relative errors and the
regime counters -- run by tools/ab.sh for every variant, so a timing is never read off a kernel that computes something else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs, _engine
from oracle import port
from tests import cases as cs
B, T = 48, 10000
kind = sys.argv[1] if len(sys.argv) > 1 else 'ekf'
wl = bench.make_workload(B, T, seed=0, kind=kind)
a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'])
if wl['kind'] == 'ekf':
    _engine.debug_set(_engine.DBG_COUNT_REGIMES, 1)
    got = fs.ekf(wl['disc'], *a, wl['ys'])
    rg = _engine.debug_counters()
    want = port.filter(port.F_EKF, wl['disc'], None, *a, wl['ys'])
else:
    got = fs.sgp_filter(wl['disc'], wl['sgps'], *a, wl['ys'])
    rg = {}
    want = port.filter(port.F_SGP, wl['disc'], wl['sgps'], *a, wl['ys'])
print('   check vs port:', ' '.join(f'{n} {cs.max_rel_err(g, w):.1e}' for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll'))), rg)
