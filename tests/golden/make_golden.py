"""Generates tests/golden/*.npz: small input / expected-output vectors for every function of the path.

The expected outputs come from the NumPy oracle (oracle/np_filters.py) AFTER it passed the reference's own tests
(tests/test_oracle_*.py).  The reference itself cannot be run in this pipeline (no JAX), so these
are oracle outputs, not reference outputs -- see oracle/__init__.py "How it is pinned".

    python -m tests.golden.make_golden
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import np_filters as nf          # noqa: E402
from tests import cases as cs                # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def pairs(c, cd_T):
    o_s = cs.osig(c.sgps)
    out = {}
    f = nf.ekf(c.o_disc, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    out['ekf'] = f
    out['eks'] = nf.eks(c.o_disc, f[0], f[1], c.dt)
    f = nf.sgp_filter(c.o_disc, o_s, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    out['sgp_filter'] = f
    out['sgp_smoother'] = nf.sgp_smoother(c.o_disc, o_s, f[0], f[1], c.dt)
    ys = c.ys[:cd_T]
    f = nf.cd_ekf(c.o_drift, c.o_disp, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    out['cd_ekf'] = f
    out['cd_eks'] = nf.cd_eks(c.o_drift, c.o_disp, f[0], f[1], c.dt)
    f = nf.cd_sgp_filter(c.o_drift, c.o_disp(None), o_s, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    out['cd_sgp_filter'] = f
    out['cd_sgp_smoother'] = nf.cd_sgp_smoother(c.o_drift, c.o_disp(None), o_s, f[0], f[1], c.dt)
    return out


def save(name, c, res, extra=None):
    flat = {'ys': c.ys, 'dt': c.dt, 'Xi': c.Xi, 'm0': c.m0, 'P0': c.P0}
    if getattr(c, 'H', None) is not None:
        flat['H'] = c.H
    for k, v in res.items():
        for i, a in enumerate(v):
            flat[f'{k}.{i}'] = a
    flat.update(extra or {})
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **flat)
    print(name, {k: v[0].shape for k, v in res.items()})


def main():
    c = cs.linear_case(0, T=200)
    res = pairs(c, 200)
    res['kf'] = nf.kf(c.F, c.Sigma, c.H, c.Xi, c.m0, c.P0, c.ys)
    res['rts'] = nf.rts(c.F, c.Sigma, res['kf'][0], res['kf'][1])
    save('linear_ou', c, res, {'F': c.F, 'Sigma': c.Sigma})

    c = cs.chirp_case(T=200, seed=21)
    save('chirp_gh3', c, pairs(c, 100), {'params': np.array([0.1, 0.1, 0.1, 1., 1., 7.])})

    c = cs.harmonic_case(T=120, seed=22, nh=3)
    save('harmonic3_cubature', c, pairs(c, 60), {'params': np.array([0.1, 0.1, 0.1, 1., 1., 7.])})

    c = cs.lascala_case(T=120, seed=23)
    save('lascala_gh3', c, pairs(c, 60), {'params': np.array([0.1, 1., 1., 7.])})

    c = cs.kpt_case(T=200, seed=24)
    res = {'ekf_for_kpt': nf.ekf_for_kpt(c.F, c.Sigma, c.o_h, c.Xi, c.m0, c.P0, c.dt, c.ys)}
    save('kpt2', c, res, {'F': c.F, 'Sigma': c.Sigma, 'params': np.array([0.5, 1e-4, 0.1, 8., 1.]), 'fs': 1000.})


if __name__ == '__main__':
    main()
