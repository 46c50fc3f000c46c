"""Runs a test case through one implementation of the path and returns {function name: outputs}.

backend 'hip'   chirpgp_amd.filters_smoothers  (public functions -> ctypes -> C-ABI -> HIP kernels)
backend 'port'  oracle/c/port.c                (checker)
backend 'numpy' oracle/np_filters.py           (checker; slow)
"""
import copy
import numpy as np

from oracle import np_filters as nf
from oracle import port
from tests import cases as cs


def _with_gamma(drift, gamma):
    d = copy.copy(drift)
    d.gamma = gamma
    return d


def run_pairs(backend, c, cd_T=None, hip_kw=None, only=None):
    hip_kw = hip_kw or {}
    ys_cd = c.ys[..., :cd_T] if cd_T else c.ys
    out = {}

    def want(*names):
        return only is None or any(n in only for n in names)

    if backend == 'hip':
        from chirpgp_amd import filters_smoothers as fs
        if want('ekf', 'eks'):
            out['ekf'] = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys, **hip_kw)
            out['eks'] = fs.eks(c.disc, out['ekf'][0], out['ekf'][1], c.dt, **hip_kw)
        if want('sgp_filter', 'sgp_smoother'):
            out['sgp_filter'] = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys, **hip_kw)
            out['sgp_smoother'] = fs.sgp_smoother(c.disc, c.sgps, out['sgp_filter'][0], out['sgp_filter'][1], c.dt, **hip_kw)
        if want('cd_ekf', 'cd_eks'):
            out['cd_ekf'] = fs.cd_ekf(c.drift, c.disp, c.H, c.Xi, c.m0, c.P0, c.dt, ys_cd, **hip_kw)
            out['cd_eks'] = fs.cd_eks(c.drift, c.disp, out['cd_ekf'][0], out['cd_ekf'][1], c.dt, **hip_kw)
        if want('cd_sgp_filter', 'cd_sgp_smoother'):
            out['cd_sgp_filter'] = fs.cd_sgp_filter(c.drift, c.disp(None), c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys_cd, **hip_kw)
            out['cd_sgp_smoother'] = fs.cd_sgp_smoother(c.drift, c.disp(None), c.sgps, out['cd_sgp_filter'][0],
                                                        out['cd_sgp_filter'][1], c.dt, **hip_kw)
    elif backend == 'port':
        dg = _with_gamma(c.drift, c.disp.outer())
        if want('ekf', 'eks'):
            out['ekf'] = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
            out['eks'] = port.smoother(port.S_EKS, c.disc, None, c.dt, out['ekf'][0], out['ekf'][1])
        if want('sgp_filter', 'sgp_smoother'):
            out['sgp_filter'] = port.filter(port.F_SGP, c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
            out['sgp_smoother'] = port.smoother(port.S_SGP, c.disc, c.sgps, c.dt, out['sgp_filter'][0], out['sgp_filter'][1])
        if want('cd_ekf', 'cd_eks'):
            out['cd_ekf'] = port.filter(port.F_CD_EKF, dg, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys_cd)
            out['cd_eks'] = port.smoother(port.S_CD_EKS, dg, None, c.dt, out['cd_ekf'][0], out['cd_ekf'][1])
        if want('cd_sgp_filter', 'cd_sgp_smoother'):
            out['cd_sgp_filter'] = port.filter(port.F_CD_SGP, dg, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys_cd)
            out['cd_sgp_smoother'] = port.smoother(port.S_CD_SGP, dg, c.sgps, c.dt, out['cd_sgp_filter'][0], out['cd_sgp_filter'][1])
    else:
        o_s = cs.osig(c.sgps)
        if want('ekf', 'eks'):
            out['ekf'] = nf.ekf(c.o_disc, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
            out['eks'] = nf.eks(c.o_disc, out['ekf'][0], out['ekf'][1], c.dt)
        if want('sgp_filter', 'sgp_smoother'):
            out['sgp_filter'] = nf.sgp_filter(c.o_disc, o_s, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
            out['sgp_smoother'] = nf.sgp_smoother(c.o_disc, o_s, out['sgp_filter'][0], out['sgp_filter'][1], c.dt)
        if want('cd_ekf', 'cd_eks'):
            out['cd_ekf'] = nf.cd_ekf(c.o_drift, c.o_disp, c.H, c.Xi, c.m0, c.P0, c.dt, ys_cd)
            out['cd_eks'] = nf.cd_eks(c.o_drift, c.o_disp, out['cd_ekf'][0], out['cd_ekf'][1], c.dt)
        if want('cd_sgp_filter', 'cd_sgp_smoother'):
            out['cd_sgp_filter'] = nf.cd_sgp_filter(c.o_drift, c.o_disp(None), o_s, c.H, c.Xi, c.m0, c.P0, c.dt, ys_cd)
            out['cd_sgp_smoother'] = nf.cd_sgp_smoother(c.o_drift, c.o_disp(None), o_s, out['cd_sgp_filter'][0],
                                                        out['cd_sgp_filter'][1], c.dt)
    return out


def smoothers_on(backend, c, filt, hip_kw=None):
    """Smoothers evaluated on GIVEN filtering results (so that a smoother is compared on identical inputs)."""
    hip_kw = hip_kw or {}
    out = {}
    if backend == 'hip':
        from chirpgp_amd import filters_smoothers as fs
        if 'ekf' in filt:
            out['eks'] = fs.eks(c.disc, filt['ekf'][0], filt['ekf'][1], c.dt, **hip_kw)
        if 'sgp_filter' in filt:
            out['sgp_smoother'] = fs.sgp_smoother(c.disc, c.sgps, filt['sgp_filter'][0], filt['sgp_filter'][1], c.dt, **hip_kw)
        if 'cd_ekf' in filt:
            out['cd_eks'] = fs.cd_eks(c.drift, c.disp, filt['cd_ekf'][0], filt['cd_ekf'][1], c.dt, **hip_kw)
        if 'cd_sgp_filter' in filt:
            out['cd_sgp_smoother'] = fs.cd_sgp_smoother(c.drift, c.disp(None), c.sgps, filt['cd_sgp_filter'][0],
                                                        filt['cd_sgp_filter'][1], c.dt, **hip_kw)
    return out


def compare(got, want, rtol, what):
    worst = {}
    for k in want:
        if k not in got:
            continue
        for i, (g, w) in enumerate(zip(got[k], want[k])):
            cs.assert_close(g, w, rtol, f'{what}.{k}[{i}]')
            worst[f'{k}[{i}]'] = cs.max_rel_err(g, w)
    return worst


def load_golden(name):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'))
    res = {}
    for k in z.files:
        if '.' in k:
            fn, i = k.split('.')
            res.setdefault(fn, {})[int(i)] = z[k]
    return z, {fn: tuple(v[i] for i in sorted(v)) for fn, v in res.items()}
