"""Smoothers with selected outputs (cgp_smoother_select; SURVEY 8f-2 "fuse as an epilogue"): the smoothed marginal of one state component
-- mean, variance, E[f(V)] by 1-D Gauss-Hermite (quadratures.py:234-274; demos/ekfs_mle.py:69-77) -- written by the smoother kernel itself,
with or without the full rows.  Checked against the full rows of the same library, against the oracle's gaussian_expectation applied to
the ORACLE smoother's rows, and through the raw C-ABI."""
import ctypes as C

import numpy as np
import pytest

from tests import cases as cs

pytestmark = pytest.mark.gpu

WAVE, LANE, GENERIC, TIME_SPLIT, NO_TIME_SPLIT = 0x2, 0x4, 0x10, 0x800, 0x1000
ALL = dict(mean=True, var=True, expect='softplus')


def _oracle_expect(mean, var, func=None):
    from oracle import np_quadratures as oq
    from oracle.np_models import g
    with np.errstate(invalid='ignore'):
        return oq.gaussian_expectation(np.asarray(mean).reshape(-1), np.sqrt(np.asarray(var).reshape(-1)), func=func or g,
                                       force_shape=True)[:, 0].reshape(np.shape(mean))


def _check(sel, mss, Pss, k, tol=1e-13, what=''):
    cs.assert_close(sel['mean'], mss[..., k], tol, what + ' mean')
    cs.assert_close(sel['var'], Pss[..., k, k], tol, what + ' var')
    cs.assert_close(sel['expect'], _oracle_expect(mss[..., k], Pss[..., k, k]), 1e-12, what + ' expect')


def _filtered(c, B, seed, method='ekf'):
    from oracle import port
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(seed).standard_normal((B, c.ys.size))
    if method == 'ekf':
        return port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    return port.filter(port.F_SGP, c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)


@pytest.mark.parametrize('T', [2, 37, 64, 65, 449, 1300])
@pytest.mark.parametrize('flags', [WAVE | NO_TIME_SPLIT, WAVE | TIME_SPLIT])
def test_eks_walk_selection_against_full_rows_and_the_oracle(T, flags):
    """d = 4, one wavefront per trial (cgp_walk4.hpp), whole-record and time-split forms, ragged T (one row, one tile, a wrapped last tile)."""
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    c = cs.chirp_case(T=T, seed=51)
    f = _filtered(c, 5, T)
    want = port.smoother(port.S_EKS, c.disc, None, c.dt, f[0], f[1])
    full = fs.eks(c.disc, f[0], f[1], c.dt, flags=flags)
    for k in (2, 0):
        mss, Pss, sel = fs.eks(c.disc, f[0], f[1], c.dt, flags=flags, select=dict(comp=k, **ALL))
        for g, w in zip((mss, Pss), full):
            cs.assert_close(g, w, 1e-13, 'full rows beside the selection')
        _check(sel, full[0], full[1], k, what=f'T={T} k={k}')
        _check(sel, want[0], want[1], k, tol=1e-9, what=f'vs the oracle smoother T={T} k={k}')     # the whole chain against the oracle
        n0, n1, only = fs.eks(c.disc, f[0], f[1], c.dt, flags=flags, want=(False, False), select=dict(comp=k, **ALL))
        assert n0 is None and n1 is None
        for key in ('mean', 'var', 'expect'):
            np.testing.assert_array_equal(only[key], sel[key])                                      # the same with nothing else written


def test_single_record_and_negative_component_index():
    """(T, d) in -> (T,) out; comp = -2 is the frequency state of every chirp model (demos/_pipeline.py)."""
    from chirpgp_amd import filters_smoothers as fs
    c = cs.chirp_case(T=300, seed=52)
    f = _filtered(c, 1, 3)
    mss, Pss = fs.eks(c.disc, f[0][0], f[1][0], c.dt)
    _, _, sel = fs.eks(c.disc, f[0][0], f[1][0], c.dt, want=(False, False), select=dict(comp=-2, expect='softplus'))
    assert set(sel) == {'expect'} and sel['expect'].shape == (300,)
    cs.assert_close(sel['expect'], _oracle_expect(mss[:, 2], Pss[:, 2, 2]), 1e-12, 'expect')
    assert sel.mean is None                                  # attribute access of what was not asked for


@pytest.mark.parametrize('method', ['rts', 'sgp_smoother', 'lascala_eks'])
def test_the_other_d4_walks(method):
    from chirpgp_amd import filters_smoothers as fs, models as pm
    T, B = 700, 4
    if method == 'rts':
        import bench
        c = cs.chirp_case(T=T, seed=53)
        F, Sigma = bench.frozen_frequency_linear_model(np.array([0.1, 0.1, 0.1, 1., 1., 7.]), c.dt)
        from oracle import port
        ys = c.ys[None, :] + 0.05 * np.random.default_rng(1).standard_normal((B, T))
        f = port.filter(port.F_EKF, pm.linear_cond_m_cov(F, Sigma), None, c.H, c.Xi, c.m0, c.P0, 0., ys)
        run = lambda **kw: fs.rts(F, Sigma, f[0], f[1], **kw)
    elif method == 'sgp_smoother':
        c = cs.chirp_case(T=T, seed=54)
        f = _filtered(c, B, 2, 'sgp')
        run = lambda **kw: fs.sgp_smoother(c.disc, c.sgps, f[0], f[1], c.dt, **kw)
    else:
        c = cs.lascala_case(T=T, seed=55)
        f = _filtered(c, B, 3)
        run = lambda **kw: fs.eks(c.disc, f[0], f[1], c.dt, **kw)
    full = run()
    _, _, sel = run(want=(False, False), select=dict(comp=2, **ALL))
    _check(sel, full[0], full[1], 2, what=method)


@pytest.mark.parametrize('T,B', [(500, 200), (501, 70), (38, 130), (16, 64), (2, 3), (1000, 257)])
def test_one_lane_per_trial_eks_selection(T, B):
    """The large-batch kernel (cgp_lane4.hpp): selected outputs leave as whole 128-byte lines of the [B][T] arrays -- every line phase
    (T = 500: four, T = 501: sixteen), ragged last wavefront, records shorter than a line."""
    from chirpgp_amd import filters_smoothers as fs
    c = cs.chirp_case(T=T, seed=56)
    f = _filtered(c, B, T + B)
    full = fs.eks(c.disc, f[0], f[1], c.dt, flags=LANE)
    for sel_kw in (ALL, dict(expect='softplus'), dict(mean=True, var=True)):
        mss, Pss, sel = fs.eks(c.disc, f[0], f[1], c.dt, flags=LANE, select=dict(comp=2, **sel_kw))
        for g, w in zip((mss, Pss), full):
            cs.assert_close(g, w, 1e-13, 'full rows beside the selection')
        _, _, only = fs.eks(c.disc, f[0], f[1], c.dt, flags=LANE, want=(False, False), select=dict(comp=2, **sel_kw))
        for key in sel_kw:
            np.testing.assert_array_equal(only[key], sel[key])
        if 'mean' in sel_kw:
            cs.assert_close(sel['mean'], full[0][..., 2], 1e-13, 'mean')
            cs.assert_close(sel['var'], full[1][..., 2, 2], 1e-13, 'var')
        if 'expect' in sel_kw:
            cs.assert_close(sel['expect'], _oracle_expect(full[0][..., 2], full[1][..., 2, 2]), 1e-12, 'expect')


def test_one_lane_per_trial_cd_eks_selection():
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    import copy
    c = cs.chirp_case(T=240, seed=57)
    dg = copy.copy(c.drift)
    dg.gamma = c.disp.outer()
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(5).standard_normal((150, c.ys.size))
    f = port.filter(port.F_CD_EKF, dg, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    full = fs.cd_eks(c.drift, c.disp, f[0], f[1], c.dt, flags=LANE)
    _, _, sel = fs.cd_eks(c.drift, c.disp, f[0], f[1], c.dt, flags=LANE, want=(False, False), select=dict(comp=2, **ALL))
    _check(sel, full[0], full[1], 2, what='cd_eks lane')
    for k in (1, 3):
        _, _, sel = fs.cd_eks(c.drift, c.disp, f[0], f[1], c.dt, flags=LANE, want=(False, False), select=dict(comp=k, mean=True, var=True))
        cs.assert_close(sel['mean'], full[0][..., k], 1e-13, 'mean')
        cs.assert_close(sel['var'], full[1][..., k, k], 1e-13, 'var')


@pytest.mark.parametrize('nh', [2, 3])
@pytest.mark.parametrize('flags', [WAVE | NO_TIME_SPLIT, WAVE | TIME_SPLIT])
def test_tile_layout_smoothers_selection(nh, flags):
    """d = 6 / 8 (cgp_coop8.hpp), BASELINE C5's smoother: eks and the cubature sgp_smoother, whole-record and time-split."""
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    c = cs.harmonic_case(T=520, seed=58, nh=nh)
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(6).standard_normal((3, c.ys.size))
    f = port.filter(port.F_SGP, c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    k = c.d - 2
    for run in (lambda **kw: fs.sgp_smoother(c.disc, c.sgps, f[0], f[1], c.dt, flags=flags, **kw),
                lambda **kw: fs.eks(c.disc, f[0], f[1], c.dt, flags=flags, **kw)):
        full = run()
        mss, Pss, sel = run(select=dict(comp=-2, **ALL))
        for g, w in zip((mss, Pss), full):
            cs.assert_close(g, w, 1e-13, 'full rows beside the selection')
        _check(sel, full[0], full[1], k, what=f'harmonic {nh}')
        _, _, only = run(want=(False, False), select=dict(comp=-2, **ALL))
        for key in ALL:
            np.testing.assert_array_equal(only[key], sel[key])


def test_kernels_without_the_epilogue_gather_from_their_rows():
    """cd_sgp_smoother, the generic kernels, d = 3: the selection is gathered from the full rows by a second launch (temporary rows where
    the caller wants none) -- same numbers, no error."""
    from chirpgp_amd import filters_smoothers as fs
    c = cs.chirp_case(T=150, seed=59)
    f = _filtered(c, 3, 9)
    full = fs.cd_sgp_smoother(c.drift, c.disp(None), c.sgps, f[0], f[1], c.dt)
    _, _, sel = fs.cd_sgp_smoother(c.drift, c.disp(None), c.sgps, f[0], f[1], c.dt, want=(False, False), select=dict(comp=2, **ALL))
    _check(sel, full[0], full[1], 2, what='cd_sgp_smoother')
    full = fs.eks(c.disc, f[0], f[1], c.dt, flags=GENERIC | WAVE)
    mss, _, sel = fs.eks(c.disc, f[0], f[1], c.dt, flags=GENERIC | WAVE, want=(True, False), select=dict(comp=2, **ALL))
    _check(sel, full[0], full[1], 2, what='generic eks')
    cs.assert_close(mss, full[0], 1e-13, 'mss')
    lin = cs.linear_case(0, T=200)
    from oracle import port
    fl = port.filter(port.F_EKF, lin.disc, None, lin.H, lin.Xi, lin.m0, lin.P0, lin.dt, lin.ys[None, :])
    full = fs.rts(lin.F, lin.Sigma, fl[0], fl[1])
    _, _, sel = fs.rts(lin.F, lin.Sigma, fl[0], fl[1], want=(False, False), select=dict(comp=1, mean=True, var=True, expect='square'))
    cs.assert_close(sel['mean'], full[0][..., 1], 1e-13, 'd = 3 mean')
    cs.assert_close(sel['expect'], full[0][..., 1] ** 2 + full[1][..., 1, 1], 1e-12, 'E[V^2] = m^2 + P')


@pytest.mark.parametrize('func', ['exp', 'identity', 'square'])
def test_the_other_integrands(func):
    from chirpgp_amd import filters_smoothers as fs
    c = cs.chirp_case(T=260, seed=60)
    f = _filtered(c, 4, 11)
    full = fs.eks(c.disc, f[0], f[1], c.dt)
    m, v = full[0][..., 2] * 0.1, full[1][..., 2, 2]
    for flags in (WAVE, LANE):
        _, _, sel = fs.eks(c.disc, f[0], f[1], c.dt, flags=flags, want=(False, False), select=dict(comp=2, mean=True, var=True, expect=func, order=7))
        m, v = sel['mean'], sel['var']
        exact = {'exp': np.exp(m + v / 2), 'identity': m, 'square': m * m + v}[func]
        # Gauss-Hermite of order 7 is exact for polynomials up to degree 13; exp with variances ~1e-2: far below 1e-10
        cs.assert_close(sel['expect'], exact, 1e-10, func)


def test_nan_variance_and_bad_arguments():
    """A negative smoothed variance gives NaN in `expect` (sqrt in the reference's call), NaN inputs flow through; argument errors are
    errors of the call, by name."""
    from chirpgp_amd import filters_smoothers as fs, _engine
    c = cs.chirp_case(T=200, seed=61)
    f = _filtered(c, 3, 12)
    f[1][1, 100, 2, 2] = np.nan
    for flags in (WAVE, LANE):
        full = fs.eks(c.disc, f[0], f[1], c.dt, flags=flags)
        _, _, sel = fs.eks(c.disc, f[0], f[1], c.dt, flags=flags, want=(False, False), select=dict(comp=2, **ALL))
        np.testing.assert_array_equal(np.isnan(sel['expect']), np.isnan(full[1][..., 2, 2]) | np.isnan(full[0][..., 2]))
        assert np.isnan(sel['expect'][1, :101]).all() and np.isfinite(sel['expect'][1, 101:]).all() and np.isfinite(sel['expect'][0]).all()
    with pytest.raises(ValueError, match='outside the state dimension'):
        fs.eks(c.disc, f[0], f[1], c.dt, select=dict(comp=4, mean=True))
    with pytest.raises(ValueError, match='expect must be one of'):
        fs.eks(c.disc, f[0], f[1], c.dt, select=dict(comp=2, expect=lambda x: x))
    with pytest.raises(ValueError, match='none of'):
        fs.eks(c.disc, f[0], f[1], c.dt, select=dict(comp=2))
    with pytest.raises(ValueError, match='writes both'):
        fs.eks(c.disc, f[0], f[1], c.dt, want=(True, False))


def test_raw_c_abi():
    """cgp_smoother_select as a C caller sees it: the struct of include/chirpgp_hip.h, NULL rows, error codes and messages."""
    import torch
    from chirpgp_amd import _engine as E, filters_smoothers as fs
    lib = E.load_library()
    ctx = E.context()
    c = cs.chirp_case(T=300, seed=62)
    f = _filtered(c, 6, 13)
    m, P = torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()
    keep = []
    model = E._model_struct(c.disc, None, 6, keep)
    xi, w = E._gh_rule(10)
    xi_d, w_d = torch.from_numpy(xi).cuda(), torch.from_numpy(w).cuda()
    out = torch.full((3, 6, 300), -1.0, dtype=torch.float64, device='cuda')
    o = E.CgpSmoothOut()
    o.comp, o.func, o.order = 2, 0, 10
    o.comp_mean, o.comp_var, o.expect = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
    o.xi, o.w = xi_d.data_ptr(), w_d.data_ptr()
    st = E._stream()
    rc = lib.cgp_smoother_select(ctx, E.S_EKS, C.byref(model), None, c.dt, m.data_ptr(), P.data_ptr(), 6, 300, C.byref(o), 0, st)
    assert rc == 0, lib.cgp_last_error(ctx)
    torch.cuda.synchronize()
    full = fs.eks(c.disc, f[0], f[1], c.dt)
    cs.assert_close(out[0].cpu().numpy(), full[0][..., 2], 1e-13, 'mean')
    cs.assert_close(out[2].cpu().numpy(), _oracle_expect(full[0][..., 2], full[1][..., 2, 2]), 1e-12, 'expect')
    o.comp = 7
    assert lib.cgp_smoother_select(ctx, E.S_EKS, C.byref(model), None, c.dt, m.data_ptr(), P.data_ptr(), 6, 300, C.byref(o), 0, st) == -1
    assert b'comp' in lib.cgp_last_error(ctx)
    o.comp, o.order = 2, 0
    assert lib.cgp_smoother_select(ctx, E.S_EKS, C.byref(model), None, c.dt, m.data_ptr(), P.data_ptr(), 6, 300, C.byref(o), 0, st) == -1
    assert b'order' in lib.cgp_last_error(ctx)
    # a kernel without the epilogue and no rows to gather from: refused, by name
    o.order = 10
    rc = lib.cgp_smoother_select(ctx, E.S_EKS, C.byref(model), None, c.dt, m.data_ptr(), P.data_ptr(), 6, 300, C.byref(o), GENERIC | WAVE, st)
    assert rc == -2 and b'mss and Pss' in lib.cgp_last_error(ctx)
    assert lib.cgp_smoother_select(ctx, E.S_EKS, C.byref(model), None, c.dt, m.data_ptr(), P.data_ptr(), 6, 300, None, 0, st) == -1


def test_one_step_records_and_empty_batches():
    """T = 1: the smoother returns the filtering row (filters_smoothers.py:140-142) and the selection is that row's marginal, in every
    launch shape; an empty batch returns empty selections."""
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    c = cs.chirp_case(T=1, seed=3)
    for B in (1, 3, 70):
        ys = np.zeros((B, 1)) + c.ys[None, :]
        f = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
        for flags in (WAVE, LANE, WAVE | TIME_SPLIT):
            mss, Pss, sel = fs.eks(c.disc, f[0], f[1], c.dt, flags=flags, select=dict(comp=2, **ALL))
            np.testing.assert_array_equal(mss, f[0])
            np.testing.assert_array_equal(sel['mean'], f[0][..., 2])
            np.testing.assert_array_equal(sel['var'], f[1][..., 2, 2])
            cs.assert_close(sel['expect'], _oracle_expect(f[0][..., 2], f[1][..., 2, 2]), 1e-12, 'T = 1 expect')
    e = fs.eks(c.disc, np.zeros((0, 5, 4)), np.zeros((0, 5, 4, 4)), c.dt, want=(False, False), select=dict(comp=2, mean=True))
    assert e[0] is None and e[2]['mean'].shape == (0, 5)
