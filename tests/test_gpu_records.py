"""Shared measurement records (include/chirpgp_hip.h, cgp_filter: ys_stride / ys_repeat / ys_index), the parameter-grid sweep
of BASELINE config C5 as one tested workflow (grid -> arg-min -> filter + smoother at the arg-min), and the MLE objective of
the KPT and La Scala jobs against the oracle (tetralith/jobs/kpt_mle.py:41-44, lascala_ekfs_mle.py:40-43,
lascala_ghfs_mle.py:43-46, harmonic_kpt_mle.py:44-47).  Reference for the sharing: one `ys` under value_and_grad /
a sweep, demos/ekfs_mle.py:43-48, demos/ghfs_harmonics_mle.py:50-64."""
import ctypes as C
import os
import sys

import numpy as np
import numpy.testing as npt
import pytest

from tests import cases as cs
from tests import mle_oracle as mo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'demos'))

WAVE, THREAD, X4 = dict(flags=0x2), dict(flags=0x4), dict(flags=0x2 | 0x200)


def _param_rows(G, P, seed, base):
    rng = np.random.default_rng(seed)
    return np.asarray(base) * np.exp(0.3 * rng.standard_normal((G, P)))


@pytest.mark.parametrize('kw', [WAVE, THREAD, X4, {}], ids=['wave', 'lane', 'x4', 'default'])
def test_shared_record_equals_replicated_records_bitwise(kw):
    """G parameter vectors on ONE record: (T,) + trials_per_record = G gives exactly what G replicated records give."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    G, T = 13, 333
    c = cs.chirp_case(T=T, seed=3)
    _, _, disc, m0, P0, H = pm.build_chirp_model(_param_rows(G, 6, 1, [0.1, 0.1, 0.1, 1., 1., 7.]))
    dense = fs.ekf(disc, H, c.Xi, m0, P0, c.dt, np.tile(c.ys, (G, 1)), **kw)
    shared = fs.ekf(disc, H, c.Xi, m0, P0, c.dt, c.ys, trials_per_record=G, **kw)
    for a, b in zip(shared, dense):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('method', ['sgp_filter', 'cd_ekf', 'cd_sgp_filter', 'sgp_filter_d8', 'ekf_d8', 'kf', 'ekf_for_kpt'])
def test_records_per_group_and_index_against_the_port(method):
    """R records, k trials per record, a record_index that reorders and drops records: every kernel family against the C port
    run on the explicitly gathered records."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    import copy
    R, k, T = 4, 3, 150
    idx = np.array([2, 0, 3])
    n = idx.size * k
    rows = _param_rows(n, 6, 7, [0.1, 0.1, 0.1, 1., 1., 7.])
    nh = 3 if method.endswith('d8') else 1
    yss = np.stack([cs.chirp_measurements(T, 40 + r, num_harmonics=(nh if nh > 1 else 0))[2] for r in range(R)])
    gathered = np.repeat(yss[idx], k, axis=0)
    kw = dict(trials_per_record=k, record_index=idx)
    if method in ('sgp_filter', 'cd_ekf', 'cd_sgp_filter'):
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(rows)
        sg = SigmaPoints.gauss_hermite(4, 3)
        if method == 'sgp_filter':
            got = fs.sgp_filter(disc, sg, H, 0.1, m0, P0, 1e-3, yss, **kw)
            want = port.filter(port.F_SGP, disc, sg, H, 0.1, m0, P0, 1e-3, gathered)
        else:
            dg = copy.copy(drift)
            dg.gamma = disp.outer()
            if method == 'cd_ekf':
                got = fs.cd_ekf(drift, disp, H, 0.1, m0, P0, 1e-3, yss, **kw)
                want = port.filter(port.F_CD_EKF, dg, None, H, 0.1, m0, P0, 1e-3, gathered)
            else:
                got = fs.cd_sgp_filter(drift, disp, sg, H, 0.1, m0, P0, 1e-3, yss, **kw)
                want = port.filter(port.F_CD_SGP, dg, sg, H, 0.1, m0, P0, 1e-3, gathered)
    elif method.endswith('d8'):
        drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(rows, 3)
        sg = SigmaPoints.cubature(8)
        if method == 'sgp_filter_d8':
            got = fs.sgp_filter(disc, sg, H, 0.1, m0, P0, 1e-3, yss, **kw)
            want = port.filter(port.F_SGP, disc, sg, H, 0.1, m0, P0, 1e-3, gathered)
        else:
            got = fs.ekf(disc, H, 0.1, m0, P0, 1e-3, yss, **kw)
            want = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 1e-3, gathered)
    elif method == 'kf':
        c = cs.linear_case(0, T=T)
        yss = np.stack([c.ys * (1 + 0.1 * r) for r in range(R)])
        gathered = np.repeat(yss[idx], k, axis=0)
        got = fs.kf(c.F, c.Sigma, c.H, c.Xi, c.m0, c.P0, yss, **kw)
        want = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, 0., gathered)
    else:
        prm = _param_rows(n, 5, 9, [0.5, 1e-4, 0.1, 8., 1.])
        F, Sigma, m0, P0, h = pm.build_kpt_chirp_model(prm, 1000., 2)
        yss = np.stack([cs.chirp_measurements(T, 60 + r, num_harmonics=2)[2] for r in range(R)])
        gathered = np.repeat(yss[idx], k, axis=0)
        got = fs.ekf_for_kpt(F, Sigma, h, 0.1, m0, P0, 1e-3, yss, **kw)
        spec = copy.copy(pm.linear_cond_m_cov(F, Sigma))
        spec.model_id, spec.n_harm = port.M_KPT, 2
        want = port.filter(port.F_EKF_KPT, spec, None, None, 0.1, m0, P0, 1e-3, gathered)
    assert got[0].shape[0] == n
    for i, (a, b) in enumerate(zip(got, want)):
        cs.assert_close(a, b, 1e-7, f'{method}[{i}]')


def test_c_abi_stride_zero_and_argument_checks():
    """Raw C-ABI: ys_stride = 0 shares one record between all trials; a negative stride or ys_repeat < 1 is CGP_E_ARG."""
    import torch
    from chirpgp_amd import _engine as E, models as pm
    lib, ctx = E.load_library(), E.context()
    B, T, d = 5, 100, 4
    c = cs.chirp_case(T=T, seed=8)
    ys = torch.from_numpy(c.ys).cuda()
    prm = torch.from_numpy(np.ascontiguousarray(c.disc.params)).cuda()
    H, Xi, m0, P0 = (torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))).cuda() for x in (c.H, c.Xi, c.m0, c.P0))
    model = E.CgpModel(E.M_HARMONIC_LCD, d, 1, 5, prm.data_ptr(), 0, None, 0)
    init = E.CgpInit(H.data_ptr(), 0, Xi.data_ptr(), 0, m0.data_ptr(), 0, P0.data_ptr(), 0)
    mfs = torch.empty((B, T, d), dtype=torch.float64, device='cuda')
    Pfs = torch.empty((B, T, d, d), dtype=torch.float64, device='cuda')
    nll = torch.empty((B, T), dtype=torch.float64, device='cuda')
    tail = (B, T, mfs.data_ptr(), Pfs.data_ptr(), nll.data_ptr(), 0, None)
    assert lib.cgp_filter(ctx, E.F_EKF, C.byref(model), None, C.byref(init), c.dt, ys.data_ptr(), 0, 1, None, *tail) == 0
    torch.cuda.synchronize()
    from chirpgp_amd import filters_smoothers as fs
    one = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    for b in range(B):
        assert np.array_equal(mfs[b].cpu().numpy(), one[0]) and np.array_equal(nll[b].cpu().numpy(), one[2])
    assert lib.cgp_filter(ctx, E.F_EKF, C.byref(model), None, C.byref(init), c.dt, ys.data_ptr(), -1, 1, None, *tail) == -1
    assert b'ys_stride' in lib.cgp_last_error(ctx)
    assert lib.cgp_filter(ctx, E.F_EKF, C.byref(model), None, C.byref(init), c.dt, ys.data_ptr(), T, 0, None, *tail) == -1
    assert b'ys_repeat' in lib.cgp_last_error(ctx)
    with pytest.raises(ValueError):
        fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys, trials_per_record=2, record_index=[1])


def test_mle_reads_the_record_in_place():
    """mle.batched_nll / fit_many hand the engine the records themselves: no expand / repeat_interleave / gather on the way
    (chirpgp_amd/mle.py is free of torch calls), and the result equals the replicated-record evaluation."""
    import inspect
    from chirpgp_amd import mle, models as pm
    src = inspect.getsource(mle)
    for banned in ('repeat_interleave', '.expand(', 'torch_index', 'np.repeat(yss', 'broadcast_to(ys'):
        assert banned not in src, banned
    T = 400
    ys = cs.chirp_measurements(T, 5)[2]
    th = pm.g_inv(_param_rows(13, 6, 2, [0.1, 0.1, 0.1, 1., 1., 7.]))
    a = mle.batched_nll('ekf', pm.build_chirp_model, th, ys, 0.1, 1e-3)
    b = mo.nll('ekf', pm.build_chirp_model, th, ys, 0.1, 1e-3)
    npt.assert_allclose(a, b, rtol=1e-9)


def test_harmonic_grid_sweep_workflow_against_the_port(tmp_path):
    """BASELINE C5's workflow end to end: G = 16 grid points x 3 records of the 3-harmonic model (d = 8, cubature)
    in one launch -> arg-min per record -> full sigma-point filter + smoother at the arg-min -> result files; the NLL matrix,
    the arg-min and the smoothing results against oracle/c/port.c."""
    import harmonic_sweep as hs
    from chirpgp_amd import models as pm, results
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    R, T, nh = 3, 600, 3
    grid = hs.parameter_grid(2)
    assert grid.shape == (16, 6)
    yss = np.stack([cs.chirp_measurements(T, 90 + r, num_harmonics=nh)[2] for r in range(R)])
    out = hs.sweep_and_smooth(yss, grid, 0.1, 1e-3, nh)
    sg = SigmaPoints.cubature(8)
    eff = pm.g(pm.g_inv(grid))
    _, _, disc, m0, P0, H = pm.build_harmonic_chirp_model(np.tile(eff, (R, 1)), nh)
    want = port.filter(port.F_SGP, disc, sg, H, 0.1, m0, P0, 1e-3, np.repeat(yss, 16, axis=0), nll_final_only=True)[2].reshape(R, 16)
    cs.assert_close(out['nll'], want, 1e-8, 'sweep nll')
    assert np.array_equal(out['argmin'], np.argmin(np.where(np.isfinite(want), want, np.inf), axis=1))
    _, _, disc_b, m0_b, P0_b, H_b = pm.build_harmonic_chirp_model(eff[out['argmin']], nh)
    f = port.filter(port.F_SGP, disc_b, sg, H_b, 0.1, m0_b, P0_b, 1e-3, yss)
    s = port.smoother(port.S_SGP, disc_b, sg, 1e-3, f[0], f[1])
    cs.assert_close(out['mss'], s[0], 1e-7, 'sweep mss')
    cs.assert_close(out['Pss'], s[1], 1e-7, 'sweep Pss')
    # the driver itself, with result files
    o2, errs = hs.main(['--records', '2', '--T', '500', '--points', '2', '--save', str(tmp_path)])
    assert np.isfinite(errs).all()
    z = np.load(results.result_path(str(tmp_path), 'harmonic_sweep', 'const', 1))
    assert set(z.files) == {'smoothing_mean', 'smoothing_cov', 'rmse'} and z['smoothing_mean'].shape == (500, 8)


KPT_INIT = np.array([0.02, 1e-5, 1e-5, 8., 1.])            # tetralith/jobs/kpt_mle.py:38
LASCALA_INIT = np.array([0.1, 1., 1., 7.])                # tetralith/jobs/lascala_ekfs_mle.py:37


def _record(T, seed, nh=0, dt=1e-3, Xi=0.1):
    from chirpgp_amd.toymodels import gen_chirp, gen_harmonic_chirp, meow_freq, constant_mag
    ts = np.linspace(dt, dt * T, T)
    _, phase = meow_freq(offset=8.)
    clean = gen_chirp(ts, constant_mag(1.), phase) if nh == 0 else gen_harmonic_chirp(ts, [constant_mag(1.)] * nh, phase)
    return clean + np.sqrt(Xi) * np.random.default_rng(seed).standard_normal(T)


MLE_CASES = [pytest.param('ekf_for_kpt', 'kpt', 1, 0, id='kpt_mle'), pytest.param('ekf_for_kpt', 'kpt', 3, 3, id='harmonic_kpt_mle'),
             pytest.param('ekf', 'lascala', 0, 0, id='lascala_ekfs_mle'), pytest.param('sgp_filter', 'lascala', 0, 0, id='lascala_ghfs_mle'),
             pytest.param('ekf', 'harmonic', 3, 3, id='harmonic_ekfs_mle')]


def _family(family, nh):
    from chirpgp_amd import models as pm
    if family == 'kpt':
        return pm.build_kpt_chirp_model, KPT_INIT, dict(fs=1000., num_harmonics=nh)
    if family == 'lascala':
        return pm.build_lascala_model, LASCALA_INIT, {}
    return pm.build_harmonic_chirp_model, np.array([0.1, 0.1, 0.1, 1., 1., 7.]), dict(num_harmonics=nh)


@pytest.mark.parametrize('method,family,nh,sig_h', MLE_CASES)
def test_job_objective_and_gradient_against_the_oracle(method, family, nh, sig_h):
    """The objectives of the five remaining reference jobs: NLL to 1e-9 relative, difference gradient to 1e-4 of its scale,
    at the job's start point and at a second point."""
    from chirpgp_amd import mle, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    build, init, bkw = _family(family, nh)
    sg = SigmaPoints.gauss_hermite(4, 3) if method == 'sgp_filter' else None
    ys = _record(1200, 321, sig_h)
    fun = mle.make_objective(method, build, ys, 0.1, 1e-3, sgps=sg, **bkw)
    for theta in (mo.g_inv(init), mo.g_inv(init * np.linspace(1.4, 0.8, init.size))):
        f, grad = fun(theta)
        f_o, grad_o = mo.value_and_grad(method, build, theta, ys, 0.1, 1e-3, sgps=sg, **bkw)
        npt.assert_allclose(f, f_o, rtol=1e-9)
        npt.assert_allclose(grad, grad_o, rtol=1e-4, atol=1e-4 * np.abs(grad_o).max())


@pytest.mark.parametrize('method,family,nh,sig_h', [MLE_CASES[0], MLE_CASES[2]])
def test_job_fit_reaches_the_oracle_optimum(method, family, nh, sig_h):
    """mle.fit on the KPT and La Scala objectives against SciPy L-BFGS-B on the port's objective, same start: the engine's
    optimum is at least as good (1e-6 relative), and the port agrees on the NLL at the engine's optimum."""
    from chirpgp_amd import mle, models as pm
    build, init, bkw = _family(family, nh)
    ys = _record(1500, 654, sig_h)
    opt, res = mle.fit(method, build, init, ys, 0.1, 1e-3, maxiter=300, **bkw)
    opt_o, res_o = mo.fit(method, build, init, ys, 0.1, 1e-3, **bkw)
    assert np.isfinite(res.fun) and res.fun <= res_o.fun + 1e-6 * abs(res_o.fun) + 1e-6 * max(1.0, abs(res.fun)), (res.fun, res_o.fun)
    npt.assert_allclose(mo.nll(method, build, pm.g_inv(opt), ys, 0.1, 1e-3, **bkw)[0], res.fun, rtol=1e-9)


@pytest.mark.parametrize('job', ['kpt_mle', 'harmonic_kpt_mle', 'lascala_ekfs_mle', 'lascala_ghfs_mle', 'harmonic_ekfs_mle'])
def test_job_counterparts_run_end_to_end(job, tmp_path):
    """demos/jobs.py: one Monte-Carlo run of each remaining reference job (short record, constant magnitude): the MLE lowers
    the NLL, the RMSE is finite and the result file has the reference's layout and name."""
    import jobs
    from chirpgp_amd import results
    rows = jobs.run_job(job, num_mcs=1, T=1000, results=str(tmp_path), maxiter=40, seed=5, mags=('const',), quiet=True)
    (mc, name, err, nll0, nll1), = rows
    assert np.isfinite(err) and nll1 < nll0
    z = np.load(results.result_path(str(tmp_path), job, 'const', 0))
    d = {'kpt_mle': 3, 'harmonic_kpt_mle': 5, 'harmonic_ekfs_mle': 8}.get(job, 4)
    assert set(z.files) == {'smoothing_mean', 'smoothing_cov', 'rmse'} and z['smoothing_mean'].shape == (1000, d)


@pytest.mark.parametrize('job', ['ekfs_mle', 'ghfs_mle', 'kpt_mle', 'harmonic_ekfs_mle', 'cd_ekfs_mle'])
def test_lockstep_job_matches_the_record_by_record_loop(job, tmp_path):
    """demos/jobs.py --lockstep: the Monte-Carlo runs of a magnitude law in ONE lock-step fit + one batched filter / smoother
    launch.  Same records as the record-by-record loop (same random stream); every run lowers its NLL; the optimum found is
    no worse than the loop's (scipy's L-BFGS-B) beyond a small slack; and the smoothing results saved ARE the batched
    smoother's at the lock-step parameters (checked against a single-record call at those parameters)."""
    import jobs
    import _pipeline
    from chirpgp_amd import results
    n, T = 3, 800
    kw = dict(num_mcs=n, T=T, maxiter=40, seed=11, mags=('const', 'damped'), quiet=True)
    rows_l = jobs.run_job(job, results=str(tmp_path / 'lock'), lockstep=True, **kw)
    rows_s = sorted(jobs.run_job(job, results=str(tmp_path / 'loop'), **kw))
    assert [(r[0], r[1]) for r in rows_l] == [(r[0], r[1]) for r in rows_s] and len(rows_l) == 2 * n
    for (mc, name, err_l, nll0_l, nll1_l), (_, _, err_s, nll0_s, nll1_s) in zip(rows_l, rows_s):
        npt.assert_allclose(nll0_l, nll0_s, rtol=1e-12)            # same record, same start point
        assert np.isfinite(err_l) and nll1_l < nll0_l
        assert nll1_l <= nll1_s + 0.02 * abs(nll0_s - nll1_s), (job, mc, name, nll1_l, nll1_s)
        z = np.load(results.result_path(str(tmp_path / 'lock'), job, name, mc))
        assert np.isfinite(z['smoothing_mean']).all() and z['smoothing_mean'].shape[0] == T
    # the batched launch with per-record parameters == the single-record launches at the same parameters
    method, family, model_h, signal_h, sg = jobs.JOBS[job]
    yss = np.stack([next(ys for _, nm, ys in _pipeline.records_of_run(11 + mc, T, 1e-3, 0.1, signal_h) if nm == 'damped') for mc in range(n)])
    sgps = sg() if sg else None
    r = _pipeline.run_records(method, yss, 0.1, 1e-3, sgps=sgps, num_harmonics=model_h, maxiter=5, family=family)
    build, _, bkw = _pipeline._family_setup(method, family, model_h, 1e-3, None)
    for i in range(n):
        mss, Pss, est = _pipeline._filter_and_smooth(method, build, bkw, r['opt_params'][i], sgps, 0.1, 1e-3, yss[i])
        npt.assert_allclose(r['mss'][i], mss, rtol=1e-10, atol=1e-12)
        npt.assert_allclose(r['Pss'][i], Pss, rtol=1e-10, atol=1e-12)
        npt.assert_allclose(r['est_freq'][i], est, rtol=1e-10, atol=1e-12)
