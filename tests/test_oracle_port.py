"""The plain-C restatement (oracle/c/port.c) against the NumPy oracle (oracle/np_filters.py).

The NumPy oracle is generic (callables + complex-step Jacobians); the C port has the enumerated models with
hand-derived Jacobians.  Agreement to ~1e-9 on every method x model pins the analytic Jacobians,
the model enumeration and the C port that the GPU parity tests and bench.py's cpu_baseline then use.
"""
import numpy as np
import pytest

from oracle import np_filters as nf
from oracle import port
from tests import cases as cs

RTOL = 1e-8


def _check_filter(got, want, what):
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        cs.assert_close(g, w, RTOL, f'{what}.{n}')


def _check_smoother(got, want, what):
    for g, w, n in zip(got, want, ('mss', 'Pss')):
        cs.assert_close(g, w, RTOL, f'{what}.{n}')


def _all_methods(c, cd_T=None):
    """Runs the 5 filter/smoother pairs of a case through both implementations and compares."""
    o_s = cs.osig(c.sgps)
    gam = c.disp.outer()
    drift_g = _with_gamma(c.drift, gam)
    # discrete EKF / EKS
    f_np = nf.ekf(c.o_disc, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    f_c = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    _check_filter(f_c, f_np, f'{c}.ekf')
    _check_smoother(port.smoother(port.S_EKS, c.disc, None, c.dt, f_np[0], f_np[1]),
                    nf.eks(c.o_disc, f_np[0], f_np[1], c.dt), f'{c}.eks')
    # sigma-point
    f_np = nf.sgp_filter(c.o_disc, o_s, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    f_c = port.filter(port.F_SGP, c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    _check_filter(f_c, f_np, f'{c}.sgp_filter')
    _check_smoother(port.smoother(port.S_SGP, c.disc, c.sgps, c.dt, f_np[0], f_np[1]),
                    nf.sgp_smoother(c.o_disc, o_s, f_np[0], f_np[1], c.dt), f'{c}.sgp_smoother')
    # continuous-discrete (shorter: 4 RK4 stages of Python per step)
    ys = c.ys[:cd_T] if cd_T else c.ys
    f_np = nf.cd_ekf(c.o_drift, c.o_disp, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    f_c = port.filter(port.F_CD_EKF, drift_g, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    _check_filter(f_c, f_np, f'{c}.cd_ekf')
    _check_smoother(port.smoother(port.S_CD_EKS, drift_g, None, c.dt, f_np[0], f_np[1]),
                    nf.cd_eks(c.o_drift, c.o_disp, f_np[0], f_np[1], c.dt), f'{c}.cd_eks')
    f_np = nf.cd_sgp_filter(c.o_drift, c.o_disp(None), o_s, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    f_c = port.filter(port.F_CD_SGP, drift_g, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    _check_filter(f_c, f_np, f'{c}.cd_sgp_filter')
    _check_smoother(port.smoother(port.S_CD_SGP, drift_g, c.sgps, c.dt, f_np[0], f_np[1]),
                    nf.cd_sgp_smoother(c.o_drift, c.o_disp(None), o_s, f_np[0], f_np[1], c.dt), f'{c}.cd_sgp_smoother')


def _with_gamma(drift, gamma):
    import copy
    d = copy.copy(drift)
    d.gamma = gamma
    return d


@pytest.mark.parametrize('idx', [0, 1])
def test_port_linear(idx):
    c = cs.linear_case(idx, T=300)
    _all_methods(c)
    # kf / rts are ekf / eks on the linear descriptor
    _check_filter(port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys),
                  nf.kf(c.F, c.Sigma, c.H, c.Xi, c.m0, c.P0, c.ys), 'kf')


def test_port_chirp():
    _all_methods(cs.chirp_case(T=250), cd_T=120)


def test_port_chirp_cubature_lam0():
    _all_methods(cs.chirp_case(T=150, sg='cub', params=(0., 0.3, 0.2, 0.5, 2., 6.)), cd_T=80)


def test_port_harmonic3():
    _all_methods(cs.harmonic_case(T=150, nh=3, freq_scale=1.5), cd_T=60)


def test_port_lascala():
    _all_methods(cs.lascala_case(T=150), cd_T=60)


def test_port_kpt():
    c = cs.kpt_case(T=300)
    import copy
    spec = cs.pm.linear_cond_m_cov(c.F, c.Sigma)
    spec = copy.copy(spec)
    spec.model_id, spec.n_harm = port.M_KPT, c.nh
    got = port.filter(port.F_EKF_KPT, spec, None, None, c.Xi, c.m0, c.P0, c.dt, c.ys)
    want = nf.ekf_for_kpt(c.F, c.Sigma, c.o_h, c.Xi, c.m0, c.P0, c.dt, c.ys)
    _check_filter(got, want, 'ekf_for_kpt')


def test_port_batched_and_per_trial_params():
    """Leading batch axis on ys and per-trial model parameters (the vmap of tetralith/jobs/crlb_ekf.py:68-72)."""
    B, T = 5, 120
    rng = np.random.default_rng(3)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * rng.uniform(0.7, 1.3, size=(B, 6))
    drift, disp, disc, m0, P0, H = cs.pm.build_chirp_model(params)
    ys = np.stack([cs.chirp_measurements(T, 100 + i)[2] for i in range(B)])
    mfs, Pfs, nll = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 1e-3, ys)
    nll_last = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 1e-3, ys, nll_final_only=True)[2]
    np.testing.assert_array_equal(nll_last, nll[:, -1])
    for i in (0, 3):
        _, _, o_disc, om0, oP0, oH = cs.om.build_chirp_model(params[i])
        want = nf.ekf(o_disc, oH, 0.1, om0, oP0, 1e-3, ys[i])
        _check_filter((mfs[i], Pfs[i], nll[i]), want, f'batched[{i}]')


def test_port_nan_semantics():
    """Indefinite P0 -> all-NaN from the first sigma-point step on, exactly where the NumPy oracle has NaN."""
    sg = cs.SigmaPoints.cubature(2)
    F, Sigma = 0.9 * np.eye(2), 0.01 * np.eye(2)
    P0 = np.array([[1., 2.], [2., 1.]])
    spec = cs.pm.linear_cond_m_cov(F, Sigma)
    got = port.filter(port.F_SGP, spec, sg, np.array([1., 0.]), 0.1, np.zeros(2), P0, 0.1, np.ones(5))
    want = nf.sgp_filter(lambda u, dt: (F @ u, Sigma), cs.osig(sg), np.array([1., 0.]), 0.1, np.zeros(2), P0, 0.1, np.ones(5))
    for g, w in zip(got, want):
        assert np.array_equal(np.isnan(g), np.isnan(w)) and np.all(np.isnan(g))


def test_native_timed_build_agrees_with_the_checker_build():
    """bench.py's cpu_baseline times `gcc -O3 -march=native -DFIXED_D=d -ffp-contract=fast` of the SAME source
    (oracle/port.py: build_native); only the rounding may differ from the checker build (contracted multiply-adds)."""
    from tests import backends as bk
    c = cs.chirp_case(T=400, seed=21)
    nat = port.native(4)
    assert nat.port_fixed_d() == 4
    got = bk.run_pairs('port', c, cd_T=120)
    dg = bk._with_gamma(c.drift, c.disp.outer())
    f = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys, use=nat)
    s = port.smoother(port.S_EKS, c.disc, None, c.dt, f[0], f[1], use=nat)
    f2 = port.filter(port.F_CD_SGP, dg, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys[:120], use=nat)
    for a, b in zip(f + s + f2, got['ekf'] + got['eks'] + got['cd_sgp_filter']):
        cs.assert_close(a, b, 1e-9, 'native build')
    # a fixed-dimension build refuses other dimensions instead of mis-indexing
    h = cs.harmonic_case(T=20)
    with pytest.raises(RuntimeError):
        port.filter(port.F_EKF, h.disc, None, h.H, h.Xi, h.m0, h.P0, h.dt, h.ys, use=nat)
    port.set_num_threads(1, nat)
    assert port.num_threads(nat) == 1
    port.set_num_threads(port.num_threads(), nat)
