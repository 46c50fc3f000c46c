"""CPU tests of the host-side helpers that sit either side of the hot path."""
import numpy as np
import numpy.testing as npt

from chirpgp_amd import models as pm, results, toymodels, tools
from chirpgp_amd.quadratures import SigmaPoints
from oracle import np_models as om, np_quadratures as oq, np_tools as ot


def test_result_file_layout(tmp_path):
    d = str(tmp_path)
    for mc in range(3):
        results.save_result(d, 'ekfs_mle', 'const', mc, np.zeros(5), np.ones(5), np.nan if mc == 1 else 0.5 + mc)
    z = np.load(results.result_path(d, 'ekfs_mle', 'const', 0))
    assert sorted(z.files) == ['rmse', 'smoothing_cov', 'smoothing_mean']
    mean, std, nans = results.load_rmse_table(d, 'ekfs_mle', ['const'], 3)['const']
    assert nans == 1 and abs(mean - 1.5) < 1e-12


def test_descriptors_evaluate_like_the_oracle_models():
    rng = np.random.default_rng(0)
    u = rng.standard_normal(4)
    npt.assert_allclose(pm.disc_chirp_lcd(0.3, 0.2, 0.7, 1.3)(u, 1e-2)[0], om.disc_chirp_lcd(0.3, 0.2, 0.7, 1.3)(u, 1e-2)[0], rtol=1e-14)
    npt.assert_allclose(pm.disc_chirp_lcd(0., 0.2, 0.7, 1.3)(u, 1e-2)[1], om.disc_chirp_lcd(0., 0.2, 0.7, 1.3)(u, 1e-2)[1], rtol=1e-14)
    u8 = rng.standard_normal(8)
    a = pm.disc_harmonic_chirp_lcd(0.3, 0.2, 0.7, 1.3, 3, 1.5)(u8, 1e-2)
    b = om.disc_harmonic_chirp_lcd(0.3, 0.2, 0.7, 1.3, 3, 1.5)(u8, 1e-2)
    npt.assert_allclose(a[0], b[0], rtol=1e-13)
    npt.assert_allclose(a[1], b[1], rtol=1e-13)
    npt.assert_allclose(pm.model_harmonic_chirp(0.3, 0.2, 0.7, 1.3, 0.1, 3, 1.5)[0](u8),
                        om.model_harmonic_chirp(0.3, 0.2, 0.7, 1.3, 0.1, 3, 1.5)[0](u8), rtol=1e-13)
    npt.assert_allclose(pm.disc_model_lascala_lcd(0.7, 1.3)(u, 1e-2)[0], om.disc_model_lascala_lcd(0.7, 1.3)(u, 1e-2)[0], rtol=1e-14)
    for build_p, build_o, args in ((pm.build_chirp_model, om.build_chirp_model, ()), (pm.build_harmonic_chirp_model, om.build_harmonic_chirp_model, (2,))):
        p = np.array([0.1, 0.2, 0.3, 1.1, 0.9, 7.])
        rp, ro = build_p(p, *args), build_o(p, *args)
        for x, y in zip(rp[3:], ro[3:]):
            npt.assert_allclose(x, y, rtol=1e-14)
        npt.assert_allclose(rp[1](None), ro[1](None), rtol=1e-14)
    # batched builders: one row per trial
    P = np.array([0.1, 0.2, 0.3, 1.1, 0.9, 7.]) * np.linspace(0.8, 1.2, 5)[:, None]
    _, disp, disc, m0, P0, H = pm.build_chirp_model(P)
    assert disc.params.shape == (5, 5) and m0.shape == (5, 4) and P0.shape == (5, 4, 4) and disp.outer().shape == (5, 4, 4)
    npt.assert_allclose(P0[3], om.build_chirp_model(P[3])[4], rtol=1e-14)


def test_sigma_points_match_the_oracle():
    for d, order in ((1, 5), (3, 4), (4, 3)):
        a, b = SigmaPoints.gauss_hermite(d, order), oq.SigmaPoints.gauss_hermite(d, order)
        npt.assert_allclose(a.xi, b.xi, atol=1e-14)
        npt.assert_allclose(a.w, b.w, rtol=1e-13)
    a, b = SigmaPoints.cubature(8), oq.SigmaPoints.cubature(8)
    npt.assert_array_equal(a.xi, b.xi)
    npt.assert_array_equal(a.w, b.w)


def test_toymodels_and_tools_match_the_oracle():
    ts = np.linspace(1e-3, 3.0, 3000)
    f1, p1 = toymodels.meow_freq(offset=8.)
    f2, p2 = ot.meow_freq(offset=8.)
    npt.assert_array_equal(f1(ts), f2(ts))
    npt.assert_array_equal(p1(ts), p2(ts))
    npt.assert_array_equal(toymodels.gen_chirp(ts, toymodels.constant_mag(1.), p1), ot.gen_chirp(ts, ot.constant_mag(1.), p2))
    A = np.array([[0., 1.], [-3., -2 * np.sqrt(3)]])
    for x, y in zip(tools.lti_sde_to_disc(A, np.array([0., 2.]), 0.1), ot.lti_sde_to_disc(A, np.array([0., 2.]), 0.1)):
        npt.assert_allclose(x, y, rtol=1e-13)
    npt.assert_allclose(tools.rmse(np.ones((4, 2)), np.zeros((4, 2))), 2.0)


def test_rk4_pair_helpers_match_the_oracle():
    """quadratures.rk4_m_cov / rk4_m_cov_backward (reference names) against the oracle's restatement on a linear ODE, and
    against the exact solution of dm = A m, dP = A P + P A^T + Q to fourth order."""
    from chirpgp_amd.quadratures import rk4_m_cov, rk4_m_cov_backward
    A = np.array([[0., 1.], [-2., -0.3]])
    Q = np.array([[0., 0.], [0., 0.5]])
    ode = lambda m, P: (A @ m, A @ P + P @ A.T + Q)
    m, P = np.array([1., -1.]), np.array([[0.3, 0.1], [0.1, 0.2]])
    a = rk4_m_cov(ode, m, P, 0.01)
    b = oq.rk4_m_cov(ode, m, P, 0.01)
    npt.assert_allclose(a[0], b[0], rtol=1e-15)
    npt.assert_allclose(a[1], b[1], rtol=1e-15)
    import scipy.linalg
    npt.assert_allclose(a[0], scipy.linalg.expm(A * 0.01) @ m, rtol=1e-9)
    back = lambda m_, P_, mf, Pf: (A @ m_ + 0.1 * (m_ - mf), A @ P_ + P_ @ A.T - 0.2 * Pf)
    a = rk4_m_cov_backward(back, m, P, m * 0.9, P * 1.1, -0.01)
    b = oq.rk4_m_cov_backward(back, m, P, m * 0.9, P * 1.1, -0.01)
    npt.assert_allclose(a[0], b[0], rtol=1e-15)
    npt.assert_allclose(a[1], b[1], rtol=1e-15)


def test_remaining_reference_model_names():
    """models.disc_chirp_lcd_cond_v (host utility, models.py:313-330), disc_chirp_euler_maruyama, disc_chirp_tme."""
    import pytest
    u = np.array([0.3, -0.2])
    for lam in (0.2, 0.0):
        a = pm.disc_chirp_lcd_cond_v(lam, 0.4)(u, 1.7, 1e-2)
        b = om.disc_chirp_lcd_cond_v(lam, 0.4)(u, 1.7, 1e-2)
        npt.assert_allclose(a[0], b[0], rtol=1e-14)
        npt.assert_allclose(a[1], b[1], rtol=1e-14)
    assert pm.disc_chirp_euler_maruyama() is NotImplemented
    with pytest.raises(NotImplementedError):
        pm.disc_chirp_tme(0.1, 0.1, 1., 1.)
