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


# ------------------------------------------------------------------------------------------------ round 6: host side of the exact gradient, refusals
def test_tangent_directions_are_the_derivatives_of_the_model_constants():
    """mle.tangent_directions (complex step, what cgp_ekf_nll_grad propagates) against fourth-order central differences of the same constants
    along theta = g_inv(params): both builders with an exact gradient, lam > 0 and the lam = 0 branch's own derivative away from 0."""
    from chirpgp_amd import mle
    rng = np.random.default_rng(5)
    for build, P in ((pm.build_chirp_model, 6), (pm.build_lascala_model, 4)):
        consts = mle._constants_of(build)
        assert consts is not None
        n = P or 6
        thetas = rng.uniform(-1.0, 1.5, (3, n))
        dt, Xi = 1e-3, 0.1
        got = mle.tangent_directions(build, thetas, dt, Xi)
        assert got.shape == (3, n, 24) and np.isfinite(got).all()
        for k in range(n):
            def at(e):
                th = thetas.copy(); th[:, k] += e
                return np.real(consts(pm.g(th).T.astype(np.complex128), dt, Xi)).T            # (G, 24)
            h = 1e-4
            fd = (-at(2 * h) + 8 * at(h) - 8 * at(-h) + at(-2 * h)) / (12 * h)
            # (the quotient carries eps |c| / h of rounding on constants that do not depend on theta_k at all)
            # ... and eps / h absolutely where the constant is a difference of O(1) terms (the Matern covariance's dt^3 entry, as in the reference)
            tol = 1e-7 * np.abs(fd).max(axis=0) + 1e-9 * np.abs(at(0.0)).max(axis=0) + 1e-11
            assert (np.abs(got[:, k, :] - fd) <= tol).all(), (build.__name__, k, np.abs(got[:, k, :] - fd).max(axis=0), tol)
    assert mle.has_exact_gradient('ekf', pm.build_chirp_model, 0.1)
    assert not mle.has_exact_gradient('sgp', pm.build_chirp_model, 0.1)
    assert not mle.has_exact_gradient('ekf', pm.build_harmonic_chirp_model, 0.1)
    assert not mle.has_exact_gradient('ekf', pm.build_chirp_model, np.array([0.1, 0.2]))       # a per-trial Xi: the difference form
    assert mle._exact_by_default('ekf', pm.build_chirp_model, 0.1, mle.EXACT_FROM_RECORDS, {})
    assert not mle._exact_by_default('ekf', pm.build_chirp_model, 0.1, mle.EXACT_FROM_RECORDS - 1, {})


def test_plain_callables_and_unknown_keywords_are_refused_by_name():
    """No CPU fallback: a Python closure is a TypeError before anything touches the GPU, and so is a keyword a runtime-compiled model does not take."""
    import pytest
    from chirpgp_amd import filters_smoothers as fs
    H, m0, P0, ys = np.eye(4)[0], np.zeros(4), np.eye(4), np.zeros(10)
    with pytest.raises(TypeError, match='descriptor'):
        fs.ekf(lambda u, dt: (u, np.eye(4)), H, 0.1, m0, P0, 1e-3, ys)
    with pytest.raises(TypeError, match='drift'):
        fs.cd_ekf(lambda u: u, lambda u: np.eye(4), H, 0.1, m0, P0, 1e-3, ys)
    with pytest.raises(TypeError, match='SigmaPoints'):
        fs.sgp_filter(pm.disc_chirp_lcd(0.1, 0.1, 1.0, 1.0), 'gh3', H, 0.1, m0, P0, 1e-3, ys)
    with pytest.raises(ValueError, match='d = 3'):
        fs.sgp_filter(pm.disc_chirp_lcd(0.1, 0.1, 1.0, 1.0), SigmaPoints.gauss_hermite(3, 3), H, 0.1, m0, P0, 1e-3, ys)
    with pytest.raises(TypeError, match='time_split'):
        fs._custom_kw(dict(time_split=(2, 64)))
    assert fs._custom_kw(dict(flags=0), ('flags',)) == dict(flags=0)


def test_gauss_hermite_rule_of_the_fused_expectation():
    """_engine._gh_rule(order) is the reference's 1-D rule (quadratures.py:156-196): sqrt(2) x the physicists' roots, weights / sqrt(pi)."""
    from chirpgp_amd import _engine
    for order in (3, 10, 32):
        xi, w = _engine._gh_rule(order)
        x, ww = np.polynomial.hermite.hermgauss(order)
        # (the reference finds the roots with np.roots: 3e-15 at order 10, 7e-10 at 32 -- restated, not improved)
        npt.assert_allclose(np.sort(xi), np.sqrt(2) * x, rtol=0, atol=1e-14 if order <= 10 else 5e-9)
        npt.assert_allclose(w[np.argsort(xi)], ww / np.sqrt(np.pi), rtol=0, atol=1e-14 if order <= 10 else 1e-11)
        assert abs(w.sum() - 1) < 1e-12
