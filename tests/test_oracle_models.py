"""Pins the oracle's model / quadrature / tools restatements with the reference's own tests.

* test/test_quadratures.py:19-59
* test/test_models.py:20-127 (the TME-dependent cases need the un-vendored `tme` package and are left out)
* test/test_m32.py:14-30
* test/test_utils.py:15-34, 84-106 (closed forms)
"""
import math
import numpy as np
import numpy.testing as npt
import pytest
import scipy.linalg
import scipy.special

from oracle import np_models as md
from oracle import np_tools as tl
from oracle.np_quadratures import SigmaPoints, gaussian_expectation


# ------------------------------------------------------------------ test_quadratures.py
cub = SigmaPoints.cubature(d=1)
gh = SigmaPoints.gauss_hermite(d=1, order=5)


def test_normalise():
    npt.assert_almost_equal(np.sum(cub.w), 1.)
    npt.assert_almost_equal(np.sum(gh.w), 1.)
    for d, order in ((4, 3), (3, 4), (2, 5)):
        sg = SigmaPoints.gauss_hermite(d, order)
        assert sg.xi.shape == (order ** d, d) and sg.n_points == order ** d
        npt.assert_allclose(np.sum(sg.w), 1., rtol=1e-13)


def test_integrating_polynomial():
    c1, c2, d = 0.1, 2., 1
    f = lambda x: c1 * x + c2 * x ** 2
    m, P = 0.5 * np.ones(d), 0.2 * np.eye(d)
    truth = np.reshape(c1 * m + c2 * (P + np.outer(m, m)), (-1,))
    npt.assert_almost_equal(truth, cub.expectation_from_nodes(f, cub.gen_sigma_points(m, np.sqrt(P))))
    npt.assert_almost_equal(truth, gh.expectation_from_nodes(f, gh.gen_sigma_points(m, np.sqrt(P))))


def test_integrating_sine():
    d = 1
    m, P = math.pi / 2 * np.ones(d), np.eye(d)
    truth = np.reshape(np.sin(m) * np.exp(-P / 2), (-1,))
    npt.assert_allclose(truth, cub.expectation_from_nodes(np.sin, cub.gen_sigma_points(m, np.sqrt(P))), rtol=2e-1)
    npt.assert_almost_equal(truth, gh.expectation_from_nodes(np.sin, gh.gen_sigma_points(m, np.sqrt(P))), decimal=4)


def test_gauss_hermite_order3_closed_form():
    """order-3 nodes {0, +-sqrt(3)}, 1-D weights {2/3, 1/6, 1/6}; dimension 0 varies fastest."""
    sg = SigmaPoints.gauss_hermite(4, 3)
    assert sorted(np.round(np.unique(sg.xi), 12)) == sorted(np.round([-math.sqrt(3), 0., math.sqrt(3)], 12))
    npt.assert_allclose(sorted(np.unique(np.round(sg.w, 14))),
                        sorted(np.unique(np.round([(2 / 3) ** a * (1 / 6) ** (4 - a) for a in range(5)], 14))))
    assert sg.xi[0, 0] != sg.xi[1, 0] and sg.xi[0, 1] == sg.xi[1, 1]
    hx, hw = np.polynomial.hermite.hermgauss(3)
    npt.assert_allclose(sorted(np.unique(sg.xi)), sorted(math.sqrt(2) * hx), atol=1e-14)


def test_g_expectation():
    """test_utils.py:84-95 with a NumPy stream: E[exp(V)] = exp(m + P/2)."""
    rng = np.random.default_rng(111)
    ms = rng.standard_normal((100, 1))
    Ps = rng.uniform(0.1, 1., (100, 1, 1))
    npt.assert_allclose(gaussian_expectation(ms, np.sqrt(Ps), func=np.exp, d=1, order=10), np.exp(ms + Ps[:, 0] / 2))


# ------------------------------------------------------------------ test_models.py / test_m32.py / test_utils.py
def test_g():
    x = np.random.default_rng(666).standard_normal(20)
    npt.assert_allclose(x, md.g_inv(md.g(x)), rtol=1e-14, atol=0)


@pytest.mark.parametrize('lam', [0.1, 1.])
@pytest.mark.parametrize('b', [0.1, 1.])
@pytest.mark.parametrize('ell', [0.1, 1.])
def test_chirp_models(lam, b, ell):
    """test_models.py:26-51: LCD mean == expm(drift matrix dt); LCD cov == Van Loan."""
    sigma, delta, dt = 0.1, 0.1, 0.1
    drift, dispersion, m0, P0, H = md.model_chirp(lam, b, ell, sigma, delta)
    m_and_cov = md.disc_chirp_lcd(lam, b, ell, sigma)
    drift_matrix, lcd_matrix = np.zeros((4, 4)), np.zeros((4, 4))
    for i in range(4):
        u = np.zeros(4)
        u[i] = 1.
        drift_matrix[:, i] = drift(u)
        lcd_matrix[:, i] = m_and_cov(u, dt)[0]
    npt.assert_allclose(scipy.linalg.expm(drift_matrix * dt), lcd_matrix)
    u = np.random.default_rng(1).standard_normal(4)
    F, Sigma = tl.lti_sde_to_disc(drift_matrix, dispersion(u), dt)
    npt.assert_allclose(Sigma, m_and_cov(u, dt)[1], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('num_harmonics', [1, 2, 3])
def test_harmonic_chirp_models(num_harmonics):
    """test_models.py:53-78."""
    lam, b, ell, sigma, delta, dt = 1., 1., 1., 0.1, 0.1, 0.1
    drift, dispersion, m0, P0, H = md.model_harmonic_chirp(lam, b, ell, sigma, delta, num_harmonics)
    m_and_cov = md.disc_harmonic_chirp_lcd(lam, b, ell, sigma, num_harmonics)
    dim = num_harmonics * 2 + 2
    drift_matrix, lcd_matrix = np.zeros((dim, dim)), np.zeros((dim, dim))
    for i in range(dim):
        u = np.zeros(dim)
        u[i] = 1.
        drift_matrix[:, i] = drift(u)
        lcd_matrix[:, i] = m_and_cov(u, dt)[0]
    npt.assert_allclose(scipy.linalg.expm(drift_matrix * dt), lcd_matrix)
    u = np.random.default_rng(2).standard_normal(dim)
    F, Sigma = tl.lti_sde_to_disc(drift_matrix, dispersion(u), dt)
    npt.assert_allclose(Sigma, m_and_cov(u, dt)[1], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('lam', [0.1, 1.])
@pytest.mark.parametrize('b', [0.1, 1.])
def test_lcd_chirp_cond_v(lam, b):
    """test_models.py:80-90."""
    u, dt = np.random.default_rng(3).standard_normal(4), 0.1
    full = md.disc_chirp_lcd(lam, b, 1., 1.)(u, dt)
    cond = md.disc_chirp_lcd_cond_v(lam, b)(u[:2], u[2], dt)
    npt.assert_allclose(full[0][:2], cond[0])
    npt.assert_allclose(full[1][:2, :2], cond[1])


def test_disc_m32():
    """test_models.py:107-116."""
    ell, sigma, dt = 1.1, 2.2, 1e-2
    u = np.random.default_rng(4).standard_normal(4)
    a, b = md.disc_m32(ell, sigma), md.disc_chirp_lcd(1., 1., ell, sigma)
    npt.assert_allclose(a(u[2:], dt)[0], b(u, dt)[0][2:])
    npt.assert_allclose(a(u[2:], dt)[1], b(u, dt)[1][2:, 2:])


@pytest.mark.parametrize('ell', [0.2, 1.])
@pytest.mark.parametrize('sigma', [0.2, 1.])
def test_disc_model_lascala_lcd(ell, sigma):
    """test_models.py:118-127: the lam = 0 branch; means coincide with the La Scala model."""
    u, dt = np.random.default_rng(5).standard_normal(4), 1e-2
    m1 = md.disc_model_lascala_lcd(ell, sigma)(u, dt)
    m2 = md.disc_chirp_lcd(0., 1., ell, sigma)(u, dt)
    npt.assert_allclose(m1[0], m2[0])
    npt.assert_allclose(m2[1][:2, :2], np.eye(2) * dt)


@pytest.mark.parametrize("ell, sigma, dt", [(0.1, 1., 0.1), (1., 2., 0.1), (0.456, 1.234, 0.789), (0.1789, 11.234, 0.0789)])
def test_m32(ell, sigma, dt):
    """test_m32.py:14-30."""
    lam = math.sqrt(3) / ell
    q = 4 * sigma ** 2 * lam ** 3
    A = np.array([[0., 1.], [-lam ** 2, -2 * lam]])
    Bv = np.array([0., math.sqrt(q)])
    correct_mean, correct_cov = tl.lti_sde_to_disc(A, Bv, dt)
    mean, cov = md.m32_solution(ell, sigma, dt)
    npt.assert_allclose(mean, correct_mean, atol=1e-12)
    npt.assert_allclose(cov, correct_cov, atol=1e-12)


@pytest.mark.parametrize("lam, f, dt", [(0.1, 1., 0.1), (1., 2., 0.1), (0., 0.1, 0.1), (0., 0.5, 0.1), (0., 0.1, 1.), (0., 0.1, 2.)])
def test_lti_disc(lam, f, dt):
    """test_utils.py:15-34."""
    A = np.array([[-lam, -2 * math.pi * f], [2 * math.pi * f, -lam]])
    F, Q = tl.lti_sde_to_disc(A, np.eye(2), dt)
    z = 2 * math.pi * dt * f
    expected_F = np.array([[math.cos(z), -math.sin(z)], [math.sin(z), math.cos(z)]]) * math.exp(-dt * lam)
    expected_Q = np.eye(2) * dt if lam == 0 else np.eye(2) * (1 - math.exp(-2 * dt * lam)) / (2 * lam)
    npt.assert_allclose(F, expected_F, atol=1e-15)
    npt.assert_allclose(Q, expected_Q, atol=1e-12)


@pytest.mark.parametrize('reduce_sum', [True, False])
def test_rmse(reduce_sum):
    """test_utils.py:97-106."""
    x1 = np.array([[1., 2., 3.], [4., 5., 6.]])
    x2 = np.array([[0., 1., 2.], [3., 4., 5.]])
    npt.assert_allclose(tl.rmse(x1, x2, reduce_sum), 3. if reduce_sum else np.ones(3))


def test_toy_chirp_phase_is_integral_of_frequency():
    """test_toymodels.py:45-55 idea: d phase / dt == freq (finite difference), also across a tiling seam."""
    ts, freq, phase = tl.tiled_meow(7000, dt=1e-3, offset=8.)
    fd = np.gradient(phase, ts)
    npt.assert_allclose(fd[5:-5], freq[5:-5], rtol=2e-3, atol=2e-3)
    assert np.all(np.abs(np.diff(phase)[3135:3150] - 8e-3) < 1e-6)     # continuous over the seam at k = 3141


def test_kpt_model_shapes():
    F, Sigma, m0, P0, h = md.build_kpt_chirp_model([0.1, 0.2, 0.3, 100., 1.], fs=1000., num_harmonics=2)
    assert F.shape == (4, 4) and Sigma.shape == (4, 4) and m0.shape == (4,)
    assert F[-1, 0] == 1. and Sigma[-1, -1] == 0.
    x = np.array([0.3, 1., 0.5, 0.1])
    npt.assert_allclose(h(x), 1. * math.sin(md.g(0.4)) + 0.5 * math.sin(2 * md.g(0.4)))
