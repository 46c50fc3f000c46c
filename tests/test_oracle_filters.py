"""Pins the NumPy oracle with the reference's own tests for the path, re-expressed without JAX.

* test/test_filters_smoothers.py:19-85  (all 10 exported functions, same seed-666 data, same tolerances)
* test/test_crlb.py:61-68               (batched kf: Pfs identical across trials; E[(mf-x)(mf-x)^T] ~ Pf)
* an independent check the reference does not have: KF / RTS against dense Gaussian conditioning.
"""
import math
import numpy as np
import numpy.testing as npt
import pytest

from oracle import np_filters as fs
from oracle import np_models as md
from oracle import np_tools as tl
from oracle.np_quadratures import SigmaPoints
from tests.refcases import linear_ou_cases


@pytest.mark.parametrize('idx', [0, 1])
def test_equivalence_on_linear_models(idx):
    """test_filters_smoothers.py:19-85."""
    c = linear_ou_cases()[idx]
    F, Sigma, H, Xi, m0, P0, dt, ys, A, B = (c[k] for k in ('F', 'Sigma', 'H', 'Xi', 'm0', 'P0', 'dt', 'ys', 'A', 'B'))

    drift = lambda u: A @ u
    dispersion = lambda _: B
    m_and_cov = lambda u, _: (F @ u, Sigma)

    kf_results = fs.kf(F, Sigma, H, Xi, m0, P0, ys)
    ekf_results = fs.ekf(m_and_cov, H, Xi, m0, P0, dt, ys)
    cd_ekf_results = fs.cd_ekf(drift, dispersion, H, Xi, m0, P0, dt, ys)
    sgps = SigmaPoints.gauss_hermite(d=c['dim_x'], order=4)
    ghkf_results = fs.sgp_filter(m_and_cov, sgps, H, Xi, m0, P0, dt, ys)
    cd_ghkf_results = fs.cd_sgp_filter(drift, B, sgps, H, Xi, m0, P0, dt, ys)

    for i in range(3):
        npt.assert_allclose(kf_results[i], ekf_results[i])
        npt.assert_allclose(kf_results[i], ghkf_results[i])
        npt.assert_allclose(kf_results[i], cd_ekf_results[i], rtol=1e-5)
        npt.assert_allclose(kf_results[i], cd_ghkf_results[i], rtol=1e-5)

    rts_results = fs.rts(F, Sigma, kf_results[0], kf_results[1])
    eks_results = fs.eks(m_and_cov, ekf_results[0], ekf_results[1], dt)
    cd_eks_results = fs.cd_eks(drift, dispersion, cd_ekf_results[0], cd_ekf_results[1], dt)
    ghks_results = fs.sgp_smoother(m_and_cov, sgps, ghkf_results[0], ghkf_results[1], dt)
    cd_ghks_results = fs.cd_sgp_smoother(drift, B, sgps, cd_ghkf_results[0], cd_ghkf_results[1], dt)

    for i in range(2):
        npt.assert_allclose(rts_results[i], eks_results[i])
        npt.assert_allclose(rts_results[i], ghks_results[i])
        npt.assert_allclose(rts_results[i], cd_eks_results[i], atol=1e-1)
        npt.assert_allclose(cd_eks_results[i], cd_ghks_results[i])

    # N5 of SURVEY.md: the last smoothing row is the last filtering row.
    npt.assert_array_equal(rts_results[0][-1], kf_results[0][-1])
    npt.assert_array_equal(rts_results[1][-1], kf_results[1][-1])


def test_kf_rts_against_dense_gaussian_conditioning():
    """Independent of any recursion: build the joint Gaussian of (x_1..x_T, y_1..y_T) for a small linear
    model and condition densely.  Filtering marginal k conditions on y_1..y_k, smoothing on y_1..y_T;
    the cumulative nll is -log N(y_1..y_k)."""
    rng = np.random.default_rng(7)
    d, T = 3, 12
    F = np.array([[0.9, 0.1, 0.], [-0.2, 0.8, 0.05], [0., 0.1, 0.95]])
    L = rng.standard_normal((d, d)) * 0.3
    Sigma = L @ L.T + 0.05 * np.eye(d)
    H = np.array([1., -0.5, 0.25])
    Xi = 0.3
    m0 = np.array([0.5, -1., 0.2])
    P0 = np.diag([0.4, 0.2, 0.1])
    ys = rng.standard_normal(T)

    mfs, Pfs, nll = fs.kf(F, Sigma, H, Xi, m0, P0, ys)
    mss, Pss = fs.rts(F, Sigma, mfs, Pfs)

    # Joint prior over x_1..x_T
    means, covs = [], {}
    m, P = m0, P0
    for k in range(T):
        m, P = F @ m, F @ P @ F.T + Sigma
        means.append(m)
        covs[(k, k)] = P
        for j in range(k):
            covs[(k, j)] = F @ covs[(k - 1, j)] if k - 1 > j else F @ covs[(j, j)]
    mx = np.concatenate(means)
    Cx = np.zeros((T * d, T * d))
    for (k, j), C in covs.items():
        Cx[k * d:(k + 1) * d, j * d:(j + 1) * d] = C
        Cx[j * d:(j + 1) * d, k * d:(k + 1) * d] = C.T
    Hbig = np.kron(np.eye(T), H[None, :])
    my, Cy, Cxy = Hbig @ mx, Hbig @ Cx @ Hbig.T + Xi * np.eye(T), Cx @ Hbig.T

    for k in range(T):
        n = k + 1
        gain = Cxy[:, :n] @ np.linalg.inv(Cy[:n, :n])
        post_m = mx + gain @ (ys[:n] - my[:n])
        post_C = Cx - gain @ Cxy[:, :n].T
        npt.assert_allclose(mfs[k], post_m[k * d:(k + 1) * d], rtol=1e-9, atol=1e-11)
        npt.assert_allclose(Pfs[k], post_C[k * d:(k + 1) * d, k * d:(k + 1) * d], rtol=1e-9, atol=1e-11)
        r = ys[:n] - my[:n]
        _, logdet = np.linalg.slogdet(Cy[:n, :n])
        npt.assert_allclose(nll[k], 0.5 * (n * math.log(2 * math.pi) + logdet + r @ np.linalg.solve(Cy[:n, :n], r)),
                            rtol=1e-10)
    gain = Cxy @ np.linalg.inv(Cy)
    post_m = mx + gain @ (ys - my)
    post_C = Cx - gain @ Cxy.T
    for k in range(T):
        npt.assert_allclose(mss[k], post_m[k * d:(k + 1) * d], rtol=1e-8, atol=1e-10)
        npt.assert_allclose(Pss[k], post_C[k * d:(k + 1) * d, k * d:(k + 1) * d], rtol=1e-8, atol=1e-10)


def test_crlb_lgssm_batched_kf():
    """test_crlb.py:19-73 with a NumPy stream and 20 000 trials instead of 10^6 (tolerance atol=1e-1 unchanged)."""
    ell, sigma, dt, T = 1., 1., 0.1, 10
    A = np.array([[0., 1.], [-3 / ell ** 2, -2 * math.sqrt(3) / ell]])
    Bv = np.array([0., 2 * sigma * (math.sqrt(3) / ell) ** 1.5])
    F, Sigma = tl.lti_sde_to_disc(A, Bv, dt)
    chol = np.linalg.cholesky(Sigma)
    Xi, H = 1., np.array([1., 0.])
    m0 = np.zeros(2)
    P0 = np.diag([sigma ** 2, 3 / ell ** 2 * sigma ** 2])
    n = 20000
    rng = np.random.default_rng(666)
    x = m0[:, None] + np.sqrt(P0) @ rng.standard_normal((2, n))
    xss, yss = np.zeros((n, T, 2)), np.zeros((n, T))
    for k in range(T):
        x = F @ x + chol @ rng.standard_normal((2, n))
        xss[:, k] = x.T
        yss[:, k] = H @ x + math.sqrt(Xi) * rng.standard_normal(n)

    mfs, Pfs, _ = fs.batched(fs.kf, (6,), F, Sigma, H, Xi, m0, P0, yss[:200])
    npt.assert_array_equal(Pfs[3], Pfs[150])                       # test_crlb.py:65-66
    # Monte-Carlo moment check on all n trials with the vectorised mean recursion (covariances are shared)
    P = Pfs[0]
    m = np.tile(m0[:, None], (1, n))
    Pk = P0
    for k in range(T):
        mp, Pp = F @ m, F @ Pk @ F.T + Sigma
        S = H @ Pp @ H + Xi
        K = Pp @ H / S
        m = mp + K[:, None] * (yss[:, k] - H @ mp)
        Pk = Pp - np.outer(K, K) * S
        npt.assert_allclose(Pk, P[k], rtol=1e-12)
        res = m.T - xss[:, k]
        npt.assert_allclose(np.einsum('ni,nj->ij', res, res) / n, P[k], atol=1e-1)   # test_crlb.py:71-73
    npt.assert_allclose(mfs[:, -1], m.T[:200], rtol=1e-10)


def test_jacobian_complex_step_matches_closed_form():
    """N1 / N2 of SURVEY.md: analytic Jacobians of the chirp LCD mean and SDE drift."""
    lam, b, ell, sigma, dt = 0.3, 0.2, 0.7, 1.3, 1e-2
    u = np.array([0.3, -0.8, 1.7, 0.4])
    f = md.disc_chirp_lcd(lam, b, ell, sigma)
    J = fs.jacobian(lambda v: f(v, dt)[0], u)
    e = math.exp(-lam * dt)
    sig = 1 / (1 + math.exp(-u[2]))
    th, thp = dt * 2 * math.pi * md.g(u[2]), dt * 2 * math.pi * sig
    M, _ = md.m32_solution(ell, sigma, dt)
    Jref = md.blkdiag(e * np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]]), M)
    Jref[0, 2] = e * thp * (-math.sin(th) * u[0] - math.cos(th) * u[1])
    Jref[1, 2] = e * thp * (math.cos(th) * u[0] - math.sin(th) * u[1])
    npt.assert_allclose(J, Jref, rtol=1e-13, atol=1e-15)

    drift = md.model_chirp(lam, b, ell, sigma, 0.1)[0]
    Ja = fs.jacobian(drift, u)
    w = 2 * math.pi * md.g(u[2])
    gam = math.sqrt(3) / ell
    Aref = np.array([[-lam, -w, 0, 0], [w, -lam, 0, 0], [0, 0, 0, 1], [0, 0, -gam ** 2, -2 * gam]])
    Aref[0, 2] += -2 * math.pi * sig * u[1]
    Aref[1, 2] += 2 * math.pi * sig * u[0]
    npt.assert_allclose(Ja, Aref, rtol=1e-13, atol=1e-15)


def test_nan_semantics_non_pd_cholesky():
    """CS-4 of SURVEY.md: a non-PD covariance gives NaN outputs from then on, never an exception."""
    sg = SigmaPoints.cubature(2)
    f = lambda u, dt: (0.9 * u, 0.01 * np.eye(2))
    P0 = np.array([[1., 2.], [2., 1.]])          # indefinite
    mfs, Pfs, nll = fs.sgp_filter(f, sg, np.array([1., 0.]), 0.1, np.zeros(2), P0, 0.1, np.ones(5))
    assert np.all(np.isnan(mfs)) and np.all(np.isnan(Pfs)) and np.all(np.isnan(nll))


def test_lorenz63_discrete_vs_continuous_discrete():
    """The intent of the reference's test/test_ekfs.py:11-62 -- ekf / eks on a discretised Lorenz-63 against cd_ekf /
    cd_eks on the SDE, same loose tolerances -- without its two unavailable ingredients: the order-2 TME discretisation
    (third-party `tme`) is replaced by one RK4 step of the drift for the mean and the second-order expansion
    Gamma dt + (J Gamma + Gamma J^T) dt^2 / 2 for the covariance (both agree with TME-2 to O(dt^3)), and the jax.random
    draws by NumPy's."""
    import math
    kappa, lam, mu = 10., 28., 2.
    Gamma = 25. * np.eye(3)                                   # dispersion 5 I

    def drift(u):
        return np.array([kappa * (u[1] - u[0]), u[0] * (lam - u[2]) - u[1], u[0] * u[1] - mu * u[2]])

    def drift_jac(u):
        return np.array([[-kappa, kappa, 0.], [lam - u[2], -1., -u[0]], [u[1], u[0], -mu]])

    def m_and_cov(u, dt):
        k1 = drift(u); k2 = drift(u + 0.5 * dt * k1); k3 = drift(u + 0.5 * dt * k2); k4 = drift(u + dt * k3)
        J = drift_jac(np.real(u))
        return u + dt * (k1 + 2 * k2 + 2 * k3 + k4) / 6, Gamma * dt + (J @ Gamma + Gamma @ J.T) * dt ** 2 / 2

    dt, T, Xi = 1e-3, 2000, 2.
    H, m0, P0 = np.array([1., 0., 0.]), np.zeros(3), np.eye(3)
    rng = np.random.default_rng(666)
    x = m0 + rng.standard_normal(3)
    traj = np.empty((T, 3))
    for k in range(T):
        m, cov = m_and_cov(x, dt)
        x = m + np.linalg.cholesky(cov) @ rng.standard_normal(3)
        traj[k] = x
    ys = traj[:, 0] + math.sqrt(Xi) * rng.standard_normal(T)

    f = fs.ekf(m_and_cov, H, Xi, m0, P0, dt, ys)
    s = fs.eks(m_and_cov, f[0], f[1], dt)
    cf = fs.cd_ekf(drift, lambda _: 5. * np.eye(3), H, Xi, m0, P0, dt, ys)
    cs_ = fs.cd_eks(drift, lambda _: 5. * np.eye(3), cf[0], cf[1], dt)
    npt.assert_allclose(f[0], cf[0], rtol=0.2, atol=0.05)     # the reference has no atol; zero crossings of a state need one
    npt.assert_allclose(f[1], cf[1], rtol=0.21, atol=1e-3)
    npt.assert_allclose(f[2], cf[2], rtol=1e-5, atol=1e-2)
    assert np.all(np.isfinite(s[0])) and np.all(np.isfinite(cs_[0]))
    npt.assert_allclose(s[0][:-200], cs_[0][:-200], rtol=0.2, atol=0.5)      # states range over +-20; zero crossings
