"""Models compiled at run time (cgp_model_from_source; csrc/cgp_custom.hpp): the reference's filters take ANY traceable callable and
differentiate it with jax.jacfwd; the engine takes the model as a few lines of device source, instantiates the generic kernels on it with
hiprtc and differentiates it with dual numbers in the kernel.  First user: the Lorenz-63 system of the reference's own test/test_ekfs.py:11-62
-- the one case of the reference's test suite the enumerated models could not run (VERDICT r5 missing #5)."""
import math

import numpy as np
import numpy.testing as npt
import pytest

from tests import cases as cs

pytestmark = pytest.mark.gpu

LORENZ = r'''
// Lorenz-63 (test/test_ekfs.py:11-28): p = [kappa, lam, mu, gamma_diag]
template <class T> __device__ void drift(const T* u, const double* p, T* a) {
    a[0] = p[0] * (u[1] - u[0]);
    a[1] = u[0] * (p[1] - u[2]) - u[1];
    a[2] = u[0] * u[1] - p[2] * u[2];
}
// its discretisation: one RK4 step of the drift for the mean, Gamma dt + (J Gamma + Gamma J^T) dt^2 / 2 for the covariance
template <class T> __device__ void cond_mean(const T* u, const double* p, double dt, T* m) {
    T k1[3], k2[3], k3[3], k4[3], x[3];
    drift(u, p, k1);
    for (int i = 0; i < 3; i++) x[i] = u[i] + 0.5 * dt * k1[i];
    drift(x, p, k2);
    for (int i = 0; i < 3; i++) x[i] = u[i] + 0.5 * dt * k2[i];
    drift(x, p, k3);
    for (int i = 0; i < 3; i++) x[i] = u[i] + dt * k3[i];
    drift(x, p, k4);
    for (int i = 0; i < 3; i++) m[i] = u[i] + dt * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6;
}
__device__ void cond_cov(const double* u, const double* p, double dt, double* cov) {
    const double G = p[3];
    const double J[3][3] = {{-p[0], p[0], 0.0}, {p[1] - u[2], -1.0, -u[0]}, {u[1], u[0], -p[2]}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) cov[i * 3 + j] = (i == j ? G * dt : 0.0) + (J[i][j] * G + G * J[j][i]) * dt * dt / 2;
}
'''
KAPPA, LAM, MU = 10., 28., 2.


def lorenz_host():
    Gamma = 25. * np.eye(3)

    def drift(u):
        return np.array([KAPPA * (u[1] - u[0]), u[0] * (LAM - u[2]) - u[1], u[0] * u[1] - MU * u[2]])

    def drift_jac(u):
        return np.array([[-KAPPA, KAPPA, 0.], [LAM - u[2], -1., -u[0]], [u[1], u[0], -MU]])

    def m_and_cov(u, dt):
        k1 = drift(u); k2 = drift(u + 0.5 * dt * k1); k3 = drift(u + 0.5 * dt * k2); k4 = drift(u + dt * k3)
        J = drift_jac(np.real(u))
        return u + dt * (k1 + 2 * k2 + 2 * k3 + k4) / 6, Gamma * dt + (J @ Gamma + Gamma @ J.T) * dt ** 2 / 2
    return drift, m_and_cov


def lorenz_data(T=2000, dt=1e-3, Xi=2., seed=666):
    drift, m_and_cov = lorenz_host()
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(3)
    traj = np.empty((T, 3))
    for k in range(T):
        m, cov = m_and_cov(x, dt)
        x = m + np.linalg.cholesky(cov) @ rng.standard_normal(3)
        traj[k] = x
    return traj[:, 0] + math.sqrt(Xi) * rng.standard_normal(T)


def test_lorenz63_of_the_references_own_test_on_the_hip_path():
    """test/test_ekfs.py:50-62: ekf / eks on the discretised system against cd_ekf / cd_eks on the SDE, the reference's tolerances -- and
    each of the four against the NumPy oracle on the same callables at 1e-9 (the kernels' dual-number Jacobian against the oracle's
    complex step)."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from oracle import np_filters as nf
    drift, m_and_cov = lorenz_host()
    dt, T, Xi = 1e-3, 2000, 2.
    H, m0, P0 = np.array([1., 0., 0.]), np.zeros(3), np.eye(3)
    ys = lorenz_data(T, dt, Xi)
    p = np.array([KAPPA, LAM, MU, 25.])
    disc = pm.custom_cond_m_cov(LORENZ, 3, p, host=m_and_cov)
    sde, disp = pm.custom_sde(LORENZ, 3, p, 5. * np.eye(3), host=drift)

    f = fs.ekf(disc, H, Xi, m0, P0, dt, ys)
    s = fs.eks(disc, f[0], f[1], dt)
    cf = fs.cd_ekf(sde, disp, H, Xi, m0, P0, dt, ys)
    csm = fs.cd_eks(sde, disp, cf[0], cf[1], dt)
    # the reference's assertions (an atol where a state crosses zero, as in tests/test_oracle_filters.py)
    npt.assert_allclose(f[0], cf[0], rtol=0.2, atol=0.05)
    npt.assert_allclose(f[1], cf[1], rtol=0.21, atol=1e-3)
    npt.assert_allclose(f[2], cf[2], rtol=1e-5, atol=1e-2)
    assert np.all(np.isfinite(s[0])) and np.all(np.isfinite(csm[0]))
    npt.assert_allclose(s[0][:-200], csm[0][:-200], rtol=0.2, atol=0.5)
    # the oracle on the same model (Lorenz-63 is chaotic: the recursions amplify roundings by ~1e3 over 2000 steps)
    wf = nf.ekf(m_and_cov, H, Xi, m0, P0, dt, ys)
    ws = nf.eks(m_and_cov, wf[0], wf[1], dt)
    wcf = nf.cd_ekf(drift, lambda _: 5. * np.eye(3), H, Xi, m0, P0, dt, ys)
    wcs = nf.cd_eks(drift, lambda _: 5. * np.eye(3), wcf[0], wcf[1], dt)
    for got, want, name in ((f, wf, 'ekf'), (cf, wcf, 'cd_ekf')):
        for g, w, n in zip(got, want, ('m', 'P', 'nll')):
            cs.assert_close(g, w, 1e-9, f'lorenz {name} {n}')
    for got, want, name in ((fs.eks(disc, wf[0], wf[1], dt), ws, 'eks'), (fs.cd_eks(sde, disp, wcf[0], wcf[1], dt), wcs, 'cd_eks')):
        for g, w, n in zip(got, want, ('m', 'P')):
            cs.assert_close(g, w, 1e-9, f'lorenz {name} {n} on the oracle filter rows')
    assert disc(np.ones(3), dt)[0].shape == (3,)            # the host callable rides along


def test_sigma_point_methods_on_custom_models():
    """sgp_filter / sgp_smoother / cd_sgp_filter / cd_sgp_smoother on Lorenz-63 as source (cubature and Gauss-Hermite order 3), against the
    NumPy oracle on the same callables -- a discretisation whose COVARIANCE depends on the state, so the fan has to evaluate it at every
    point (filters_smoothers.py:118-120), which the enumerated models never exercise."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import np_filters as nf
    drift, m_and_cov = lorenz_host()
    dt, T, Xi = 1e-3, 600, 2.
    H, m0, P0 = np.array([1., 0., 0.]), np.zeros(3), np.eye(3)
    ys = np.stack([lorenz_data(T, dt, Xi, seed=700 + i) for i in range(3)])
    p = np.array([KAPPA, LAM, MU, 25.])
    disc = pm.custom_cond_m_cov(LORENZ, 3, p, host=m_and_cov)
    sde, disp = pm.custom_sde(LORENZ, 3, p, 5. * np.eye(3), host=drift)
    b = 5. * np.eye(3)
    for sg in (SigmaPoints.cubature(3), SigmaPoints.gauss_hermite(3, 3)):
        osg = cs.osig(sg)
        f = fs.sgp_filter(disc, sg, H, Xi, m0, P0, dt, ys)
        cf = fs.cd_sgp_filter(sde, b, sg, H, Xi, m0, P0, dt, ys)
        for i in range(3):
            wf = nf.sgp_filter(m_and_cov, osg, H, Xi, m0, P0, dt, ys[i])
            wcf = nf.cd_sgp_filter(drift, b, osg, H, Xi, m0, P0, dt, ys[i])
            for g, w, n in zip(f, wf, ('m', 'P', 'nll')):
                cs.assert_close(g[i], w, 1e-9, f'sgp_filter {sg.n_points} points {n}')
            for g, w, n in zip(cf, wcf, ('m', 'P', 'nll')):
                cs.assert_close(g[i], w, 1e-9, f'cd_sgp_filter {sg.n_points} points {n}')
            if i == 0:
                ws = nf.sgp_smoother(m_and_cov, osg, wf[0], wf[1], dt)
                wcs = nf.cd_sgp_smoother(drift, b, osg, wcf[0], wcf[1], dt)
                for g, w, n in zip(fs.sgp_smoother(disc, sg, wf[0], wf[1], dt), ws, ('m', 'P')):
                    cs.assert_close(g, w, 1e-9, f'sgp_smoother {sg.n_points} points {n}')
                for g, w, n in zip(fs.cd_sgp_smoother(sde, b, sg, wcf[0], wcf[1], dt), wcs, ('m', 'P')):
                    cs.assert_close(g, w, 1e-9, f'cd_sgp_smoother {sg.n_points} points {n}')


CHIRP = r'''
// the chirp LCD model of models.py:264-311 written as a custom model: p = [lam, b, ell, sigma]
template <class T> __device__ void cond_mean(const T* u, const double* p, double dt, T* m) {
    const double lam = p[0], ell = p[2];
    const T th = dt * (2 * 3.14159265358979323846) * softplus(u[2]);
    const T c = cos(th), s = sin(th);
    const double rho = ::exp(-lam * dt), g = ::sqrt(3.0) / ell, eta = dt * g, e = ::exp(-eta);
    m[0] = rho * (c * u[0] - s * u[1]);
    m[1] = rho * (s * u[0] + c * u[1]);
    m[2] = ((1 + eta) * e) * u[2] + (dt * e) * u[3];
    m[3] = (-dt * g * g * e) * u[2] + ((1 - eta) * e) * u[3];
}
__device__ void cond_cov(const double* u, const double* p, double dt, double* cov) {
    const double lam = p[0], b = p[1], ell = p[2], sigma = p[3];
    const double q = lam == 0.0 ? b * b * dt : b * b / (2 * lam) * (1 - ::exp(-2 * lam * dt));
    const double g = ::sqrt(3.0) / ell, eta = dt * g, beta = sigma * sigma * ::exp(-2 * eta);
    for (int i = 0; i < 16; i++) cov[i] = 0.0;
    cov[0] = q; cov[5] = q;
    cov[10] = sigma * sigma - beta * (2 * eta + 2 * eta * eta + 1);
    cov[11] = cov[14] = 2 * dt * dt * g * g * g * beta;
    cov[15] = g * g * (sigma * sigma + beta * (2 * eta - 2 * eta * eta - 1));
}
'''


def test_a_custom_chirp_model_reproduces_the_compiled_in_one():
    """The same model through both doors, batched with per-trial parameters: dual numbers against the analytic Jacobian of SURVEY N1,
    the library's exp / log / sincos against the engine's polynomials -- 1e-10."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    B, T = 70, 600
    rng = np.random.default_rng(5)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * rng.uniform(0.8, 1.25, size=(B, 6))
    params[3, 0] = 0.0                                       # the lam = 0 branch
    drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
    ys = np.stack([cs.chirp_measurements(T, 300 + i)[2] for i in range(B)])
    custom = pm.custom_cond_m_cov(CHIRP, 4, params[:, [0, 1, 3, 4]])
    want = fs.ekf(disc, H, 0.1, m0, P0, 1e-3, ys, flags=0x4)
    got = fs.ekf(custom, H, 0.1, m0, P0, 1e-3, ys)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        cs.assert_close(g, w, 1e-10, f'custom chirp ekf {n}')
    ws = fs.eks(disc, want[0], want[1], 1e-3, flags=0x4)
    gs = fs.eks(custom, want[0], want[1], 1e-3)
    for g, w, n in zip(gs, ws, ('mss', 'Pss')):
        cs.assert_close(g, w, 1e-10, f'custom chirp eks {n}')
    last = fs.ekf(custom, H, 0.1, m0, P0, 1e-3, ys, nll_final_only=True, want=(False, False, True))
    npt.assert_allclose(last[2], got[2][:, -1], rtol=1e-14)
    one = fs.ekf(pm.custom_cond_m_cov(CHIRP, 4, params[5, [0, 1, 3, 4]]), H, 0.1, m0[5], P0[5], 1e-3, ys[5])        # a single record
    assert one[0].shape == (T, 4)
    cs.assert_close(one[0], got[0][5], 1e-13, 'single record')


def test_errors_are_named():
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    bad = pm.custom_cond_m_cov(CHIRP.replace('m[3] =', 'm[3] = undefined_symbol +'), 4, np.ones(4))
    with pytest.raises(RuntimeError, match='undefined_symbol'):
        fs.ekf(bad, np.array([0., 1., 0., 0.]), 0.1, np.zeros(4), np.eye(4), 1e-3, np.zeros(10))
    ok = pm.custom_cond_m_cov(CHIRP, 4, np.array([0.1, 0.1, 1., 1.]))
    with pytest.raises(ValueError, match='sigma points are for d'):
        fs.sgp_filter(ok, SigmaPoints.cubature(3), np.array([0., 1., 0., 0.]), 0.1, np.zeros(4), np.eye(4), 1e-3, np.zeros(10))
    with pytest.raises(TypeError, match='time_split'):
        fs.ekf(ok, np.array([0., 1., 0., 0.]), 0.1, np.zeros(4), np.eye(4), 1e-3, np.zeros(10), time_split=(2, 64))
    with pytest.raises(RuntimeError, match='dimension'):
        fs.ekf(pm.custom_cond_m_cov(CHIRP, 9, np.ones(4)), np.ones(9), 0.1, np.zeros(9), np.eye(9), 1e-3, np.zeros(10))
    with pytest.raises(TypeError, match='no host callable'):
        ok(np.zeros(4), 1e-3)


# ---------------------------------------------------------------------------------------------- ekf_for_kpt with a measurement function of one's own
KPT_H2 = r'''
// models.py:539-580 for two harmonics: h(u) = sum_k u[k] sin(k g(u[0] + u[d-1])), d = 4
template <class T> __device__ T measure(const T* u, const double* q) {
    const T phase = softplus(u[0] + u[3]);
    return u[1] * sin(phase) + u[2] * sin(phase * 2.0);
}
'''
OTHER_H = r'''
// nothing the library enumerates: per-trial coefficients q, a product and a tanh
template <class T> __device__ T measure(const T* u, const double* q) {
    return q[0] * u[0] * sin(u[1]) + q[1] * tanh(u[2]) + exp(u[0] * q[2]);
}
'''


def test_ekf_for_kpt_with_a_measurement_function_as_source():
    """filters_smoothers.py:267-314 takes ANY traceable h (H = jacfwd(h)(mp), :304): the reference's own harmonic measurement handed over as
    source reproduces the compiled-in one (1e-11; the generic kernel's naive softplus against the tile-layout kernel's), and a measurement
    function of another kind, with per-trial coefficients, agrees with the NumPy oracle (complex-step Jacobian) at 1e-9."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from oracle import np_filters as onf
    c = cs.kpt_case(T=300, seed=31, nh=2)
    ys = c.ys[None, :300] + 0.05 * np.random.default_rng(3).standard_normal((5, 300))
    want = fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, ys)
    h = pm.custom_measurement(KPT_H2, 4)
    got = fs.ekf_for_kpt(c.F, c.Sigma, h, c.Xi, c.m0, c.P0, c.dt, ys)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        cs.assert_close(g, w, 1e-11, f'custom kpt measurement {n}')
    last = fs.ekf_for_kpt(c.F, c.Sigma, h, c.Xi, c.m0, c.P0, c.dt, ys, nll_final_only=True, want=(False, False, True))[2]
    cs.assert_close(last, want[2][:, -1], 1e-11, 'final NLL')
    # another h, d = 3, dense F, per-trial q
    rng = np.random.default_rng(8)
    d, T, B = 3, 200, 4
    F = np.eye(d) + 0.05 * rng.standard_normal((d, d))
    A = 0.1 * rng.standard_normal((d, d))
    Sigma = A @ A.T + 0.01 * np.eye(d)
    q = np.stack([np.array([1.0, 0.5, 0.1]) + 0.1 * rng.standard_normal(3) for _ in range(B)])
    m0, P0, Xi = np.array([0.3, 0.2, -0.1]), 0.5 * np.eye(d), 0.05
    ys2 = rng.standard_normal((B, T))
    h2 = pm.custom_measurement(OTHER_H, d, q)
    got2 = fs.ekf_for_kpt(F, Sigma, h2, Xi, m0, P0, 1e-2, ys2)
    for b in range(B):
        def host(u, qb=q[b]):
            return qb[0] * u[0] * np.sin(u[1]) + qb[1] * np.tanh(u[2]) + np.exp(u[0] * qb[2])
        want2 = onf.ekf_for_kpt(F, Sigma, host, Xi, m0, P0, 1e-2, ys2[b])
        for g, w, n in zip(got2, want2, ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g[b], w, 1e-9, f'custom measurement trial {b} {n}')
    with pytest.raises(ValueError, match='F must be 3 x 3'):
        fs.ekf_for_kpt(np.eye(4), np.eye(4), h2, Xi, np.zeros(4), np.eye(4), 1e-2, ys2)
    with pytest.raises(ValueError, match='at most d'):
        pm.custom_measurement(OTHER_H, 3, np.ones(5))
    with pytest.raises(TypeError, match='no host callable'):
        h2(np.zeros(3))
    # the compiled programs can be unloaded and come back on the next call
    from chirpgp_amd import _engine
    assert _engine.release_custom_models() >= 2 and _engine.release_custom_models() == 0
    again = fs.ekf_for_kpt(c.F, c.Sigma, h, c.Xi, c.m0, c.P0, c.dt, ys)
    for g, w in zip(again, got):
        assert np.array_equal(g, w)
