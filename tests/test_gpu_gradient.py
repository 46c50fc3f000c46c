"""The EKF objective's EXACT gradient from the tangent kernel (cgp_ekf_nll_grad; SURVEY 8f-1 "in-kernel forward-tangent gradient";
VERDICT r5 #5): forward tangents of (m, P, nll) through the scan, where the reference takes jax.value_and_grad through its scan
(demos/ekfs_mle.py:43-51).  Checked against the derivative computed in 100-digit arithmetic (tests/golden/exact_grad.npz), against the
oracle's fourth-order difference quotient at the demos' full record length, and through the optimiser."""
import os

import numpy as np
import numpy.testing as npt
import pytest

from tests import mle_oracle as mo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'exact_grad.npz')
INIT = np.array([0.1, 0.1, 0.1, 1., 1., 7.])


def _record(T, seed, dt=1e-3, Xi=0.1):
    from chirpgp_amd.toymodels import gen_chirp, meow_freq, constant_mag
    ts = np.linspace(dt, dt * T, T)
    _, phase = meow_freq(offset=8.)
    return gen_chirp(ts, constant_mag(1.), phase) + np.sqrt(Xi) * np.random.default_rng(seed).standard_normal(T)


@pytest.mark.parametrize('name', ['exact_track', 'exact_lost'])
def test_value_and_gradient_against_100_digit_arithmetic(name):
    """1e-8 of the gradient's scale (1e-4 with the difference quotients of rounds 1 - 5), value 1e-11."""
    from chirpgp_amd import mle, models as pm
    z = np.load(GOLD)
    theta, ys = z[f'{name}.theta'], z[f'{name}.ys']
    f, grad = mle.value_and_grad(pm.build_chirp_model, theta[None, :], ys, float(z[f'{name}.Xi']), float(z[f'{name}.dt']))
    want = z[f'{name}.grad']
    err = np.abs(grad[0] - want).max() / np.abs(want).max()
    print(f'{name}: nll {f[0]:.12g} (exact {float(z[name + ".nll"]):.12g}), gradient error {err:.2e} of its scale {np.abs(want).max():.3g}')
    npt.assert_allclose(f[0], float(z[f'{name}.nll']), rtol=1e-11)
    assert err < 1e-8, (grad, want)
    # ... and the difference form it replaces, for the record: 1e-4 .. 1e-6
    fun = mle.make_objective('ekf', pm.build_chirp_model, ys, float(z[f'{name}.Xi']), float(z[f'{name}.dt']), exact=False)
    _, gfd = fun(theta)
    print(f'   central differences of 13 passes: {np.abs(gfd - want).max() / np.abs(want).max():.2e}')


def test_gradient_at_the_demos_record_length_and_batched():
    """T = 3141 (demos/ekfs_mle.py:16-18) against the oracle's fourth-order difference quotient (good to ~1e-8); four parameter vectors
    on three records in ONE launch, record addressing as in the difference form; the La Scala builder (4 parameters)."""
    from chirpgp_amd import mle, models as pm
    ys = _record(3141, 555)
    th = np.stack([mo.g_inv(INIT), mo.g_inv(INIT * np.array([1.3, 2.0, 0.7, 1.5, 3.0, 1.2]))])
    f, grad = mle.value_and_grad(pm.build_chirp_model, th, ys, 0.1, 1e-3)
    for i in range(2):
        f_o, g_o = mo.value_and_grad('ekf', pm.build_chirp_model, th[i], ys, 0.1, 1e-3)
        npt.assert_allclose(f[i], f_o, rtol=1e-9)
        npt.assert_allclose(grad[i], g_o, rtol=2e-7, atol=2e-7 * np.abs(g_o).max())
    recs = np.stack([_record(800, 600 + r) for r in range(3)])
    thetas = np.repeat(th, 3, axis=0)[:6].reshape(3, 2, 6)          # two parameter vectors per record
    thetas = np.concatenate([thetas[r] for r in range(3)])
    f, grad = mle.value_and_grad(pm.build_chirp_model, thetas, recs, 0.1, 1e-3)
    for r in range(3):
        for j in range(2):
            f_o, g_o = mo.value_and_grad('ekf', pm.build_chirp_model, thetas[2 * r + j], recs[r], 0.1, 1e-3)
            npt.assert_allclose(f[2 * r + j], f_o, rtol=1e-9)
            npt.assert_allclose(grad[2 * r + j], g_o, rtol=2e-7, atol=2e-7 * np.abs(g_o).max())
    f1, g1 = mle.value_and_grad(pm.build_chirp_model, thetas[[2, 3]], recs, 0.1, 1e-3, record_index=[1])
    npt.assert_array_equal(f1, f[[2, 3]])
    npt.assert_array_equal(g1, grad[[2, 3]])
    la = mo.g_inv(np.array([0.1, 1., 1., 7.]))
    f, grad = mle.value_and_grad(pm.build_lascala_model, la[None, :], recs[0], 0.1, 1e-3)
    f_o, g_o = mo.value_and_grad('ekf', pm.build_lascala_model, la, recs[0], 0.1, 1e-3)
    npt.assert_allclose(f[0], f_o, rtol=1e-9)
    npt.assert_allclose(grad[0], g_o, rtol=2e-7, atol=2e-7 * np.abs(g_o).max())


def test_fit_with_exact_gradients():
    """mle.fit on the tangent kernel: the oracle optimum (SciPy L-BFGS-B on the port's objective) in no more iterations, and the
    difference form's optimum reproduced."""
    from chirpgp_amd import mle, models as pm
    ys = _record(3141, 555)
    assert mle.has_exact_gradient('ekf', pm.build_chirp_model, 0.1) and not mle.has_exact_gradient('sgp_filter', pm.build_chirp_model, 0.1)
    opt, res = mle.fit('ekf', pm.build_chirp_model, INIT, ys, 0.1, 1e-3, maxiter=300, exact=True)
    opt_fd, res_fd = mle.fit('ekf', pm.build_chirp_model, INIT, ys, 0.1, 1e-3, maxiter=300, exact=False)
    opt_o, res_o = mo.fit('ekf', pm.build_chirp_model, INIT, ys, 0.1, 1e-3)
    print(f'exact: nll {res.fun:.9g} in {res.nit} iterations ({res.nfev} launches); differences: {res_fd.fun:.9g} in {res_fd.nit} ({res_fd.nfev}); '
          f'oracle + SciPy: {res_o.fun:.9g} in {res_o.nit} ({res_o.nfev})')
    npt.assert_allclose(res.fun, res_o.fun, rtol=1e-6)
    assert res.nit <= res_o.nit + 2
    keep = np.array([0, 2, 3, 4, 5])
    npt.assert_allclose(opt[keep], opt_o[keep], rtol=3e-3)       # (delta sits in a flat direction: the two optima's NLL agree to 5e-9)
    npt.assert_allclose(mo.nll('ekf', pm.build_chirp_model, pm.g_inv(opt), ys, 0.1, 1e-3)[0], res.fun, rtol=1e-9)


def test_lockstep_fit_many_takes_the_tangent_kernel():
    from chirpgp_amd import mle, models as pm
    T, R = 1200, 3
    recs = np.stack([_record(T, 700 + r) for r in range(R)])
    many, info = mle.fit_many('ekf', pm.build_chirp_model, INIT, recs, 0.1, 1e-3, maxiter=200, exact=True)
    fd, info_fd = mle.fit_many('ekf', pm.build_chirp_model, INIT, recs, 0.1, 1e-3, maxiter=200, exact=False)
    for r in range(R):
        _, res_o = mo.fit('ekf', pm.build_chirp_model, INIT, recs[r], 0.1, 1e-3)
        assert info['fun'][r] <= res_o.fun + 1e-5 * abs(res_o.fun), (r, info['fun'][r], res_o.fun)
    with pytest.raises(ValueError, match='tangent kernel'):
        mle.fit_many('sgp_filter', pm.build_chirp_model, INIT, recs, 0.1, 1e-3, maxiter=2, exact=True)
    assert not mle._exact_by_default('ekf', pm.build_chirp_model, 0.1, 3, {}) and mle._exact_by_default('ekf', pm.build_chirp_model, 0.1, 4000, {})
    print('launches exact / differences:', info['launches'], info_fd['launches'], 'fun', info['fun'], info_fd['fun'])


def test_c_abi_argument_errors():
    import ctypes as C
    import torch
    from chirpgp_amd import _engine as E, models as pm
    lib, ctx = E.load_library(), E.context()
    drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(INIT, 2)
    keep = []
    model = E._model_struct(disc, None, 1, keep)
    init = E._init_struct(H, 0.1, m0, P0, 6, 1, keep)
    x = torch.zeros(64, dtype=torch.float64, device='cuda')
    rc = lib.cgp_ekf_nll_grad(ctx, C.byref(model), C.byref(init), 1e-3, x.data_ptr(), 64, 1, None, 1, 64, x.data_ptr(), 1, x.data_ptr(), x.data_ptr(), 0, None)
    assert rc == -2 and b'd = 4' in lib.cgp_last_error(ctx)
    drift, disp, disc, m0, P0, H = pm.build_chirp_model(INIT)
    model = E._model_struct(disc, None, 1, keep)
    init = E._init_struct(H, 0.1, m0, P0, 4, 1, keep)
    assert lib.cgp_ekf_nll_grad(ctx, C.byref(model), C.byref(init), 1e-3, x.data_ptr(), 64, 1, None, 1, 64, None, 1, x.data_ptr(), x.data_ptr(), 0, None) == -1
    assert lib.cgp_ekf_nll_grad(ctx, C.byref(model), C.byref(init), 1e-3, x.data_ptr(), 64, 0, None, 1, 64, x.data_ptr(), 1, x.data_ptr(), x.data_ptr(), 0, None) == -1
    assert lib.cgp_ekf_nll_grad(ctx, C.byref(model), C.byref(init), 1e-3, x.data_ptr(), 64, 1, None, 0, 64, x.data_ptr(), 1, x.data_ptr(), x.data_ptr(), 0, None) == 0
