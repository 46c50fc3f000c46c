"""ekf_for_kpt (filters_smoothers.py:267-314; tetralith/jobs/kpt_mle.py, harmonic_kpt_mle.py) in the tile layout (cgp_kpt8.hpp: one wavefront
per trial, covariance one entry per lane, the harmonic measurement as a wave-uniform scalar chain) against the C port and the committed
golden vector, for n_harm = 1, 2, 3 (d = 3, 4, 5), a dense F, per-trial parameters, ragged record lengths and NaN measurements."""
import numpy as np
import pytest

from tests import cases as cs
from tests import backends as bk

pytestmark = pytest.mark.gpu
RTOL = 1e-9
GENERIC = dict(flags=0x2 | 0x10)             # CGP_WAVE_PER_TRIAL | CGP_GENERIC_KERNEL: the kernel the tile layout replaces
WAVE = dict(flags=0x2)


def _port(c, ys):
    from chirpgp_amd import models as pm
    from oracle import port
    spec = pm.linear_cond_m_cov(c.F, c.Sigma)
    spec.model_id, spec.n_harm = pm.M_KPT, c.nh
    return port.filter(port.F_EKF_KPT, spec, None, None, c.Xi, c.m0, c.P0, c.dt, ys)


@pytest.mark.parametrize('nh', [1, 2, 3])
@pytest.mark.parametrize('T', [200, 64, 65, 1])
def test_tile_layout_kernel_against_the_port(nh, T):
    from chirpgp_amd import filters_smoothers as fs
    c = cs.kpt_case(T=max(T, 2), seed=20 + nh, nh=nh)
    ys = c.ys[None, :T] + 0.05 * np.random.default_rng(nh).standard_normal((7, T))
    want = _port(c, ys)
    for kw, name in ((WAVE, 'tile layout'), (GENERIC, 'generic')):
        got = fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, ys, **kw)
        for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g, w, RTOL, f'kpt{nh} T={T} {name} {n}')
    last = fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, ys, nll_final_only=True, want=(False, False, True), **WAVE)[2]
    cs.assert_close(last, want[2][:, -1], RTOL, 'final NLL')
    only_m = fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, ys, want=(True, False, False), **WAVE)
    assert only_m[1] is None and only_m[2] is None
    cs.assert_close(only_m[0], want[0], RTOL, 'means only')


def test_golden_vector_and_default_launch():
    """tests/golden/kpt2.npz (made by the NumPy oracle, tests/golden/make_golden.py) through the default launch: at B = 1 that IS the tile-
    layout kernel."""
    from chirpgp_amd import filters_smoothers as fs
    c = cs.kpt_case(T=200, seed=24)
    _, want = bk.load_golden('kpt2')
    got = {'ekf_for_kpt': fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, c.ys)}
    bk.compare(got, want, RTOL, 'kpt2 golden')
    assert all(np.array_equal(a, b) for a, b in zip(got['ekf_for_kpt'], fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, c.ys, **WAVE)))


@pytest.mark.parametrize('nh', [1, 3])
def test_dense_dynamics_and_per_trial_parameters(nh):
    """The reference takes ANY (F, Sigma) (filters_smoothers.py:298): a dense, stable F with a full Sigma per trial -- no structure of
    build_kpt_chirp_model's F = I + e_{d-1} e_0^T is assumed by the matrix-instruction prediction -- and per-trial Xi, m0, P0; NaN
    measurements poison their trial from that step on and no other."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from oracle import port
    d, B, T = nh + 2, 9, 150
    rng = np.random.default_rng(100 + nh)
    c = cs.kpt_case(T=T, seed=31, nh=nh)
    F = np.eye(d)[None] * 0.995 + 0.002 * rng.standard_normal((B, d, d))
    F[:, -1, 0] += 1.0
    A = 0.003 * rng.standard_normal((B, d, d))
    Sigma = A @ np.swapaxes(A, -1, -2) + 1e-6 * np.eye(d)
    spec = pm.linear_cond_m_cov(F, Sigma)
    spec.model_id, spec.n_harm = pm.M_KPT, nh
    Xi = 0.1 * (1 + rng.random(B))
    m0 = c.m0[None] * (1 + 0.05 * rng.standard_normal((B, d)))
    P0 = c.P0[None] * (1 + 0.1 * rng.random((B, 1, 1)))
    ys = c.ys[None, :] + 0.05 * rng.standard_normal((B, T))
    ys[4, 70] = np.nan
    want = port.filter(port.F_EKF_KPT, spec, None, None, Xi, m0, P0, c.dt, ys)
    got = fs.ekf_for_kpt(F, Sigma, c.h, Xi, m0, P0, c.dt, ys, **WAVE)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        cs.assert_close(g, w, RTOL, f'dense F nh={nh} {n}')
    assert np.isnan(got[0][4, 70:]).all() and np.isfinite(got[0][4, :70]).all() and np.isfinite(got[0][3]).all()
