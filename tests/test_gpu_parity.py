"""GPU parity tests: the HIP path, called through the public functions (-> ctypes -> C-ABI), against the oracle.

Tolerance: the north star states 1e-5 relative on filtered / smoothed means and covariances (BASELINE.json); the gate
below is RTOL = 1e-5 everywhere (element-wise, with a 1e-3 * array-scale floor for structurally-zero covariance
entries; NaN positions must be identical), and a tighter 1e-9 where the arithmetic is benign (linear models).
Both launch shapes are exercised: one wavefront per trial and one lane per trial.
"""
import numpy as np
import numpy.testing as npt
import pytest

from tests import backends as bk
from tests import cases as cs

pytestmark = pytest.mark.gpu

RTOL = 1e-5
WAVE = dict(flags=0x2)
THREAD = dict(flags=0x4)
WAVE_SEQ = dict(flags=0x2 | 0x8 | 0x10)     # wave per trial, step-by-step smoother scan, generic (non-cooperative) filter kernels
WAVE_GENERIC = dict(flags=0x2 | 0x10)       # wave per trial, generic filter kernels, lane-scan time-parallel smoothers (no tile-layout kernels)
WAVE_LITERAL = dict(flags=0x2 | 0x40)       # cooperative kernels summing over every sigma point (no collapsed quadrature)
WAVE_DPP = dict(flags=0x2 | 0x80)           # d = 4 EKF on the DPP cooperative kernel instead of the MFMA one
WAVE_X4 = dict(flags=0x2 | 0x200)          # d = 4 matrix-core EKF with four trials per wavefront (default only above B = 1024)
SHAPES = [pytest.param(WAVE, id='wave_per_trial'), pytest.param(WAVE_X4, id='wave_four_trials'), pytest.param(WAVE_DPP, id='wave_dpp_ekf'), pytest.param(WAVE_SEQ, id='wave_sequential_scan'), pytest.param(WAVE_GENERIC, id='wave_generic_kernels'),
          pytest.param(THREAD, id='lane_per_trial'), pytest.param(WAVE_LITERAL, id='wave_literal_sigma_sum')]


def _fs():
    from chirpgp_amd import filters_smoothers as fs
    return fs


# ------------------------------------------------------------------ the reference's own test, on the HIP path
@pytest.mark.parametrize('idx', [0, 1])
def test_equivalence_on_linear_models(idx):
    """test/test_filters_smoothers.py:19-85 with the engine's functions (same data, same tolerances)."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    fs = _fs()
    c = cs.linear_case(idx)
    F, Sigma, H, Xi, m0, P0, dt, ys = c.F, c.Sigma, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys
    drift, dispersion = c.drift, c.disp
    m_and_cov = pm.linear_cond_m_cov(F, Sigma)
    B = dispersion(None)

    kf_results = fs.kf(F, Sigma, H, Xi, m0, P0, ys)
    ekf_results = fs.ekf(m_and_cov, H, Xi, m0, P0, dt, ys)
    cd_ekf_results = fs.cd_ekf(drift, dispersion, H, Xi, m0, P0, dt, ys)
    sgps = SigmaPoints.gauss_hermite(d=3, order=4)
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
    npt.assert_array_equal(rts_results[0][-1], kf_results[0][-1])
    npt.assert_array_equal(rts_results[1][-1], kf_results[1][-1])


# ------------------------------------------------------------------ golden vectors (NumPy-oracle outputs)
GOLDEN = {
    'linear_ou': (lambda: cs.linear_case(0, T=200), 200),
    'chirp_gh3': (lambda: cs.chirp_case(T=200, seed=21), 100),
    'harmonic3_cubature': (lambda: cs.harmonic_case(T=120, seed=22, nh=3), 60),
    'lascala_gh3': (lambda: cs.lascala_case(T=120, seed=23), 60),
}


@pytest.mark.parametrize('kw', SHAPES)
@pytest.mark.parametrize('name', sorted(GOLDEN))
def test_golden_vectors(name, kw):
    make, cd_T = GOLDEN[name]
    c = make()
    _, want = bk.load_golden(name)
    got = bk.run_pairs('hip', c, cd_T, hip_kw=kw)
    worst = bk.compare(got, want, RTOL, f'{name}')
    print(name, {k: f'{v:.1e}' for k, v in worst.items()})


def test_golden_kf_rts_and_kpt():
    fs = _fs()
    z, want = bk.load_golden('linear_ou')
    kf = fs.kf(z['F'], z['Sigma'], z['H'], float(z['Xi']), z['m0'], z['P0'], z['ys'])
    bk.compare({'kf': kf, 'rts': fs.rts(z['F'], z['Sigma'], kf[0], kf[1])}, want, 1e-9, 'linear_ou')
    c = cs.kpt_case(T=200, seed=24)
    _, want = bk.load_golden('kpt2')
    for kw in (WAVE, THREAD):
        got = {'ekf_for_kpt': fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, c.ys, **kw)}
        bk.compare(got, want, RTOL, 'kpt2')


# ------------------------------------------------------------------ longer runs against the C port
def _batch_case(make, B, **kw):
    cs_ = [make(seed=100 + i, **kw) for i in range(B)]
    c = cs_[0]
    c.ys = np.stack([x.ys for x in cs_])
    return c


@pytest.mark.parametrize('kw', SHAPES)
def test_chirp_gh3_long_batched(kw):
    """Config C2 / C3 shape at a size the C port finishes in seconds: d = 4, T = 3000, B = 6."""
    c = _batch_case(cs.chirp_case, 6, T=3000)
    want = bk.run_pairs('port', c, cd_T=600)
    got = bk.run_pairs('hip', c, cd_T=600, hip_kw=kw)
    worst = bk.compare(got, want, RTOL, 'chirp_long')
    print({k: f'{v:.1e}' for k, v in worst.items()})
    # smoothers on identical inputs (the oracle's filtering results)
    sm = bk.smoothers_on('hip', c, want, hip_kw=kw)
    bk.compare(sm, want, RTOL, 'chirp_long/smoothers_on_oracle_inputs')


@pytest.mark.parametrize('kw', SHAPES)
def test_harmonics_long_batched(kw):
    """Config C5 shape: 3 harmonics (d = 8) with cubature, and 2 harmonics (d = 6) with freq_scale != 1."""
    c = _batch_case(cs.harmonic_case, 3, T=1500, nh=3)
    bk.compare(bk.run_pairs('hip', c, cd_T=300, hip_kw=kw), bk.run_pairs('port', c, cd_T=300), RTOL, 'harmonic3')
    c = _batch_case(cs.harmonic_case, 3, T=800, nh=2, freq_scale=1.7)
    bk.compare(bk.run_pairs('hip', c, cd_T=200, hip_kw=kw), bk.run_pairs('port', c, cd_T=200), RTOL, 'harmonic2')


@pytest.mark.parametrize('kw', [pytest.param(WAVE, id='mfma_speculative'), pytest.param(WAVE_X4, id='mfma_four_trials'),
                                pytest.param(WAVE_DPP, id='dpp'), pytest.param(THREAD, id='lane_per_trial')])
def test_frequency_state_crossing_the_softplus_regimes(kw):
    """The wave-per-trial EKF runs 64-step chunks speculatively in the common regime (frequency state >= 6) and repeats a
    chunk with the reference's naive softplus when a step left it: records that start below, cross and re-cross the
    boundary, one that stays below (every chunk repeated, then the checked loop), one that stays above."""
    import math
    from oracle import np_models as om_
    T, dt, Xi = 1500, 1e-3, 0.05
    ts = dt * np.arange(1, T + 1)
    recs = []
    for f_lo, f_hi, seed in [(2.0, 14.0, 1), (9.0, 3.0, 2), (1.0, 2.0, 3), (8.0, 20.0, 4)]:
        freq = f_lo + (f_hi - f_lo) * 0.5 * (1 - np.cos(2 * math.pi * ts / ts[-1] * 1.5))
        phase = np.cumsum(freq) * dt
        recs.append(np.sin(2 * math.pi * phase) + math.sqrt(Xi) * np.random.default_rng(seed).standard_normal(T))
    c = cs.chirp_case(T=8, params=(0.1, 0.5, 0.1, 0.3, 3., float(om_.g_inv(2.0))), Xi=Xi, dt=dt)
    c.ys = np.stack(recs)
    want = bk.run_pairs('port', c, only=('ekf', 'eks'))
    got = bk.run_pairs('hip', c, hip_kw=kw, only=('ekf', 'eks'))
    u2 = want['ekf'][0][:, :, 2]
    assert (u2[0] < 6).any() and (u2[0] > 6).any() and (u2[2] < 6).all() and (u2[3, 200:] > 6).all()
    bk.compare(got, want, RTOL, 'regime_crossing')


def test_chirp_lam0_and_cubature():
    c = cs.chirp_case(T=800, sg='cub', params=(0., 0.3, 0.2, 0.5, 2., 6.))
    bk.compare(bk.run_pairs('hip', c, cd_T=300), bk.run_pairs('port', c, cd_T=300), RTOL, 'lam0')


# ------------------------------------------------------------------ edge cases
@pytest.mark.parametrize('T', [1, 2, 63, 64, 65, 129])
def test_ragged_lengths(T):
    """T = 1 (the smoother returns the filter row), and the 64-step measurement-chunk boundaries."""
    c = cs.chirp_case(T=T, seed=31)
    for kw in (WAVE, THREAD):
        bk.compare(bk.run_pairs('hip', c, hip_kw=kw, only=('ekf', 'sgp_filter')),
                   bk.run_pairs('port', c, only=('ekf', 'sgp_filter')), RTOL, f'T={T}')


@pytest.mark.parametrize('T', [1, 2, 3, 4, 5, 63, 64, 65, 66, 67, 68, 129, 130, 200])
def test_ragged_lengths_in_the_d4_walk_smoothers(T):
    """d = 4 eks / sgp_smoother / rts on the wavefront walk (cgp_walk4.hpp): tiles of 64 steps whose results leave four steps
    per store instruction -- records that end inside a store group, a tile, and the last tile walked past the start of the
    record (its stores wrap out of the buffer windows); one trial of the batch carries NaN from the middle on."""
    c = _batch_case(cs.chirp_case, 3, T=T)
    c.ys = c.ys.copy()
    if T > 8:
        c.ys[1, T // 2] = np.nan
    only = ('ekf', 'eks', 'sgp_filter', 'sgp_smoother')
    want = bk.run_pairs('port', c, only=only)
    got = bk.run_pairs('hip', c, hip_kw=WAVE, only=only)
    bk.compare(got, want, RTOL, f'd4 walk T={T}')
    sm = bk.smoothers_on('hip', c, want, hip_kw=WAVE)
    bk.compare(sm, {k: want[k] for k in ('eks', 'sgp_smoother')}, RTOL, f'd4 walk T={T} smoothers on oracle inputs')
    # linear model, d = 4: rts on the same kernel
    from oracle import np_filters as nf
    fs = _fs()
    rng = np.random.default_rng(1000 + T)
    F = 0.8 * np.eye(4) + 0.05 * rng.standard_normal((4, 4)); Sigma = 0.1 * np.eye(4)      # stable (spectral radius below 0.95 for these seeds): a well-conditioned smoother
    H = np.array([1., 0.5, 0., 0.2]); ys = rng.standard_normal(T)
    wf = nf.kf(F, Sigma, H, 0.3, np.zeros(4), np.eye(4), ys)
    ws = nf.rts(F, Sigma, wf[0], wf[1])
    gs = fs.rts(F, Sigma, wf[0], wf[1], **WAVE)
    npt.assert_allclose(gs[0], ws[0], rtol=1e-9, atol=1e-12)
    npt.assert_allclose(gs[1], ws[1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('T', [1, 2, 3, 33, 63, 64, 65, 66, 97, 129, 130])
def test_ragged_lengths_in_the_tile_layout_kernels(T):
    """d = 6 and d = 8 (cgp_coop8.hpp): the 64-step measurement chunks of the filters, and the cooperative smoother's tiles of
    64 steps walked in two halves of four batches -- records that end inside a batch, a half, a tile; the last tile is walked
    past the start of the record with its loads and stores out of the buffer windows."""
    for nh in (3, 2):
        c = _batch_case(cs.harmonic_case, 3, T=T, nh=nh)
        want = bk.run_pairs('port', c, only=('ekf', 'eks', 'sgp_filter', 'sgp_smoother'))
        got = bk.run_pairs('hip', c, hip_kw=WAVE, only=('ekf', 'eks', 'sgp_filter', 'sgp_smoother'))
        bk.compare(got, want, RTOL, f'nh={nh} T={T}')
        sm = bk.smoothers_on('hip', c, want, hip_kw=WAVE)
        bk.compare(sm, {k: want[k] for k in ('eks', 'sgp_smoother')}, RTOL, f'nh={nh} T={T} smoothers on oracle inputs')


@pytest.mark.parametrize('B', [1, 63, 65, 130])
def test_ragged_batches(B):
    """Partial last wavefront in the one-lane-per-trial shape."""
    c = _batch_case(cs.chirp_case, B, T=40)
    bk.compare(bk.run_pairs('hip', c, hip_kw=THREAD, only=('ekf', 'cd_ekf')),
               bk.run_pairs('port', c, only=('ekf', 'cd_ekf')), RTOL, f'B={B}')


def test_empty_inputs():
    fs = _fs()
    c = cs.chirp_case(T=8)
    mfs, Pfs, nll = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, np.zeros((0,)))
    assert mfs.shape == (0, 4) and Pfs.shape == (0, 4, 4) and nll.shape == (0,)
    mfs, Pfs, nll = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, np.zeros((0, 5)))
    assert mfs.shape == (0, 5, 4)


def test_per_trial_parameters_and_nll_only():
    """Parameter sweep (config C5 / MLE objective): per-trial model parameters, m0, P0; nll-only output."""
    from chirpgp_amd import models as pm
    from oracle import port
    fs = _fs()
    B, T = 9, 500
    rng = np.random.default_rng(3)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * rng.uniform(0.7, 1.3, size=(B, 6))
    drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
    ys = np.stack([cs.chirp_measurements(T, 200 + i)[2] for i in range(B)])
    want = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 1e-3, ys)
    for kw in (WAVE, THREAD):
        got = fs.ekf(disc, H, 0.1, m0, P0, 1e-3, ys, **kw)
        for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g, w, RTOL, f'sweep.{n}')
        last = fs.ekf(disc, H, 0.1, m0, P0, 1e-3, ys, nll_final_only=True, want=(False, False, True), **kw)
        assert last[0] is None and last[1] is None
        # the same recursion, except that the matrix-core EKF keeps an NLL-only launch on its common-regime polynomials (the
        # objective of a finite-difference gradient must not change regime between probes): roundings apart
        npt.assert_allclose(last[2], got[2][:, -1], rtol=1e-12)


def test_nan_semantics():
    """Indefinite P0: NaN exactly where the oracle has NaN (everywhere), no exception, run continues."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    fs = _fs()
    sg = SigmaPoints.cubature(2)
    spec = pm.linear_cond_m_cov(0.9 * np.eye(2), 0.01 * np.eye(2))
    P0 = np.array([[1., 2.], [2., 1.]])
    args = (np.array([1., 0.]), 0.1, np.zeros(2), P0, 0.1, np.ones(5))
    want = port.filter(port.F_SGP, spec, sg, *args)
    for kw in (WAVE, THREAD):
        got = fs.sgp_filter(spec, sg, *args, **kw)
        for g, w in zip(got, want):
            assert np.array_equal(np.isnan(g), np.isnan(w)) and np.all(np.isnan(g))
    # a batch where only one trial breaks down: the others are untouched
    P0b = np.stack([np.eye(2), P0, 2 * np.eye(2)])
    ysb = np.ones((3, 5))
    got = fs.sgp_filter(spec, sg, args[0], 0.1, np.zeros(2), P0b, 0.1, ysb)
    assert np.all(np.isnan(got[0][1])) and not np.any(np.isnan(got[0][0])) and not np.any(np.isnan(got[0][2]))


def test_torch_tensors_stay_on_device():
    import torch
    fs = _fs()
    c = cs.chirp_case(T=100)
    ys = torch.from_numpy(c.ys).cuda()
    mfs, Pfs, nll = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    assert mfs.is_cuda and Pfs.is_cuda and nll.is_cuda and mfs.dtype == torch.float64
    mss, Pss = fs.eks(c.disc, mfs, Pfs, c.dt)
    assert mss.is_cuda and tuple(Pss.shape) == (100, 4, 4)


def test_gaussian_expectation_on_device():
    from chirpgp_amd.quadratures import gaussian_expectation, SigmaPoints
    from chirpgp_amd.models import g
    from oracle import np_quadratures as oq
    rng = np.random.default_rng(5)
    ms, sd = rng.standard_normal(1000) * 3 + 5, rng.uniform(0.05, 1.5, 1000)
    got = gaussian_expectation(ms, sd, func=g, force_shape=True)
    want = oq.gaussian_expectation(ms[:50], sd[:50], force_shape=True)
    npt.assert_allclose(got[:50], want, rtol=1e-12)
    sg = SigmaPoints.gauss_hermite(1, 10)
    npt.assert_allclose(got[:, 0], (np.log1p(np.exp(ms[:, None] + sd[:, None] * sg.xi[:, 0][None, :])) * sg.w).sum(1), rtol=1e-12)


# ------------------------------------------------------------------ full benchmark size: size-independent properties
def test_full_size_properties_and_parity():
    """BASELINE config C2 at full size (d = 4, T = 10 000, B = 1000) on device-resident data:
    the whole EKF + EKS result against the C port (which finishes this in seconds on the host cores)."""
    import torch
    from chirpgp_amd import models as pm
    from oracle import port
    import bench
    fs = _fs()
    B, T = 1000, 10000
    wl = bench.make_workload(B, T, seed=0)
    ys = torch.from_numpy(wl['ys']).cuda()
    mfs, Pfs, nll = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
    mss, Pss = fs.eks(wl['disc'], mfs, Pfs, wl['dt'])
    assert bool(torch.isfinite(mfs).all()) and bool(torch.isfinite(Pss).all())
    # N5: last smoothing row == last filtering row, bit for bit
    assert torch.equal(mss[:, -1], mfs[:, -1]) and torch.equal(Pss[:, -1], Pfs[:, -1])
    # cumulative nll is non-decreasing only in expectation; but it must equal the running sum of its own increments
    # symmetric covariances
    # symmetric covariances, to rounding: the lane-cooperative kernels compute entries (i, j) and (j, i) in different lanes, which
    # sum in different orders -- like the reference's own (F P) F^T and G X G^T, which are not bit-symmetric either
    scale = Pfs.abs().amax(dim=(-1, -2), keepdim=True)
    assert float(((Pfs - Pfs.transpose(-1, -2)).abs() / scale).max()) < 1e-12
    assert float(((Pss - Pss.transpose(-1, -2)).abs() / scale).max()) < 1e-12
    # smoothing never increases the marginal variance (up to rounding)
    dvar = torch.diagonal(Pfs - Pss, dim1=-2, dim2=-1)
    assert float(dvar.min()) > -1e-9 * float(torch.diagonal(Pfs, dim1=-2, dim2=-1).abs().max())
    # EVERY trial against the port (slabs of 250 trials bound the host memory; the port runs all 1000 x 10 000 in about a second)
    worst = dict.fromkeys(('mfs', 'Pfs', 'nll', 'mss', 'Pss'), 0.0)
    for lo in range(0, B, 250):
        sl = slice(lo, lo + 250)
        w_f = port.filter(port.F_EKF, wl['disc'], None, wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], wl['ys'][sl])
        w_s = port.smoother(port.S_EKS, wl['disc'], None, wl['dt'], w_f[0], w_f[1])
        for g, w, n in zip((mfs[sl], Pfs[sl], nll[sl], mss[sl], Pss[sl]), w_f + w_s, worst):
            g = g.cpu().numpy()
            cs.assert_close(g, w, RTOL, f'full.{n}')
            worst[n] = max(worst[n], cs.max_rel_err(g, w))
    print({n: f'{e:.2e}' for n, e in worst.items()})
    # the bench kernels spend accuracy nobody asked for on a shorter chain (lean softplus polynomials, one Newton step on
    # the reciprocals: cgp_fastmath.hpp) -- but no more than this: 1e-9 relative over the whole record, ALL 1000 trials checked
    for n, err in worst.items():
        assert err <= 1e-9, (n, err)


def test_mle_through_the_filter():
    """'Next' row 8f-1: L-BFGS-B on the engine's batched NLL-only objective (value + central differences = one launch).
    The batched objective must equal the filter's own last cumulative NLL, its gradient must match a finer difference
    quotient, and the fit must lower the NLL and give a sane frequency RMSE on the toy chirp (demos/ekfs_mle.py)."""
    from chirpgp_amd import mle, models as pm
    from chirpgp_amd.quadratures import gaussian_expectation
    from chirpgp_amd.toymodels import gen_chirp, meow_freq, constant_mag
    from chirpgp_amd.tools import rmse
    fs = _fs()
    dt, T, Xi = 1e-3, 3141, 0.1
    ts = np.linspace(dt, dt * T, T)
    freq, phase = meow_freq(offset=8.)
    ys = gen_chirp(ts, constant_mag(1.), phase) + np.sqrt(Xi) * np.random.default_rng(555).standard_normal(T)
    init = np.array([0.1, 0.1, 0.1, 1., 1., 7.])
    fun = mle.make_objective('ekf', pm.build_chirp_model, ys, Xi, dt)
    f0, g0 = fun(pm.g_inv(init))
    _, _, disc, m0, P0, H = pm.build_chirp_model(init)
    npt.assert_allclose(f0, fs.ekf(disc, H, Xi, m0, P0, dt, ys)[2][-1], rtol=1e-10)
    th = pm.g_inv(init)
    for i in (0, 3, 5):
        e = np.zeros(6)
        e[i] = 1e-4
        fd = (fun(th + e)[0] - fun(th - e)[0]) / 2e-4
        npt.assert_allclose(g0[i], fd, rtol=1e-3, atol=1e-3)
    opt, res = mle.fit('ekf', pm.build_chirp_model, init, ys, Xi, dt, maxiter=60)
    assert res.fun < f0 - 1.0 and np.all(np.isfinite(opt)) and np.all(opt > 0)
    _, _, disc, m0, P0, H = pm.build_chirp_model(opt)
    mfs, Pfs, _ = fs.ekf(disc, H, Xi, m0, P0, dt, ys)
    mss, Pss = fs.eks(disc, mfs, Pfs, dt)
    est = gaussian_expectation(mss[:, 2], np.sqrt(Pss[:, 2, 2]), func=pm.g, force_shape=True)[:, 0]
    assert rmse(freq(ts), est) < 3.0, rmse(freq(ts), est)


@pytest.mark.parametrize('d', [1, 2, 3, 4, 5, 6, 7, 8])
def test_linear_models_all_dims(d):
    """kf / rts / sgp / cd_* on random stable linear models of every compiled dimension, against the C port."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    rng = np.random.default_rng(40 + d)
    A = -np.eye(d) * rng.uniform(0.5, 2.0, d) + 0.3 * rng.standard_normal((d, d))
    A = A - np.eye(d) * max(0.0, np.max(np.real(np.linalg.eigvals(A))) + 0.2)
    Bm = 0.5 * np.eye(d) + 0.1 * rng.standard_normal((d, d))
    dt = 0.01
    from chirpgp_amd.tools import lti_sde_to_disc
    F, Sigma = lti_sde_to_disc(A, Bm, dt)
    Sigma = 0.5 * (Sigma + Sigma.T)
    H = rng.standard_normal(d)
    m0, P0 = rng.standard_normal(d), np.eye(d) * 0.3
    T = 300
    ys = rng.standard_normal(T)
    drift, disp = pm.linear_sde(A, Bm)
    c = cs.Case(f'lin{d}', d=d, dt=dt, H=H, Xi=0.2, m0=m0, P0=P0, ys=ys, disc=pm.linear_cond_m_cov(F, Sigma),
                drift=drift, disp=disp, sgps=SigmaPoints.cubature(d))
    want = bk.run_pairs('port', c)
    for kw in (WAVE, THREAD):
        bk.compare(bk.run_pairs('hip', c, hip_kw=kw), want, RTOL, f'linear d={d}')


def test_large_sigma_sets_fall_back_to_one_lane_per_trial():
    """A sigma-point set too large for the LDS stage (Gauss-Hermite order 3 in d = 8: 6561 points) must still be served
    (by the one-lane-per-trial kernels that read it from global memory), and a mid-size one (d = 6, 729 points) staged."""
    for nh, T in ((3, 12), (2, 40)):
        from chirpgp_amd.quadratures import SigmaPoints
        c = cs.harmonic_case(T=T, nh=nh, seed=51)
        c.sgps = SigmaPoints.gauss_hermite(2 * nh + 2, 3)
        got = bk.run_pairs('hip', c, only=('sgp_filter', 'sgp_smoother'))
        want = bk.run_pairs('port', c, only=('sgp_filter', 'sgp_smoother'))
        bk.compare(got, want, RTOL, f'gh3 d={2 * nh + 2}')


def test_d4_sigma_sets_in_the_window_between_the_lane_and_the_wave_stage():
    """the large-batch lane kernel of the sigma-point filter has 28.7 KB of static LDS in front of the staged set, so a set of
    34 .. 44 KB (about 880 .. 1120 points at d = 4) fits the wave kernels' stage but not the lane kernel's 64 KB launch -- it must take the
    generic lane kernel (set read from global memory), not fail.  881 points (Gauss-Hermite orders 5 and 4 at half weight each), a smaller
    one (order 5: 625 points) on the lane kernel proper, and a larger one (order 6: 1296) past both stages; every launch shape."""
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    fs = _fs()
    c = cs.chirp_case(T=40, seed=52)
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(1).standard_normal((70, c.ys.size))
    g5, g4, g6 = SigmaPoints.gauss_hermite(4, 5), SigmaPoints.gauss_hermite(4, 4), SigmaPoints.gauss_hermite(4, 6)
    mixed = SigmaPoints(4, g5.n_points + g4.n_points, np.concatenate([0.5 * g5.w, 0.5 * g4.w]), None, np.concatenate([g5.xi, g4.xi]))
    for sg, tag in ((mixed, '881 points'), (g5, '625 points'), (g6, '1296 points')):
        want = port.filter(port.F_SGP, c.disc, sg, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
        for kw in (THREAD, WAVE, {}):
            got = fs.sgp_filter(c.disc, sg, c.H, c.Xi, c.m0, c.P0, c.dt, ys, **kw)
            for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
                cs.assert_close(g, w, RTOL, f'{tag} {kw} {n}')


def test_sigma_point_sweep_nll_only():
    """Config C5's sweep: 3-harmonic model, cubature, one parameter vector per trial, NLL-only, both launch shapes."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    fs = _fs()
    G, T = 7, 300
    rng = np.random.default_rng(9)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * rng.uniform(0.8, 1.2, size=(G, 6))
    drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, 3)
    sg = SigmaPoints.cubature(8)
    ys = np.tile(cs.chirp_measurements(T, 77, num_harmonics=3)[2], (G, 1))
    want = port.filter(port.F_SGP, disc, sg, H, 0.1, m0, P0, 1e-3, ys, nll_final_only=True)[2]
    for kw in (WAVE, THREAD):
        got = fs.sgp_filter(disc, sg, H, 0.1, m0, P0, 1e-3, ys, nll_final_only=True, want=(False, False, True), **kw)[2]
        cs.assert_close(got, want, RTOL, 'sweep nll')


def test_c_abi_argument_errors():
    """Misuse through the C-ABI returns CGP_E_ARG / CGP_E_UNSUPPORTED with a message, never a launch."""
    import ctypes as C
    import torch
    from chirpgp_amd import _engine as E
    lib, ctx = E.load_library(), E.context()
    d = 4
    params = torch.zeros(5, dtype=torch.float64, device='cuda')
    buf = torch.zeros(64, dtype=torch.float64, device='cuda')
    model = E.CgpModel(E.M_HARMONIC_LCD, d, 1, 5, params.data_ptr(), 0, None, 0)
    init = E.CgpInit(buf.data_ptr(), 0, buf.data_ptr(), 0, buf.data_ptr(), 0, buf.data_ptr(), 0)
    args = (C.byref(init), 0.1, buf.data_ptr(), 4, 1, None, 1, 4, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 0, None)
    assert lib.cgp_filter(ctx, E.F_CD_EKF, C.byref(model), None, *args) == -1            # SDE method, discrete model
    assert b'SDE model' in lib.cgp_last_error(ctx)
    assert lib.cgp_filter(ctx, E.F_SGP, C.byref(model), None, *args) == -1               # sigma method without points
    assert lib.cgp_filter(ctx, 99, C.byref(model), None, *args) == -1
    bad = E.CgpModel(E.M_HARMONIC_LCD, 6, 1, 5, params.data_ptr(), 0, None, 0)           # d inconsistent with n_harm
    assert lib.cgp_filter(ctx, E.F_EKF, C.byref(bad), None, *args) == -1
    lin9 = E.CgpModel(E.M_LINEAR, 9, 0, 162, params.data_ptr(), 0, None, 0)
    assert lib.cgp_filter(ctx, E.F_EKF, C.byref(lin9), None, *args) == -2                # not compiled in
    assert b'dimension' in lib.cgp_last_error(ctx)
    assert lib.cgp_filter(ctx, E.F_EKF, C.byref(model), None, C.byref(init), 0.1, buf.data_ptr(), 4, 1, None, 0, 4, None, None, None, 0, None) == 0   # B = 0
    from chirpgp_amd import filters_smoothers as fs
    with pytest.raises(NotImplementedError):
        fs.kf(np.eye(9), np.eye(9), np.ones(9), 0.1, np.zeros(9), np.eye(9), np.zeros(5))


def test_filter_means_only_large_batch():
    """The CRLB job's shape (tetralith/jobs/crlb_ekf.py): many short records, only the filtering means kept."""
    from chirpgp_amd.models import model_chirp, disc_chirp_lcd
    from oracle import port
    fs = _fs()
    _, _, m0, P0, H = model_chirp(0.1, 0.1, 1., 1., 0.1)
    disc = disc_chirp_lcd(0.1, 0.1, 1., 1.)
    B, T = 3000, 50
    ys = np.random.default_rng(1).standard_normal((B, T))
    mfs, Pfs, nll = fs.ekf(disc, H, 0.1, m0, P0, 0.01, ys, want=(True, False, False))
    assert Pfs is None and nll is None and mfs.shape == (B, T, 4)
    want = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 0.01, ys[::100])[0]
    cs.assert_close(mfs[::100], want, RTOL, 'means only')


@pytest.mark.parametrize('kw', SHAPES)
def test_per_trial_measurement_model(kw):
    """H, Xi, m0, P0 all with a leading batch axis (strides in cgp_init), through the cooperative and generic kernels."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    import copy
    fs = _fs()
    B, T = 5, 260
    rng = np.random.default_rng(12)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * rng.uniform(0.9, 1.1, size=(B, 6))
    drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
    Hb = np.tile(H, (B, 1)) * rng.uniform(0.8, 1.2, size=(B, 1)) + 0.05 * rng.standard_normal((B, 4))
    Xib = rng.uniform(0.05, 0.2, size=B)
    ys = np.stack([cs.chirp_measurements(T, 400 + i)[2] for i in range(B)])
    sg = SigmaPoints.gauss_hermite(4, 3)
    dg = copy.copy(drift)
    dg.gamma = disp.outer()
    pairs = [
        (lambda: fs.ekf(disc, Hb, Xib, m0, P0, 1e-3, ys, **kw), lambda: port.filter(port.F_EKF, disc, None, Hb, Xib, m0, P0, 1e-3, ys)),
        (lambda: fs.sgp_filter(disc, sg, Hb, Xib, m0, P0, 1e-3, ys, **kw), lambda: port.filter(port.F_SGP, disc, sg, Hb, Xib, m0, P0, 1e-3, ys)),
        (lambda: fs.cd_ekf(drift, disp, Hb, Xib, m0, P0, 1e-3, ys, **kw), lambda: port.filter(port.F_CD_EKF, dg, None, Hb, Xib, m0, P0, 1e-3, ys)),
        (lambda: fs.cd_sgp_filter(drift, disp, sg, Hb, Xib, m0, P0, 1e-3, ys[:, :80], **kw),
         lambda: port.filter(port.F_CD_SGP, dg, sg, Hb, Xib, m0, P0, 1e-3, ys[:, :80])),
    ]
    for i, (got, want) in enumerate(pairs):
        for g, w, n in zip(got(), want(), ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g, w, RTOL, f'pair{i}.{n}')
    # NLL-only through the cooperative sigma-point kernel
    last = fs.sgp_filter(disc, sg, Hb, Xib, m0, P0, 1e-3, ys, nll_final_only=True, want=(False, False, True), **kw)[2]
    cs.assert_close(last, port.filter(port.F_SGP, disc, sg, Hb, Xib, m0, P0, 1e-3, ys, nll_final_only=True)[2], RTOL, 'nll-only')


def test_disc_m32_matches_kf():
    """models.py:408-416: the exact Matern-3/2 discretisation as cond_m_cov gives the Kalman filter of its (F, Sigma)."""
    from chirpgp_amd import models as pm
    fs = _fs()
    spec = pm.disc_m32(1.1, 2.2)
    F, Sigma = spec(np.eye(2)[0], 1e-2)[0], spec(np.zeros(2), 1e-2)[1]
    F = np.stack([spec(np.eye(2)[j], 1e-2)[0] for j in range(2)], axis=1)
    ys = np.random.default_rng(2).standard_normal(200)
    H, m0, P0 = np.array([1., 0.]), np.zeros(2), np.eye(2)
    a = fs.ekf(spec, H, 0.5, m0, P0, 1e-2, ys)
    b = fs.kf(F, Sigma, H, 0.5, m0, P0, ys)
    for x, y in zip(a, b):
        npt.assert_allclose(x, y, rtol=1e-12, atol=1e-14)
    npt.assert_allclose(fs.eks(spec, a[0], a[1], 1e-2)[0], fs.rts(F, Sigma, b[0], b[1])[0], rtol=1e-12, atol=1e-14)


def test_lockstep_mle_matches_per_record_fits():
    """fit_many (R records, one launch per probe) reaches the optimum SciPy's L-BFGS-B finds record by record."""
    from chirpgp_amd import mle, models as pm
    T, R = 1200, 5
    recs = np.stack([cs.chirp_case(T=T, seed=40 + i).ys for i in range(R)])
    init = [0.1, 0.1, 0.1, 1., 1., 7.]
    many, info = mle.fit_many('ekf', pm.build_chirp_model, init, recs, 0.1, 1e-3, maxiter=120)
    assert many.shape == (R, 6) and info['launches'] < 400
    for r in range(R):
        single, res = mle.fit('ekf', pm.build_chirp_model, init, recs[r], 0.1, 1e-3, maxiter=120)
        assert info['fun'][r] <= res.fun + 1e-3 * abs(res.fun), (r, info['fun'][r], res.fun)
        assert abs(info['fun'][r] - res.fun) <= 1e-6 * abs(res.fun)      # (measured 4e-10 .. 2e-8; the gate was 2e-2 up to round 5)


def test_four_trials_per_wave_dense_variant_at_large_batch():
    """Above 4096 trials the four-trials-per-wave EKF runs its two-waves-per-SIMD build: same results as one lane per
    trial (itself checked against the oracle above), ragged last wave included."""
    fs = _fs()
    c = cs.chirp_case(T=8)
    B, T = 4102, 130
    rng = np.random.default_rng(3)
    ys = np.sin(0.05 * np.arange(T))[None, :] * rng.uniform(0.5, 1.5, (B, 1)) + 0.3 * rng.standard_normal((B, T))
    a = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, ys, flags=0x2)
    b = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, ys, flags=0x4)
    for x, y in zip(a, b):
        npt.assert_allclose(x, y, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('kw', [pytest.param(WAVE, id='one_trial_per_wave'), pytest.param(WAVE_X4, id='four_trials_per_wave'),
                                pytest.param(THREAD, id='lane_per_trial')])
def test_nan_in_one_trial_of_the_chirp_ekf(kw):
    """A NaN measurement in the middle of one record, and an indefinite P0 in another: those trials turn NaN exactly where
    the oracle's do; their neighbours -- which share a wavefront with them in the four-trials-per-wave kernel, so their
    chunks are repeated on the checked step -- are unaffected."""
    c = _batch_case(cs.chirp_case, 6, T=700)
    c.ys = c.ys.copy()
    c.ys[1, 333] = np.nan
    P0 = np.repeat(np.asarray(c.P0)[None], 6, axis=0)
    P0[4] = np.array([[1., 2., 0, 0], [2., 1., 0, 0], [0, 0, 1., 0], [0, 0, 0, 1.]])
    c.P0 = P0
    from oracle import port
    want = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    got = _fs().ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys, **kw)
    for g, w in zip(got, want):
        assert np.array_equal(np.isnan(g), np.isnan(w))
        ok = ~np.isnan(w)
        npt.assert_allclose(g[ok], w[ok], rtol=RTOL, atol=1e-12)
    assert np.isnan(got[0][1, 333:]).all() and not np.isnan(got[0][1, :333]).any()
    assert not np.isnan(got[0][[0, 2, 3, 5]]).any()


@pytest.mark.parametrize('kw', [pytest.param(WAVE, id='matrix_core'), pytest.param(WAVE_DPP, id='lds_reduced')])
def test_nan_in_the_d4_sigma_point_kernels(kw):
    """sgp_filter / cd_sgp_filter / cd_sgp_smoother, d = 4, on the matrix-core kernels (cgp_mfma4_sigma.hpp, cgp_mfma4_cd.hpp)
    and on the LDS-reduced ones: a NaN measurement in one record, a P0 whose FIRST pivot fails in another, and one whose
    LAST pivot alone fails (the collapsed quadratures never take that pivot's square root: the kernels have to notice it) --
    those trials are NaN exactly where the C port's are, the others are untouched."""
    c = _batch_case(cs.chirp_case, 6, T=260)
    c.ys = c.ys.copy()
    c.ys[1, 130] = np.nan
    P0 = np.repeat(np.asarray(c.P0)[None], 6, axis=0)
    P0[2] = np.array([[1., 2., 0, 0], [2., 1., 0, 0], [0, 0, 1., 0], [0, 0, 0, 1.]])
    P0[4] = np.diag([1., 1., 1., -0.5])
    c.P0 = P0
    only = ('sgp_filter', 'sgp_smoother', 'cd_sgp_filter', 'cd_sgp_smoother')
    want = bk.run_pairs('port', c, only=only)
    got = bk.run_pairs('hip', c, hip_kw=kw, only=only)
    bk.compare(got, want, RTOL, 'nan_d4_sigma')
    for k in ('sgp_filter', 'cd_sgp_filter'):
        m = got[k][0]
        assert np.isnan(m[2]).all() and np.isnan(m[4]).all() and np.isnan(m[1, 130:]).all() and not np.isnan(m[1, :130]).any()
        assert not np.isnan(m[[0, 3, 5]]).any()


@pytest.mark.parametrize('T', [1, 2, 63, 64, 65, 129])
def test_ragged_lengths_cd_sigma_point(T):
    """The 64-step chunks of the matrix-core cd_sgp filter (measurements, NLL latch) and smoother (backward gains)."""
    c = cs.chirp_case(T=T, seed=37)
    only = ('cd_sgp_filter', 'cd_sgp_smoother')
    bk.compare(bk.run_pairs('hip', c, hip_kw=WAVE, only=only), bk.run_pairs('port', c, only=only), RTOL, f'cd T={T}')


# ------------------------------------------------------------------ BASELINE config C1
@pytest.mark.parametrize('kw', [pytest.param(WAVE, id='wave_per_trial'), pytest.param(WAVE_SEQ, id='wave_sequential_scan'),
                                pytest.param(THREAD, id='lane_per_trial'), pytest.param({}, id='default_shape')])
def test_config_c1_linear_kf_rts_on_the_toy_chirp(kw):
    """BASELINE configs[0]: linear KF + RTS, d = 4, T = 1000, batch = 1 on the toy chirp -- the chirp LCD model with its
    frequency frozen at the initial value (filters_smoothers.py:145-219; F from models.py:296-301, Sigma from :302-308),
    against the NumPy oracle's kf / rts at 1e-9 in every launch shape, and against the C port."""
    import bench
    from oracle import np_filters as nf
    from oracle import port
    fs = _fs()
    params, dt, T = [0.1, 0.1, 0.1, 1., 1., 7.], 1e-3, 1000
    F, Sigma = bench.frozen_frequency_linear_model(params, dt)
    c = cs.chirp_case(T=T, seed=71, params=tuple(params), dt=dt)
    # the frozen-frequency F is the EKF's mean map at the initial state: F m0 == cond mean(m0)
    npt.assert_allclose(F @ c.m0, c.o_disc(c.m0, dt)[0], rtol=1e-14, atol=1e-15)
    npt.assert_allclose(Sigma, c.o_disc(c.m0, dt)[1], rtol=1e-14, atol=0)
    want_f = nf.kf(F, Sigma, c.H, c.Xi, c.m0, c.P0, c.ys)
    want_s = nf.rts(F, Sigma, want_f[0], want_f[1])
    got_f = fs.kf(F, Sigma, c.H, c.Xi, c.m0, c.P0, c.ys, **kw)
    got_s = fs.rts(F, Sigma, got_f[0], got_f[1], **kw)
    assert got_f[0].shape == (T, 4) and got_f[1].shape == (T, 4, 4) and got_f[2].shape == (T,) and got_s[1].shape == (T, 4, 4)
    worst = bk.compare({'kf': got_f, 'rts': got_s}, {'kf': want_f, 'rts': want_s}, 1e-9, 'C1')
    print({k: f'{v:.1e}' for k, v in worst.items()})
    npt.assert_array_equal(got_s[0][-1], got_f[0][-1])
    npt.assert_array_equal(got_s[1][-1], got_f[1][-1])
    # smoother on the oracle's filtering results (identical inputs), and the second checker
    on_oracle = fs.rts(F, Sigma, want_f[0], want_f[1], **kw)
    bk.compare({'rts': on_oracle}, {'rts': want_s}, 1e-9, 'C1/rts_on_oracle_inputs')
    from chirpgp_amd import models as pm
    pf = port.filter(port.F_EKF, pm.linear_cond_m_cov(F, Sigma), None, c.H, c.Xi, c.m0, c.P0, dt, c.ys)
    bk.compare({'kf': got_f}, {'kf': pf}, 1e-9, 'C1/port')


def test_last_error_is_per_thread():
    """Host threads sharing one context: each sees the message of ITS last failed call (include/chirpgp_hip.h)."""
    import ctypes as C
    import threading
    import torch
    from chirpgp_amd import _engine as E
    lib, ctx = E.load_library(), E.context()
    params = torch.zeros(200, dtype=torch.float64, device='cuda')
    buf = torch.zeros(64, dtype=torch.float64, device='cuda')
    init = E.CgpInit(buf.data_ptr(), 0, buf.data_ptr(), 0, buf.data_ptr(), 0, buf.data_ptr(), 0)
    args = (C.byref(init), 0.1, buf.data_ptr(), 4, 1, None, 1, 4, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 0, None)
    lcd = E.CgpModel(E.M_HARMONIC_LCD, 4, 1, 5, params.data_ptr(), 0, None, 0)
    lin9 = E.CgpModel(E.M_LINEAR, 9, 0, 162, params.data_ptr(), 0, None, 0)
    seen, barrier = {}, threading.Barrier(2)

    def worker(name, method, model, needle):
        ok = True
        for _ in range(200):
            rc = lib.cgp_filter(ctx, method, C.byref(model), None, *args)
            barrier.wait()
            msg = lib.cgp_last_error(ctx)
            ok = ok and rc < 0 and needle in msg
            barrier.wait()
        seen[name] = ok
    t1 = threading.Thread(target=worker, args=('a', E.F_CD_EKF, lcd, b'SDE model'))
    t2 = threading.Thread(target=worker, args=('b', E.F_EKF, lin9, b'dimension'))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert seen == {'a': True, 'b': True}
    assert torch.cuda.current_device() == 0


@pytest.mark.parametrize('kw', [pytest.param(THREAD, id='lane_per_trial'), pytest.param(WAVE, id='wave_per_trial')])
def test_wide_sigma_fans_take_the_full_sincos_fallback(kw):
    """The lane-per-trial and time-parallel sigma-point paths anchor the rotation at the mean and rotate each group of
    points by the small angle d = dt (w - w0), falling back to a full sincos where |d| > 1/16 (cgp_models.hpp).  A batch
    that mixes narrow fans with wide ones (large P0[2][2], freq_scale 1e3 as in the reference's real_applications) makes
    some lanes take the fallback and others not; one trial carries a NaN measurement."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    fs = _fs()
    B, T, dt = 6, 400, 1e-3
    rng = np.random.default_rng(77)
    base = np.array([0.1, 0.1, 0.1, 1., 1., 7.])
    for nh, sg, fscale in ((1, SigmaPoints.gauss_hermite(4, 3), 1.0), (1, SigmaPoints.cubature(4), 1.0), (2, SigmaPoints.cubature(6), 1e3)):
        params = base * rng.uniform(0.9, 1.1, size=(B, 6))
        if fscale == 1.0:
            drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        else:
            params[:, 5] = pm.g_inv(np.array([0.004, 0.006, 0.008, 0.01, 0.012, 0.02]))     # kHz-scale frequencies
            drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, nh, fscale)
        P0 = np.array(P0)
        wide = np.array([1., 400., 1., 2500., 1., 30.])                     # P0[2][2] scale per trial: narrow and wide fans
        P0[:, -2, -2] *= wide if fscale == 1.0 else np.array([0.01, 1., 0.01, 1., 0.01, 1.])
        ys = np.stack([cs.chirp_measurements(T, 900 + i, dt=dt, num_harmonics=0 if nh == 1 else nh)[2] for i in range(B)])
        ys[3, 250] = np.nan
        # the fans really are wide: dt * 2 pi fs * (g(m + 3 sd) - g(m)) beyond 1/16 for some trials, inside for others
        sd = np.sqrt(P0[:, -2, -2])
        spread = dt * 2 * np.pi * fscale * (pm.g(m0[:, -2] + np.sqrt(3.) * sd) - pm.g(m0[:, -2]))
        assert (spread > 0.0625).any() and (spread < 0.0625).any(), spread
        want_f = port.filter(port.F_SGP, disc, sg, H, 0.1, m0, P0, dt, ys)
        want_s = port.smoother(port.S_SGP, disc, sg, dt, want_f[0], want_f[1])
        got_f = fs.sgp_filter(disc, sg, H, 0.1, m0, P0, dt, ys, **kw)
        got_s = fs.sgp_smoother(disc, sg, want_f[0], want_f[1], dt, **kw)      # on the oracle's filtering results
        for g_, w_, n in zip(got_f + got_s, want_f + want_s, ('mfs', 'Pfs', 'nll', 'mss', 'Pss')):
            cs.assert_close(g_, w_, RTOL, f'wide fan nh={nh} fs={fscale}: {n}')
        assert np.isnan(got_f[0][3, 250:]).all() and not np.isnan(got_f[0][[0, 1, 2, 4, 5]]).any()


@pytest.mark.parametrize('nh', [2, 3])
@pytest.mark.parametrize('kw', [pytest.param(WAVE, id='tile_layout'), pytest.param(WAVE_GENERIC, id='generic'), pytest.param(THREAD, id='lane_per_trial')])
def test_harmonic_models_per_trial_everything_and_nan(nh, kw):
    """d = 6 / 8 (the tile-layout kernels and their generic counterparts): one parameter vector, H, Xi, m0, P0 per trial; a NaN
    measurement in one record and an indefinite P0 in another turn exactly those trials NaN where the oracle's do; NLL-only."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    fs = _fs()
    B, T, d = 5, 300, 2 * nh + 2
    rng = np.random.default_rng(21 + nh)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * rng.uniform(0.9, 1.1, size=(B, 6))
    drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, nh)
    Hb = np.tile(H, (B, 1)) * rng.uniform(0.8, 1.2, size=(B, 1)) + 0.05 * rng.standard_normal((B, d))
    Xib = rng.uniform(0.05, 0.2, size=B)
    P0 = np.array(P0)
    P0[3, 0, 1] = P0[3, 1, 0] = 2.0 * P0[3, 0, 0]                      # indefinite: trial 3 is NaN from the first Cholesky on
    ys = np.stack([cs.chirp_measurements(T, 500 + i, num_harmonics=nh)[2] for i in range(B)])
    ys[1, 170] = np.nan
    sg = SigmaPoints.cubature(d)
    for meth, pmeth, smeth, psmeth, sgp in (('ekf', port.F_EKF, 'eks', port.S_EKS, None), ('sgp_filter', port.F_SGP, 'sgp_smoother', port.S_SGP, sg)):
        want = port.filter(pmeth, disc, sgp, Hb, Xib, m0, P0, 1e-3, ys)
        args = (disc,) + ((sg,) if sgp is not None else ()) + (Hb, Xib, m0, P0, 1e-3, ys)
        got = getattr(fs, meth)(*args, **kw)
        for g_, w_, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g_, w_, RTOL, f'{meth} nh={nh} {n}')
        assert np.isnan(got[0][1, 170:]).all() and not np.isnan(got[0][[0, 2, 4]]).any()
        assert np.isnan(got[0][3]).all() == (meth == 'sgp_filter')       # only the sigma-point filter factorises P (the EKF does not)
        last = getattr(fs, meth)(*args, nll_final_only=True, want=(False, False, True), **kw)[2]
        cs.assert_close(last, want[2][:, -1], RTOL, f'{meth} nh={nh} nll-only')
        # smoothers on the oracle's filtering results of the healthy trials (and the NaN ones: NaN in, NaN out)
        want_s = port.smoother(psmeth, disc, sgp, 1e-3, want[0], want[1])
        sargs = (disc,) + ((sg,) if sgp is not None else ()) + (want[0], want[1], 1e-3)
        got_s = getattr(fs, smeth)(*sargs, **kw)
        for g_, w_, n in zip(got_s, want_s, ('mss', 'Pss')):
            cs.assert_close(g_, w_, RTOL, f'{smeth} nh={nh} {n}')


# ------------------------------------------------------------------ BASELINE configs C3, C4, C5 at their full per-GPU size
@pytest.mark.parametrize('kind,every,batch', [pytest.param('sgp', 1, None, id='C3_gh3_d4_1000x10000'), pytest.param('harmonic', 1, None, id='C5_cubature_d8_1000x10000'),
                                              pytest.param('cd_sgp', 1, None, id='C4_cd_gh3_d4_512x50000'),
                                              pytest.param('cd_sgp', 64, 4096, id='C4_whole_on_one_gpu_4096x50000')])
def test_full_size_sigma_point_configs(kind, every, batch):
    """The sigma-point configurations at the size bench.py times them (B x T per GPU of BASELINE.json), results resident in HBM:
    EVERY trial against the C port over the whole record (every == 1), plus the size-independent properties on all trials
    (finite; last smoothing row == last filtering row bit for bit; the discrete smoothers do not increase the marginal variance).
    C4 also whole on one GPU (4096 x 50 000: DESIGN.md section 7's recipe for fewer than eight GPUs): the properties on all 4096
    trials, every 64th against the port."""
    import copy
    import torch
    import bench
    from oracle import port
    fs = _fs()
    label, B, T, _, _ = bench.WORKLOADS[kind]
    B = batch or B
    wl = bench.make_workload(B, T, seed=0, kind=kind)
    ys = torch.from_numpy(wl['ys']).cuda()
    a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'])
    if kind == 'cd_sgp':
        f = fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], *a, ys)
        s = fs.cd_sgp_smoother(wl['drift'], wl['disp'](None), wl['sgps'], f[0], f[1], wl['dt'])
    else:
        f = fs.sgp_filter(wl['disc'], wl['sgps'], *a, ys)
        s = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'])
    mfs, Pfs, nll = f
    mss, Pss = s
    assert bool(torch.isfinite(mfs).all()) and bool(torch.isfinite(Pss).all()) and bool(torch.isfinite(nll).all())
    assert torch.equal(mss[:, -1], mfs[:, -1]) and torch.equal(Pss[:, -1], Pfs[:, -1])
    if kind != 'cd_sgp':      # exact for the discrete smoothers; the RK4-integrated backward ODE keeps it only to its own truncation error
        dvar = torch.diagonal(Pfs - Pss, dim1=-2, dim2=-1)
        assert float(dvar.min()) > -1e-9 * float(torch.diagonal(Pfs, dim1=-2, dim2=-1).abs().max())
    sel_all = np.arange(0, B, every)
    worst = dict.fromkeys(('mfs', 'Pfs', 'nll', 'mss', 'Pss'), 0.0)
    slab = 128 if kind != 'sgp' else 250                     # trials per port call: bounds the host memory (d = 8: 1.2 MB a trial-record)
    for lo in range(0, len(sel_all), slab):
        sel = sel_all[lo:lo + slab]
        if kind == 'cd_sgp':
            dg = copy.copy(wl['drift'])
            dg.gamma = wl['disp'].outer()
            w_f = port.filter(port.F_CD_SGP, dg, wl['sgps'], *a, wl['ys'][sel])
            w_s = port.smoother(port.S_CD_SGP, dg, wl['sgps'], wl['dt'], w_f[0], w_f[1])
        else:
            w_f = port.filter(port.F_SGP, wl['disc'], wl['sgps'], *a, wl['ys'][sel])
            w_s = port.smoother(port.S_SGP, wl['disc'], wl['sgps'], wl['dt'], w_f[0], w_f[1])
        idx = torch.from_numpy(sel).cuda()
        for g, w, n in zip((mfs[idx], Pfs[idx], nll[idx], mss[idx], Pss[idx]), w_f + w_s, worst):
            g = g.cpu().numpy()
            cs.assert_close(g, w, RTOL, f'{kind} full size: {n}')
            worst[n] = max(worst[n], cs.max_rel_err(g, w))
    print(kind, B, {n: f'{e:.2e}' for n, e in worst.items()})
    for n, err in worst.items():
        assert err <= 1e-7, (kind, n, err)


def test_second_device_while_first_is_current():
    """Engine on cuda:1 while cuda:0 is the current device: context, constants, stream and outputs follow the DATA's device,
    and the C-ABI leaves the thread's current device alone.  Needs two GPUs (skipped on the one-GPU box; the driver's
    multi-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs')
    fs = _fs()
    c = cs.chirp_case(T=300, seed=31)
    torch.cuda.set_device(0)
    ys1 = torch.from_numpy(np.tile(c.ys, (5, 1))).to('cuda:1')
    f = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, ys1)
    s = fs.eks(c.disc, f[0], f[1], c.dt)
    g = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys1)
    assert torch.cuda.current_device() == 0
    assert all(t.device.index == 1 for t in f + s + g)
    f0 = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, ys1.to('cuda:0'))
    for a, b in zip(f, f0):
        assert torch.equal(a.cpu(), b.cpu())
    from oracle import port
    want = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    cs.assert_close(f[0][2].cpu().numpy(), want[0], RTOL, 'cuda:1 mfs')


@pytest.mark.parametrize('kind', ['sgp', 'harmonic_ekf', 'cd_sgp'])
def test_default_launch_shape_at_a_mid_size_batch(kind):
    """B = 5000: beyond one wavefront per SIMD but below the round-3 crossovers (cgp_api.hip:choose_wave), so the DEFAULT launch
    takes the lane-cooperative kernels with several wavefronts per SIMD -- every 250th trial against the port."""
    import bench
    from oracle import port
    import copy
    fs = _fs()
    B, T = 5000, 96
    wl = bench.make_workload(B, T, kind=kind)
    sel = np.arange(0, B, 250)
    a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'])
    if kind == 'sgp':
        f = fs.sgp_filter(wl['disc'], wl['sgps'], *a, wl['ys'])
        s = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'])
        wf = port.filter(port.F_SGP, wl['disc'], wl['sgps'], *a, wl['ys'][sel])
        ws = port.smoother(port.S_SGP, wl['disc'], wl['sgps'], wl['dt'], wf[0], wf[1])
    elif kind == 'harmonic_ekf':
        f = fs.ekf(wl['disc'], *a, wl['ys'])
        s = fs.eks(wl['disc'], f[0], f[1], wl['dt'])
        wf = port.filter(port.F_EKF, wl['disc'], None, *a, wl['ys'][sel])
        ws = port.smoother(port.S_EKS, wl['disc'], None, wl['dt'], wf[0], wf[1])
    else:
        dg = copy.copy(wl['drift']); dg.gamma = wl['disp'].outer()
        f = fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], *a, wl['ys'])
        s = fs.cd_sgp_smoother(wl['drift'], wl['disp'](None), wl['sgps'], f[0], f[1], wl['dt'])
        wf = port.filter(port.F_CD_SGP, dg, wl['sgps'], *a, wl['ys'][sel])
        ws = port.smoother(port.S_CD_SGP, dg, wl['sgps'], wl['dt'], wf[0], wf[1])
    for g, w, n in zip(tuple(x[sel] for x in f) + tuple(x[sel] for x in s), wf + ws, ('mfs', 'Pfs', 'nll', 'mss', 'Pss')):
        cs.assert_close(g, w, 1e-7, f'{kind} B=5000 {n}')


@pytest.mark.parametrize('kw', [pytest.param(WAVE, id='one_trial_per_wave'), pytest.param(WAVE_X4, id='four_trials_per_wave'),
                                pytest.param(THREAD, id='lane_per_trial')])
def test_measurement_vectors_other_than_the_chirp_builders(kw):
    """The matrix-core EKF / KF kernels take a short form of the update when H = [0, 1, 0, 0] (every chirp builder's H: H
    picks entries of Pp, cgp_mfma4.hpp) and the general form otherwise, chosen per wavefront.  A batch with a measurement
    vector PER TRIAL -- e_1 in some, dense in others, so that the four-trials-per-wave kernel sees mixed wavefronts -- on the
    chirp model (ekf + eks) and on a linear d = 4 model (kf + rts), against the C port."""
    from oracle import port, np_filters as nf
    fs = _fs()
    B, T = 9, 300
    c = _batch_case(cs.chirp_case, B, T=T)
    rng = np.random.default_rng(77)
    H = np.tile(np.array([0., 1., 0., 0.]), (B, 1))
    H[[1, 2, 6]] = np.array([0.3, 1., -0.2, 0.1]) + 0.05 * rng.standard_normal((3, 4))
    H[8] = np.array([0., 1., 0., 1e-3])                          # nearly e_1: must take the general form too
    want = port.filter(port.F_EKF, c.disc, None, H, c.Xi, c.m0, c.P0, c.dt, c.ys)
    got = fs.ekf(c.disc, H, c.Xi, c.m0, c.P0, c.dt, c.ys, **kw)
    for g, w, nm in zip(got, want, ('mfs', 'Pfs', 'nll')):
        cs.assert_close(g, w, RTOL, f'ekf, per-trial H: {nm}')
    ws = port.smoother(port.S_EKS, c.disc, None, c.dt, want[0], want[1])
    gs = fs.eks(c.disc, want[0], want[1], c.dt, **kw)
    for g, w, nm in zip(gs, ws, ('mss', 'Pss')):
        cs.assert_close(g, w, RTOL, f'eks: {nm}')
    # linear model, d = 4: one H for the batch (kf takes a single model), e_1 and dense in turn
    F = 0.9 * np.eye(4) + 0.04 * rng.standard_normal((4, 4))
    Sigma = 0.1 * np.eye(4) + 0.01 * np.ones((4, 4))
    ys = rng.standard_normal((5, T))
    for Hl in (np.array([0., 1., 0., 0.]), np.array([0.4, -1., 0.2, 0.7])):
        gf = fs.kf(F, Sigma, Hl, 0.3, np.zeros(4), np.eye(4), ys, **kw)
        for b in range(ys.shape[0]):
            wf = nf.kf(F, Sigma, Hl, 0.3, np.zeros(4), np.eye(4), ys[b])
            for g, w in zip(gf, wf):
                npt.assert_allclose(g[b], w, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('nh', [4, 5])
def test_four_and_five_harmonics_the_bat_call_models(nh):
    """The reference's real applications run the harmonic chirp model with 4 and 5 harmonics (d = 10, 12) through the
    cubature filter and smoother (real_applications/bats/myotis_myotis_analysis.py:50-74, eptesicus_nilssonii_analysis.py:49-73;
    frequency state scaled by 1e4, Xi = 1e-4).  Those dimensions run on the generic kernels: parity with the C port for
    ekf / eks / sgp_filter / sgp_smoother, one wavefront and one lane per trial, a batch of three records."""
    c = _batch_case(cs.harmonic_case, 3, T=180, nh=nh)
    only = ('ekf', 'eks', 'sgp_filter', 'sgp_smoother')
    want = bk.run_pairs('port', c, only=only)
    for kw in (WAVE, THREAD):
        got = bk.run_pairs('hip', c, hip_kw=kw, only=only)
        bk.compare(got, want, RTOL, f'nh={nh}')


def _bat_like_record(T, nh, seed, fs=250000.):
    """A downward sweep 31 -> 22 kHz with nh harmonics sampled at 250 kHz, normalised like the reference's recordings
    (real_applications/bats/myotis_myotis_analysis.py:44-46); the recordings themselves are not part of the reference."""
    rng = np.random.default_rng(seed)
    t = np.arange(1, T + 1) / fs
    f = 2.2e4 * (1 + 0.4 * np.exp(-3 * t / t[-1]))
    phase = 2 * np.pi * np.cumsum(f) / fs
    ys = sum(0.5 ** k * np.sin((k + 1) * phase) for k in range(nh))
    return ys / np.max(np.abs(ys)) + 1e-2 * rng.standard_normal(T), f


def test_bat_call_parameters_track_the_sweep():
    """The bat-call application's own set-up -- 4 harmonics, frequency state scaled by 1e4, Xi = 1e-4, sigma = 10, ell = 0.2,
    dt = 1 / 250 kHz, cubature filter + smoother (myotis_myotis_analysis.py:50-74) -- on a synthetic sweep: the smoothed
    frequency follows the truth, and the engine agrees with the C port.  The recursion is badly conditioned while it locks on
    (P0 and sigma are large): the two CPU oracles differ by 6e-6 from each other there, hence the loose gate."""
    from chirpgp_amd import filters_smoothers as fs
    from chirpgp_amd.models import g
    from chirpgp_amd.quadratures import gaussian_expectation
    nh, T = 4, 600
    c = cs.harmonic_case(T=T, seed=35, nh=nh, params=(0.1, 1., 1., 0.2, 10., 2.), freq_scale=1e4, Xi=1e-4, dt=1. / 250000)
    c.ys, truth = _bat_like_record(T, nh, 5)
    want = bk.run_pairs('port', c, only=('sgp_filter', 'sgp_smoother'))
    for kw in (WAVE, THREAD):
        got = bk.run_pairs('hip', c, hip_kw=kw, only=('sgp_filter', 'sgp_smoother'))
        for k in ('sgp_filter', 'sgp_smoother'):
            for gv, wv in zip(got[k][:2], want[k][:2]):
                assert np.isfinite(gv).all()
                err = np.max(np.abs(gv - wv)) / np.max(np.abs(wv))
                print(kw, k, f'{err:.2e}')
                assert err <= 1e-4           # above the 6e-6 by which the two CPU oracles disagree on this ill-conditioned start
        mss, Pss = got['sgp_smoother']
        est = gaussian_expectation(ms=mss[:, -2], chol_Ps=np.sqrt(Pss[:, -2, -2]), func=g, force_shape=True)[:, 0] * 1e4
        assert np.max(np.abs(est[T // 3:] / truth[T // 3:] - 1.0)) < 0.05, 'smoothed frequency off the sweep'


def test_high_frequency_regime_of_the_matrix_core_ekf():
    """Chunks that start with the frequency state at 6.5 or above run on the short polynomials of the regime u2 >= 5
    (exp to degree 6, log1p(t) / t to degree 2, 1 / (1 + t) to degree 3 without a reciprocal: cgp_fastmath.hpp, SpecRegsHigh)
    and are repeated in the common regime when a step dips below 5.  Records that live inside that regime, one that hovers
    around its two thresholds, one that sweeps through it: within 1e-9 of the C port (the polynomials are good for 1e-13 rad
    of rotation angle), i.e. well inside what the common-regime kernel delivers."""
    import math
    from oracle import np_models as om_
    T, dt, Xi = 1800, 1e-3, 0.05
    ts = dt * np.arange(1, T + 1)
    recs = []
    for f_lo, f_hi, seed in [(8.5, 12.0, 1), (30.0, 60.0, 2), (4.9, 5.8, 3), (3.0, 9.0, 4)]:
        freq = f_lo + (f_hi - f_lo) * 0.5 * (1 - np.cos(2 * math.pi * ts / ts[-1] * 2.5))
        phase = np.cumsum(freq) * dt
        recs.append(np.sin(2 * math.pi * phase) + math.sqrt(Xi) * np.random.default_rng(seed).standard_normal(T))
    c = cs.chirp_case(T=8, params=(0.1, 0.5, 0.1, 0.3, 3., float(om_.g_inv(8.0))), Xi=Xi, dt=dt)
    c.ys = np.stack(recs)
    want = bk.run_pairs('port', c, only=('ekf',))
    got = bk.run_pairs('hip', c, hip_kw=WAVE, only=('ekf',))
    u2 = want['ekf'][0][:, :, 2]
    assert (u2[0, 300:] > 5.5).all() and (u2[1, 300:] > 20).all() and (u2[2] < 5.0).any() and (u2[2] > 5.5).any() and (u2[3] < 4.0).any()
    bk.compare(got, want, 1e-9, 'high-frequency regime')


def test_rotation_increments_at_the_admitted_bound():
    """The speculative step advances the Jacobian's rotation entries by the angle increment d with tan(d / 2), sin d to d^3, which it
    admits up to |d| < 1.5 x 2^-8 (cgp_mfma4.hpp: kIncrementBound); a chunk with a larger one is repeated with the checked step
    (full sincos) -- that chunk alone.  Records whose largest increments sit just below and just above the bound (a wide
    frequency prior, sigma = 2.5 and 3: the gain moves the frequency state by up to 1 Hz a step): both sides of the bound within
    1e-9 of the C port, the kernel's own counters showing that chunks below it were kept and chunks above it repeated."""
    from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
    from oracle import port
    bound = 1.5 * 2.0 ** -8
    c = cs.chirp_case(T=2000, seed=77)
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(5).standard_normal((8, c.ys.size))
    seen = {}
    for sigma in (2.5, 3.0):
        _, _, disc, m0, P0, H = pm.build_chirp_model(np.array([0.1, 0.1, 0.1, 1., sigma, 7.]))
        want = port.filter(port.F_EKF, disc, None, H, c.Xi, m0, P0, c.dt, ys)
        u2 = np.concatenate([np.full((8, 1), m0[2]), want[0][:, :-1, 2]], axis=1)
        d = np.abs(np.diff(c.dt * 2 * np.pi * np.log1p(np.exp(u2)), axis=1)) / bound          # increment of step k + 1, in units of the bound
        _engine.debug_set(_engine.DBG_COUNT_REGIMES, 1)
        _engine.debug_counters(reset=True)
        got = fs.ekf(disc, H, c.Xi, m0, P0, c.dt, ys, **WAVE)
        counters = _engine.debug_counters(reset=True)
        _engine.debug_set(_engine.DBG_COUNT_REGIMES, 0)
        for g_, w_, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
            err = cs.max_rel_err(g_, w_)
            assert err <= 1e-9, (sigma, n, err)
        seen[sigma] = (d.max(axis=1), counters)
        print(sigma, np.round(d.max(axis=1), 3), counters)
    near, c25 = seen[2.5]
    over, c30 = seen[3.0]
    assert (near > 0.85).sum() >= 4 and (near > 1.0).sum() <= 2 and c25['redone'] <= 3       # increments at 0.85 .. 1 of the bound: kept
    assert (over > 1.0).sum() >= 5 and 3 <= c30['redone'] <= 16 and c30['checked'] == 0      # beyond it: that chunk repeated, nothing sticky


def test_wide_step_for_records_outside_the_lean_regime():
    """Records whose frequency state lives below 1.5 or wanders through it (a chirp the filter never locks on: 20 Hz against an initial
    7; a start at 0.5 on a 1 - 2 Hz chirp; a negative start): the matrix-core EKF runs their chunks in the LOW / MID / ANY regimes of its
    speculative step or on the WIDE step (cgp_mfma4.hpp; softplus as lean polynomials on exp(u), as polynomials in u^2, by the branch-free
    softplus_pair_any) -- counted by the kernel -- within 1e-9 of the C port, and only a state beyond 700 (exp overflows in the reference's
    naive softplus: NaN from there on, in the same places) falls through to the checked step."""
    import bench
    from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
    from oracle import port
    T, B = 3000, 6
    for m0_v, off, label in ((7.0, 20.0, 'no lock'), (0.5, 1.0, 'low'), (-3.0, 2.0, 'negative start'), (705.0, 8.0, 'overflow')):
        _, _, disc, m0, P0, H = pm.build_chirp_model(np.array([0.1, 0.1, 0.1, 1., 1., m0_v]))
        ys = bench.chirp_batch(B, T, 5, Xi=0.1, offset=off)
        want = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 1e-3, ys)
        _engine.debug_set(_engine.DBG_COUNT_REGIMES, 1)
        _engine.debug_counters(reset=True)
        got = fs.ekf(disc, H, 0.1, m0, P0, 1e-3, ys, **WAVE)
        rg = _engine.debug_counters(reset=True)
        _engine.debug_set(_engine.DBG_COUNT_REGIMES, 0)
        print(label, rg, [f'{cs.max_rel_err(g, w):.1e}' for g, w in zip(got, want)])
        for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
            assert cs.max_rel_err(g, w) <= 1e-9, (label, n, cs.max_rel_err(g, w))
        chunks = B * ((T + 63) // 64)
        assert rg['high'] + rg['common'] + rg['low'] + rg['mid'] + rg['redone'] + rg['wide'] + rg['checked'] == chunks
        if label == 'overflow':
            assert rg['checked'] > 0
        else:
            assert rg['wide'] + rg['low'] + rg['mid'] > 0.2 * chunks and rg['checked'] == 0, (label, rg)
            if label == 'negative start':
                assert rg['low'] > 0.8 * chunks, rg                     # chunks that start at or below -1.75: the LOW regime of the lean step
            if label == 'low':
                assert rg['mid'] > 0.1 * chunks and rg['low'] > 0.1 * chunks, rg      # a start at 0.5: the MID regime, then wherever the state goes


@pytest.mark.parametrize('kind', ['sgp', 'cd_sgp', 'harmonic'])
def test_sigma_point_filters_on_records_outside_the_lean_regime(kind):
    """The one-wavefront sigma-point filters (C3: cgp_mfma4_sigma.hpp, C4: cgp_mfma4_cd.hpp, C5: cgp_coop8.hpp) evaluate a fan's
    softplus -> sin / cos chain in the lean form first, valid for a frequency state in [1.5, 700); a wavefront with a point outside
    it repeats the fan with the branch-free full-accuracy form (round 5, cgp_models.hpp: precompute_any) and only beyond |x| = 700
    with the checked one.  Records that live there -- a chirp the filter never locks on, a start at 0.5 on a 1 Hz chirp, a negative
    start, a start beyond 700 (sigma points on either side of the ANY form's bound, some past the overflow of the reference's naive
    softplus at 709.78) -- against the C port."""
    import copy
    import bench
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    T, B = 1500, 4
    nh = 3 if kind == 'harmonic' else 0
    for m0_v, off, label in ((7.0, 20.0, 'no lock'), (0.5, 1.0, 'low'), (-3.0, 2.0, 'negative start'), (707.0, 8.0, 'beyond 700')):
        params = np.array([0.1, 0.1, 0.1, 1., 1., m0_v])
        if kind == 'harmonic':
            drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, 3)
            sg = SigmaPoints.cubature(8)
        else:
            drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
            sg = SigmaPoints.gauss_hermite(4, 3)
        ys = bench.chirp_batch(B, T, 5, Xi=0.1, offset=off, num_harmonics=nh)
        if kind == 'cd_sgp':
            dg = copy.copy(drift)
            dg.gamma = disp.outer()
            want = port.filter(port.F_CD_SGP, dg, sg, H, 0.1, m0, P0, 1e-3, ys)
            got = fs.cd_sgp_filter(drift, disp(None), sg, H, 0.1, m0, P0, 1e-3, ys, **WAVE)
        else:
            want = port.filter(port.F_SGP, disc, sg, H, 0.1, m0, P0, 1e-3, ys)
            got = fs.sgp_filter(disc, sg, H, 0.1, m0, P0, 1e-3, ys, **WAVE)
        errs = [cs.max_rel_err(g, w) for g, w in zip(got, want)]
        print(kind, label, [f'{e:.1e}' for e in errs])
        # (max_rel_err also asserts equal NaN positions.)  Beyond 700 the points still inside [1.5, 700) take the LEAN softplus of the
        # checked fan, 7e-12 relative: at a state of 700 that is 3e-11 rad of rotation angle a step -- 2e-8 on these covariances, before
        # and after round 5; the north star's gate is 1e-5
        gate = 1e-6 if label == 'beyond 700' else 1e-9
        for e, n in zip(errs, ('mfs', 'Pfs', 'nll')):
            assert e <= gate, (kind, label, n, e)


@pytest.mark.parametrize('nh', [2, 3])
def test_tile_layout_sigma_filter_axial_and_rotated_cubature(nh):
    """The d = 6 / 8 tile-layout sigma-point filter takes ONE square root per lane and step when every sigma point sits on one axis
    (CGP_SIGMA_AXIAL, asserted by the host for cubature rules) and one per pivot otherwise.  Both paths against the C port: the
    cubature rule as is, and the same rule rotated by 30 degrees in the plane of the first two coordinates -- still unit weight,
    zero mean, identity second moment, groups differing in the last coordinate only (CGP_SIGMA_STANDARD), but no longer axial."""
    from chirpgp_amd import _engine
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    fs = _fs()
    c = cs.harmonic_case(T=400, seed=61, nh=nh)
    d = 2 * nh + 2
    sg = SigmaPoints.cubature(d)
    rot = np.eye(d)
    a = np.pi / 6
    rot[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    sg_rot = SigmaPoints(d, sg.n_points, sg.w, None, sg.xi @ rot.T)
    keep = []
    assert _engine._sigma_struct(sg, d, keep, d - 2).flags == (_engine.SIGMA_STANDARD | _engine.SIGMA_AXIAL)
    assert _engine._sigma_struct(sg_rot, d, keep, d - 2).flags == _engine.SIGMA_STANDARD
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(3).standard_normal((4, c.ys.size))
    for s_ in (sg, sg_rot):
        want = port.filter(port.F_SGP, c.disc, s_, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
        got = fs.sgp_filter(c.disc, s_, c.H, c.Xi, c.m0, c.P0, c.dt, ys, **WAVE)
        for g_, w_, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g_, w_, 1e-9, f'nh={nh} axial={s_ is sg}: {n}')


def test_gaussian_expectation_integrands():
    """The reference's own test of gaussian_expectation (test/test_utils.py:84-95: E[exp(V)] = exp(m + P / 2), default tolerance)
    on the device path, and the other enumerated integrands against their closed forms; an arbitrary callable is refused."""
    from chirpgp_amd.quadratures import gaussian_expectation, identity
    rng = np.random.default_rng(111)
    ms = rng.standard_normal((100, 1))
    Ps = rng.uniform(0.1, 1., size=(100, 1, 1))
    npt.assert_allclose(gaussian_expectation(ms, np.sqrt(Ps), func=np.exp, d=1, order=10), np.exp(ms + Ps[:, 0] / 2))
    npt.assert_allclose(gaussian_expectation(ms, np.sqrt(Ps), func='exp', force_shape=True), np.exp(ms + Ps[:, 0] / 2))
    npt.assert_allclose(gaussian_expectation(ms, np.sqrt(Ps), func=identity), ms, rtol=1e-13, atol=1e-15)
    npt.assert_allclose(gaussian_expectation(ms, np.sqrt(Ps), func=np.square), ms ** 2 + Ps[:, 0], rtol=1e-13)
    with pytest.raises(NotImplementedError):
        gaussian_expectation(ms, np.sqrt(Ps), func=lambda x: np.sin(x))
    with pytest.raises(NotImplementedError):
        gaussian_expectation(np.zeros((4, 2)), np.zeros((4, 2, 2)), d=2)
