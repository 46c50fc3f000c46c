"""The large-batch kernels (cgp_lane4.hpp: one lane per trial at d = 4, LDS-DMA measurement prefetch, whole-line stores, trials grouped by
the phase of their rows against the 128-byte lines) against the C port of the reference's recursion (filters_smoothers.py:222-264, 446-490)
on the same inputs: the CRLB jobs' shape (tetralith/jobs/crlb_ekf.py:59-79, crlb_ghf.py:64-75) on a 4096-trial subset at 1e-9, ragged
batches, record lengths of every line phase and tail, every combination of outputs, shared / indexed records, per-trial models and
measurement vectors, NaN measurements -- and, at the full 262 144 x 500, size-independent properties.  The backward pass likewise: eks and
cd_eks one lane per trial (filters_smoothers.py:317-349, 400-443) on the port's own filtering results."""
import numpy as np
import pytest

from tests import cases as cs

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _close(g, w, what):
    """The gate of the CRLB-shaped comparisons: within 1e-9 of the port relative to the output's largest entry (the north star's
    '1e-5 relative' four decades tighter) AND element by element within 1e-8 (|w| + 1e-3 max |w|).  On these records -- a zero-mean GP for
    the frequency state, amplitudes of a few tenths -- the filter amplifies a rounding difference by 1e5 over 500 steps: 4096 trials
    differ from the port by up to 2e-10 in entries of size 1e-2 (the generic lane kernel of round 4 by 1.4e-10, tools/lane4_oldcheck.py)."""
    g = np.asarray(g.cpu().numpy() if hasattr(g, 'cpu') else g)
    cs.assert_close(g, w, 1e-8, what)
    assert cs.max_rel_err(g, w) <= RTOL, (what, cs.max_rel_err(g, w))
LANE = dict(flags=0x4)                       # CGP_THREAD_PER_TRIAL: the launch shape large batches take by themselves
OLD = dict(flags=0x14)                       # ... | CGP_GENERIC_KERNEL: the generic lane kernel (cgp_kernels.hpp)


def _crlb(T, B, seed=666):
    """The CRLB jobs' model and data: chirp SDE simulated on the device, dt = 0.01 (tetralith/jobs/crlb_ekf.py:39-66)."""
    from chirpgp_amd import tools
    from chirpgp_amd.models import model_chirp, disc_chirp_lcd
    _, _, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
    mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
    _, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, seed, batch=B, states=False)
    return mc, H, 0.1, m0, P0, 0.01, yss


def _filters(method):
    from chirpgp_amd import filters_smoothers as fs
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    sg = SigmaPoints.gauss_hermite(4, 3) if method == 'sgp' else None
    if method == 'ekf':
        return (lambda mc, *a, **kw: fs.ekf(mc, *a, **kw)), (lambda mc, *a, **kw: port.filter(port.F_EKF, mc, None, *a, **kw))
    return (lambda mc, *a, **kw: fs.sgp_filter(mc, sg, *a, **kw)), (lambda mc, *a, **kw: port.filter(port.F_SGP, mc, sg, *a, **kw))


@pytest.mark.parametrize('method', ['ekf', 'sgp'])
def test_crlb_shape_subset_against_the_port(method):
    """4096 trials x 500 steps of the CRLB jobs (T = 500: rows start 0, 32, 64 or 96 bytes into a line -- period 4, a 12-, 8- or 4-step
    head), all outputs and the means alone (the two kernel instantiations), every trial against the port at 1e-9."""
    hip, ref = _filters(method)
    mc, H, Xi, m0, P0, dt, yss = _crlb(500, 4096)
    want = ref(mc, H, Xi, m0, P0, dt, yss.cpu().numpy())
    got = hip(mc, H, Xi, m0, P0, dt, yss, **LANE)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        _close(g, w, f'{method} 4096 x 500 {n}')
    means = hip(mc, H, Xi, m0, P0, dt, yss, want=(True, False, False), **LANE)
    assert means[1] is None and means[2] is None
    _close(means[0], want[0], f'{method} means only')


@pytest.mark.parametrize('B,T', [(1037, 500), (130, 506), (64, 16), (70, 14), (129, 48), (65, 2), (200, 498), (3, 1000), (257, 510)])
def test_ragged_batches_heads_tails_and_every_output_combination(B, T):
    """Record lengths of every period (T mod 16 = 0: 1; 8: 2; 4 / 12: 4; 2 / 6 / 10 / 14: 8), shorter than a block, with and without a
    tail; batches that leave the last wavefronts of a group part empty or missing; each output alone and together; final NLL only."""
    hip, ref = _filters('ekf')
    mc, H, Xi, m0, P0, dt, yss = _crlb(T, B, seed=7 + B)
    want = ref(mc, H, Xi, m0, P0, dt, yss.cpu().numpy())
    for w3 in ((True, True, True), (True, False, False), (False, True, False), (False, False, True), (True, False, True), (False, True, True)):
        got = hip(mc, H, Xi, m0, P0, dt, yss, want=w3, **LANE)
        for g, w, n, on in zip(got, want, ('mfs', 'Pfs', 'nll'), w3):
            assert (g is not None) == on
            if on:
                _close(g, w, f'B={B} T={T} want={w3} {n}')
    fin = hip(mc, H, Xi, m0, P0, dt, yss, want=(False, False, True), nll_final_only=True, **LANE)[2]
    _close(fin, want[2][:, -1], 'final NLL')


def test_odd_record_lengths_and_unaligned_records_take_the_generic_lane_kernel():
    """The LDS-DMA moves 16-byte pieces: an odd T (rows of odd trials start on odd doubles) or a record base that is not 16-byte aligned
    goes to the generic lane kernel -- same results as the port either way."""
    import torch
    hip, ref = _filters('ekf')
    mc, H, Xi, m0, P0, dt, yss = _crlb(301, 200)
    want = ref(mc, H, Xi, m0, P0, dt, yss.cpu().numpy())
    got = hip(mc, H, Xi, m0, P0, dt, yss, **LANE)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        _close(g, w, f'T = 301 {n}')
    # even T, but the records start one double into the allocation
    buf = torch.empty(200 * 300 + 1, dtype=torch.float64, device='cuda')
    view = buf[1:].view(200, 300)
    view.copy_(yss[:, :300])
    want = ref(mc, H, Xi, m0, P0, dt, view.cpu().numpy())
    got = hip(mc, H, Xi, m0, P0, dt, view, **LANE)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        _close(g, w, f'unaligned base {n}')


@pytest.mark.parametrize('method', ['ekf', 'sgp'])
def test_shared_and_indexed_records_per_trial_parameters(method):
    """One record for every trial (ys_stride = 0), k parameter vectors per record (ys_repeat) and a record index -- the launches of an MLE
    sweep (include/chirpgp_hip.h: cgp_filter) -- with a model, Xi, m0 and P0 PER TRIAL: every lane's LDS-DMA reads its own record."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    R, k, T = 5, 30, 218
    B = R * k
    rng = np.random.default_rng(5)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * (1 + 0.2 * rng.random((B, 6)))
    drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
    Xi = 0.1 * (1 + rng.random(B))
    ys = np.stack([cs.chirp_case(T=T, seed=40 + r).ys for r in range(R)])
    sg = SigmaPoints.gauss_hermite(4, 3) if method == 'sgp' else None
    idx = np.array([4, 0, 3, 1, 2])
    rec = np.repeat(ys[idx], k, axis=0)                      # what trial b reads
    if method == 'ekf':
        got = fs.ekf(disc, H, Xi, m0, P0, 1e-3, ys, trials_per_record=k, record_index=idx, **LANE)
        want = port.filter(port.F_EKF, disc, None, H, Xi, m0, P0, 1e-3, rec)
    else:
        got = fs.sgp_filter(disc, sg, H, Xi, m0, P0, 1e-3, ys, trials_per_record=k, record_index=idx, **LANE)
        want = port.filter(port.F_SGP, disc, sg, H, Xi, m0, P0, 1e-3, rec)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        cs.assert_close(g, w, RTOL, f'{method} shared records {n}')
    one = fs.ekf(disc, H, Xi, m0, P0, 1e-3, ys[2], trials_per_record=B, **LANE)                  # ONE record, B parameter vectors
    cs.assert_close(one[0], port.filter(port.F_EKF, disc, None, H, Xi, m0, P0, 1e-3, np.repeat(ys[2:3], B, axis=0))[0], RTOL, 'one record')


def test_measurement_vectors_per_trial_and_the_lascala_model():
    """H = e_1 in every lane takes the short update (S = Pp_00 + Xi), any other lane in the wavefront the general one: a batch that mixes
    e_1 with dense measurement vectors from wavefront to wavefront and inside one; and the La Scala model (models.py:419-434), which
    shares the kernel (rho = 1, q = 0)."""
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    B, T = 200, 150
    c = cs.chirp_case(T=T, seed=61)
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(2).standard_normal((B, T))
    H = np.tile(c.H, (B, 1))
    H[70:130] = np.array([0.3, 1.0, -0.2, 0.05])             # wavefront 1 all dense (64 .. 127 partly), wavefront 2 mixed
    H[150] = np.array([0.0, 0.0, 1.0, 0.0])
    got = fs.ekf(c.disc, H, c.Xi, c.m0, c.P0, c.dt, ys, **LANE)
    want = port.filter(port.F_EKF, c.disc, None, H, c.Xi, c.m0, c.P0, c.dt, ys)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        cs.assert_close(g, w, RTOL, f'mixed H {n}')
    l = cs.lascala_case(T=T, seed=62)
    ys = l.ys[None, :] + 0.05 * np.random.default_rng(3).standard_normal((B, T))
    for hip, ref in ((lambda: fs.ekf(l.disc, l.H, l.Xi, l.m0, l.P0, l.dt, ys, **LANE), lambda: port.filter(port.F_EKF, l.disc, None, l.H, l.Xi, l.m0, l.P0, l.dt, ys)),
                     (lambda: fs.sgp_filter(l.disc, l.sgps, l.H, l.Xi, l.m0, l.P0, l.dt, ys, **LANE), lambda: port.filter(port.F_SGP, l.disc, l.sgps, l.H, l.Xi, l.m0, l.P0, l.dt, ys))):
        for g, w, n in zip(hip(), ref(), ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g, w, RTOL, f'lascala {n}')


def test_nan_measurements_poison_their_trial_only():
    """Numerical breakdown is not an error: a NaN measurement makes that trial NaN from there on, exactly where the
    reference's recursion has NaN, and touches no neighbour -- through the LDS-DMA, the staged NLL and the whole-line stores."""
    hip, ref = _filters('ekf')
    mc, H, Xi, m0, P0, dt, yss = _crlb(200, 300, seed=9)
    ys = yss.cpu().numpy().copy()
    ys[5, 0] = np.nan
    ys[64, 37] = np.nan
    ys[130, 199] = np.nan                                    # the very last step: the head / tail path of that trial's wavefront
    ys[299, 100] = np.nan
    want = ref(mc, H, Xi, m0, P0, dt, ys)
    got = hip(mc, H, Xi, m0, P0, dt, ys, **LANE)
    for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
        _close(g, w, f'NaN records {n}')           # NaN positions identical
    assert np.isnan(got[0][5]).all() and np.isfinite(got[0][4]).all() and np.isfinite(got[0][6]).all()
    assert np.isfinite(got[0][64, :37]).all() and np.isnan(got[0][64, 37:]).all()


def test_sigma_point_sets_other_than_the_standard_one():
    """cubature (2 d points, one group each after the host's grouping) and a set that is NOT standard (weights perturbed: the literal
    sums) through the staged lane kernel; a set too large for the LDS stage keeps the generic lane kernel."""
    from chirpgp_amd import filters_smoothers as fs
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    B, T = 100, 120
    c = cs.chirp_case(T=T, seed=71)
    ys = c.ys[None, :] + 0.05 * np.random.default_rng(4).standard_normal((B, T))
    cub = SigmaPoints.cubature(4)
    gh = SigmaPoints.gauss_hermite(4, 3)
    w = gh.w * (1 + 1e-3 * np.cos(np.arange(gh.n_points)))
    odd = gh._replace(w=w / w.sum())
    big = SigmaPoints.gauss_hermite(4, 7)                                 # 2401 points: 115 KB, beyond the stage
    for sg, name in ((cub, 'cubature'), (odd, 'non-standard weights'), (big, 'order 7')):
        got = fs.sgp_filter(c.disc, sg, c.H, c.Xi, c.m0, c.P0, c.dt, ys[:20 if sg is big else B], **LANE)
        want = port.filter(port.F_SGP, c.disc, sg, c.H, c.Xi, c.m0, c.P0, c.dt, ys[:20 if sg is big else B])
        for g, w_, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
            cs.assert_close(g, w_, RTOL, f'{name} {n}')


def test_full_size_properties():
    """262 144 x 500, the CRLB job's launch, where the port would take minutes: (i) the means-only launch equals the full launch's means
    bit for bit (two instantiations of one kernel); (ii) trials are independent -- the first and the last 1024 trials equal a launch of
    just those records, bit for bit; (iii) every output finite, covariances symmetric to the last bit (the store mirrors the packed lower
    triangle); (iv) every 512th trial against the port at 1e-9."""
    import torch
    hip, ref = _filters('ekf')
    mc, H, Xi, m0, P0, dt, yss = _crlb(500, 262144)
    full = hip(mc, H, Xi, m0, P0, dt, yss)
    means = hip(mc, H, Xi, m0, P0, dt, yss, want=(True, False, False))[0]
    assert torch.equal(means, full[0])
    del means
    for sl in (slice(0, 1024), slice(262144 - 1024, 262144)):
        part = hip(mc, H, Xi, m0, P0, dt, yss[sl].contiguous(), **LANE)
        assert all(torch.equal(p, f[sl]) for p, f in zip(part, full))
    assert all(bool(torch.isfinite(f).all()) for f in full)
    sel = np.arange(0, 262144, 512)
    idx = torch.from_numpy(sel).cuda()
    want = ref(mc, H, Xi, m0, P0, dt, yss[idx].cpu().numpy())
    for g, w, n in zip(full, want, ('mfs', 'Pfs', 'nll')):
        _close(g[idx], w, f'262144 x 500, every 512th trial: {n}')
    P = full[1]
    assert torch.equal(P, P.transpose(-1, -2))


@pytest.mark.parametrize('B,T', [(1037, 500), (130, 506), (64, 16), (70, 14), (129, 49), (65, 2), (200, 7), (3, 1001), (257, 511), (100, 5), (64, 3)])
def test_smoothers_one_lane_per_trial_against_the_port(B, T):
    """eks and cd_eks through lane4_smoother_kernel (rows by LDS-DMA a step ahead, mean lines a quad ahead, whole-line stores) on the port's
    filtering results: record lengths of every period (T mod 4 = 0: 1; 2: 2; odd: 4), records with no whole quad of rows, ragged batches;
    the last smoothing row is the last filtering row bit for bit (filters_smoothers.py:140-142)."""
    import copy
    from chirpgp_amd import filters_smoothers as fs, tools
    from chirpgp_amd.models import model_chirp, disc_chirp_lcd
    from oracle import port
    drift, disp, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
    mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
    dg = copy.copy(drift)
    dg.gamma = disp.outer()
    _, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 11 + B, batch=B, states=False)
    f = port.filter(port.F_EKF, mc, None, H, 0.1, m0, P0, 0.01, yss.cpu().numpy())
    for name, got, want in (('eks', fs.eks(mc, f[0], f[1], 0.01, **LANE), port.smoother(port.S_EKS, mc, None, 0.01, f[0], f[1])),
                            ('cd_eks', fs.cd_eks(drift, disp, f[0], f[1], 0.01, **LANE), port.smoother(port.S_CD_EKS, dg, None, 0.01, f[0], f[1]))):
        for g, w, n in zip(got, want, ('mss', 'Pss')):
            _close(g, w, f'{name} B={B} T={T} {n}')
        assert np.array_equal(got[0][:, -1], f[0][:, -1]) and np.array_equal(got[1][:, -1], f[1][:, -1])


def test_smoothers_per_trial_models_nan_rows_and_the_lascala_model():
    """Per-trial parameters; a NaN in a trial's filtering results poisons that trial's smoothing rows from there DOWN (the backward pass)
    and no other trial; the La Scala model shares the kernel."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from oracle import port
    B, T = 150, 203
    rng = np.random.default_rng(8)
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * (1 + 0.2 * rng.random((B, 6)))
    drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
    ys = np.stack([cs.chirp_case(T=T, seed=300 + (b % 5)).ys for b in range(B)]) + 0.05 * rng.standard_normal((B, T))
    ys[77, 120] = np.nan
    f = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 1e-3, ys)
    got = fs.eks(disc, f[0], f[1], 1e-3, **LANE)
    want = port.smoother(port.S_EKS, disc, None, 1e-3, f[0], f[1])
    for g, w, n in zip(got, want, ('mss', 'Pss')):
        cs.assert_close(g, w, RTOL, f'per-trial models {n}')
    assert np.isnan(got[0][77]).all() and np.isfinite(got[0][76]).all() and np.isfinite(got[0][78]).all()
    l = cs.lascala_case(T=T, seed=62)
    ysl = l.ys[None, :] + 0.05 * rng.standard_normal((70, T))
    fl = port.filter(port.F_EKF, l.disc, None, l.H, l.Xi, l.m0, l.P0, l.dt, ysl)
    for g, w, n in zip(fs.eks(l.disc, fl[0], fl[1], l.dt, **LANE), port.smoother(port.S_EKS, l.disc, None, l.dt, fl[0], fl[1]), ('mss', 'Pss')):
        cs.assert_close(g, w, RTOL, f'lascala eks {n}')


def test_smoother_full_size_properties():
    """262 144 x 500 through the DEFAULT eks launch (one lane per trial from 24 trials per SIMD on): every 512th trial against the port at
    1e-9; the first and last 1024 trials equal a launch of just those, bit for bit; smoothed covariances symmetric to the last bit."""
    import torch
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    hip, _ = _filters('ekf')
    mc, H, Xi, m0, P0, dt, yss = _crlb(500, 262144)
    f = hip(mc, H, Xi, m0, P0, dt, yss)
    s = fs.eks(mc, f[0], f[1], dt)
    for sl in (slice(0, 1024), slice(262144 - 1024, 262144)):
        part = fs.eks(mc, f[0][sl].contiguous(), f[1][sl].contiguous(), dt, **LANE)
        assert all(torch.equal(p, x[sl]) for p, x in zip(part, s))
    assert torch.equal(s[1], s[1].transpose(-1, -2)) and bool(torch.isfinite(s[0]).all())
    sel = np.arange(0, 262144, 512)
    idx = torch.from_numpy(sel).cuda()
    fm, fP = f[0][idx].cpu().numpy(), f[1][idx].cpu().numpy()
    want = port.smoother(port.S_EKS, mc, None, dt, fm, fP)
    for g, w, n in zip(s, want, ('mss', 'Pss')):
        _close(g[idx], w, f'262144 x 500, every 512th trial: {n}')
