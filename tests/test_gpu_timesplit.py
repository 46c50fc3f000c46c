"""Time-split discrete smoothers (CGP_TIME_SPLIT; default for small batches): every record cut into segments walked by
different wavefronts -- pass 1 composes each segment's affine map, pass 2 walks with the right carry (cgp_walk4.hpp,
cgp_coop8.hpp).  Exact by construction (the gain depends on the filtering results only: filters_smoothers.py:83-84, 212-215,
342-345, 524-527); checked against oracle/c/port.c on identical filtering inputs and against the one-wave-per-trial form."""
import numpy as np
import pytest

from tests import cases as cs

pytestmark = pytest.mark.gpu

WAVE, TIME_SPLIT, NO_TIME_SPLIT = 0x2, 0x800, 0x1000
SPLIT = dict(flags=WAVE | TIME_SPLIT)
WHOLE = dict(flags=WAVE | NO_TIME_SPLIT)


def _filter_inputs(c, B, seed, nan_trial=None):
    """Filtering results of B noisy copies of the case's record from the port (the smoothers are compared on identical inputs);
    ``nan_trial`` gets ONE poisoned filtering covariance in the middle of its record."""
    from oracle import port
    rng = np.random.default_rng(seed)
    ys = c.ys[None, :] + 0.05 * rng.standard_normal((B, c.ys.size))
    f = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    if nan_trial is not None:
        f[1][nan_trial, c.ys.size // 2, 1, 1] = np.nan
    return f


@pytest.mark.parametrize('T', [66, 130, 449, 450, 1000, 2049, 4161])
def test_eks_time_split_against_the_port_and_the_whole_record_walk(T):
    """Ragged T: one tile, segments of unequal length, a wrapped last tile; one trial with a NaN measurement in the middle."""
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    c = cs.chirp_case(T=T, seed=41)
    f = _filter_inputs(c, 5, T, nan_trial=3)
    want = port.smoother(port.S_EKS, c.disc, None, c.dt, f[0], f[1])
    got = fs.eks(c.disc, f[0], f[1], c.dt, **SPLIT)
    whole = fs.eks(c.disc, f[0], f[1], c.dt, **WHOLE)
    for g, w, o, n in zip(got, want, whole, ('mss', 'Pss')):
        cs.assert_close(g, w, 1e-9, f'eks split {n}')
        cs.assert_close(g, o, 1e-11, f'eks split vs whole {n}')
    assert np.isnan(got[0][3, :T // 2 + 1]).all() and np.isfinite(got[0][3, T // 2 + 1:]).all()  # NaN flows backwards in time only


@pytest.mark.parametrize('method', ['rts', 'sgp_smoother', 'sgp_smoother_literal', 'lascala_eks'])
def test_other_d4_smoothers_time_split(method):
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    T, B = 1500, 6
    if method == 'rts':
        c = cs.chirp_case(T=T, seed=43)
        import bench
        F, Sigma = bench.frozen_frequency_linear_model(np.array([0.1, 0.1, 0.1, 1., 1., 7.]), c.dt)
        from chirpgp_amd import models as pm
        lin = pm.linear_cond_m_cov(F, Sigma)
        ys = c.ys[None, :] + 0.05 * np.random.default_rng(1).standard_normal((B, T))
        f = port.filter(port.F_EKF, lin, None, c.H, c.Xi, c.m0, c.P0, 0., ys)
        want = port.smoother(port.S_EKS, lin, None, 0., f[0], f[1])
        got = fs.rts(F, Sigma, f[0], f[1], **SPLIT)
        whole = fs.rts(F, Sigma, f[0], f[1], **WHOLE)
    elif method == 'lascala_eks':
        c = cs.lascala_case(T=T, seed=44)
        f = _filter_inputs(c, B, 2)
        want = port.smoother(port.S_EKS, c.disc, None, c.dt, f[0], f[1])
        got = fs.eks(c.disc, f[0], f[1], c.dt, **SPLIT)
        whole = fs.eks(c.disc, f[0], f[1], c.dt, **WHOLE)
    else:
        c = cs.chirp_case(T=T, seed=45)
        f = _filter_inputs(c, B, 3, nan_trial=1)
        want = port.smoother(port.S_SGP, c.disc, c.sgps, c.dt, f[0], f[1])
        kw = dict(flags=SPLIT['flags'] | (0x40 if method.endswith('literal') else 0))
        got = fs.sgp_smoother(c.disc, c.sgps, f[0], f[1], c.dt, **kw)
        whole = fs.sgp_smoother(c.disc, c.sgps, f[0], f[1], c.dt, flags=kw['flags'] ^ TIME_SPLIT ^ NO_TIME_SPLIT)
    # exactness of the split: against the one-wave-per-trial form of the same kernel; accuracy: against the port (the elements
    # evaluate the models with the lean in-kernel functions, 1e-11 .. 1e-10 of the port, as in test_gpu_parity.py)
    for g, w, o, n in zip(got, want, whole, ('mss', 'Pss')):
        cs.assert_close(g, o, 1e-11, f'{method} split vs whole {n}')
        cs.assert_close(g, w, 1e-8 if 'sgp' not in method else 1e-7, f'{method} split {n}')


@pytest.mark.parametrize('nh,method', [(3, 'eks'), (3, 'sgp_smoother'), (2, 'sgp_smoother'), (3, 'rts')])
def test_d6_d8_smoothers_time_split(nh, method):
    """The tile-layout smoothers (5 <= d <= 8) in the time-split form: BASELINE C5's smoother at its 8-GPU shard size."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from oracle import port
    T, B = 1700, 4
    c = cs.harmonic_case(T=T, seed=46, nh=nh)
    rng = np.random.default_rng(4)
    ys = c.ys[None, :] + 0.05 * rng.standard_normal((B, T))
    if method == 'rts':
        d = 2 * nh + 2
        F = np.eye(d) * 0.99 + 0.01 * rng.standard_normal((d, d))
        Sigma = np.eye(d) * 0.01
        lin = pm.linear_cond_m_cov(F, Sigma)
        f = port.filter(port.F_EKF, lin, None, c.H, c.Xi, c.m0, c.P0, 0., ys)
        f[1][2, T // 3, 0, 0] = np.nan
        want = port.smoother(port.S_EKS, lin, None, 0., f[0], f[1])
        got = fs.rts(F, Sigma, f[0], f[1], **SPLIT)
        whole = fs.rts(F, Sigma, f[0], f[1], **WHOLE)
    elif method == 'eks':
        f = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
        f[1][2, T // 3, 0, 0] = np.nan
        want = port.smoother(port.S_EKS, c.disc, None, c.dt, f[0], f[1])
        got = fs.eks(c.disc, f[0], f[1], c.dt, **SPLIT)
        whole = fs.eks(c.disc, f[0], f[1], c.dt, **WHOLE)
    else:
        f = port.filter(port.F_SGP, c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
        f[1][2, T // 3, 0, 0] = np.nan
        want = port.smoother(port.S_SGP, c.disc, c.sgps, c.dt, f[0], f[1])
        got = fs.sgp_smoother(c.disc, c.sgps, f[0], f[1], c.dt, **SPLIT)
        whole = fs.sgp_smoother(c.disc, c.sgps, f[0], f[1], c.dt, **WHOLE)
    for g, w, o, n in zip(got, want, whole, ('mss', 'Pss')):
        cs.assert_close(g, w, 1e-8, f'd{2 * nh + 2} {method} split {n}')
        cs.assert_close(g, o, 1e-10, f'd{2 * nh + 2} {method} split vs whole {n}')


def test_random_record_lengths_split_equals_whole():
    """Forty random (T, B) pairs, T from 2 to 700 -- below one tile, on tile and quad boundaries, one step past them: the
    forced time-split launch (which falls back to one wave per trial when a record has too few tiles) against the whole-record
    walk, for `eks` (d = 4) and `rts` (d = 6)."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from oracle import port
    rng = np.random.default_rng(2024)
    c = cs.chirp_case(T=700, seed=51)
    d6 = 6
    F6 = np.eye(d6) * 0.97 + 0.02 * rng.standard_normal((d6, d6)); S6 = np.eye(d6) * 0.05
    lin6 = pm.linear_cond_m_cov(F6, S6)
    Ts = [2, 3, 4, 5, 63, 64, 65, 66, 127, 128, 129, 130, 191, 192, 193, 257, 321, 449, 450, 451] + list(rng.integers(6, 700, size=20))
    for T in Ts:
        T = int(T)
        B = int(rng.integers(1, 5))
        ys = c.ys[None, :T] + 0.05 * rng.standard_normal((B, T))
        f = port.filter(port.F_EKF, c.disc, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
        for a, b_ in zip(fs.eks(c.disc, f[0], f[1], c.dt, **SPLIT), fs.eks(c.disc, f[0], f[1], c.dt, **WHOLE)):
            cs.assert_close(a, b_, 1e-11, f'eks T={T} B={B}')
        f6 = port.filter(port.F_EKF, lin6, None, np.ones(d6), 0.1, np.zeros(d6), np.eye(d6), 0., rng.standard_normal((B, T)))
        want = port.smoother(port.S_EKS, lin6, None, 0., f6[0], f6[1])
        got = fs.rts(F6, S6, f6[0], f6[1], **SPLIT)
        whole = fs.rts(F6, S6, f6[0], f6[1], **WHOLE)
        for a, b_, o in zip(got, want, whole):       # (a random F: the affine records' C = Pf - G Pp G^T cancels to ~1e-11 of the scale)
            cs.assert_close(a, b_, 1e-7, f'rts d=6 T={T} B={B}')
            cs.assert_close(a, o, 1e-7, f'rts d=6 T={T} B={B} vs whole')


def test_time_split_with_one_parameter_vector_per_trial():
    """A parameter sweep through the smoothers: every trial its own model (param_stride != 0), d = 4 and d = 8."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    rng = np.random.default_rng(12)
    B, T = 6, 1100
    prm = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * np.exp(0.2 * rng.standard_normal((B, 6)))
    for nh in (1, 3):
        if nh == 1:
            _, _, disc, m0, P0, H = pm.build_chirp_model(prm)
            ys = np.stack([cs.chirp_measurements(T, 70 + b)[2] for b in range(B)])
        else:
            _, _, disc, m0, P0, H = pm.build_harmonic_chirp_model(prm, nh)
            ys = np.stack([cs.chirp_measurements(T, 70 + b, num_harmonics=nh)[2] for b in range(B)])
        f = fs.ekf(disc, H, 0.1, m0, P0, 1e-3, ys)
        for a, b_ in zip(fs.eks(disc, f[0], f[1], 1e-3, **SPLIT), fs.eks(disc, f[0], f[1], 1e-3, **WHOLE)):
            cs.assert_close(a, b_, 1e-11 if nh == 1 else 1e-10, f'per-trial params eks nh={nh}')   # (d = 8: the two record forms round differently)
        sg = SigmaPoints.gauss_hermite(4, 3) if nh == 1 else SigmaPoints.cubature(8)
        g = fs.sgp_filter(disc, sg, H, 0.1, m0, P0, 1e-3, ys)
        for a, b_ in zip(fs.sgp_smoother(disc, sg, g[0], g[1], 1e-3, **SPLIT), fs.sgp_smoother(disc, sg, g[0], g[1], 1e-3, **WHOLE)):
            cs.assert_close(a, b_, 1e-10, f'per-trial params sgp_smoother nh={nh}')


@pytest.mark.parametrize('d', [5, 7])
def test_odd_dimensions_time_split(d):
    """`rts` at d = 5 and 7: the tile layout's padded rows / columns of the composed maps (and of the workspace records) stay zero."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from oracle import port
    T, B = 1300, 3
    rng = np.random.default_rng(d)
    F = np.eye(d) * 0.98 + 0.02 * rng.standard_normal((d, d))
    Sigma = np.eye(d) * 0.02
    H = rng.standard_normal(d)
    lin = pm.linear_cond_m_cov(F, Sigma)
    ys = rng.standard_normal((B, T))
    f = port.filter(port.F_EKF, lin, None, H, 0.1, np.zeros(d), np.eye(d), 0., ys)
    want = port.smoother(port.S_EKS, lin, None, 0., f[0], f[1])
    got = fs.rts(F, Sigma, f[0], f[1], **SPLIT)
    whole = fs.rts(F, Sigma, f[0], f[1], **WHOLE)
    for g, w, o, n in zip(got, want, whole, ('mss', 'Pss')):
        cs.assert_close(g, w, 1e-9, f'rts d={d} split {n}')
        cs.assert_close(g, o, 1e-11, f'rts d={d} split vs whole {n}')


@pytest.mark.parametrize('kind', ['sgp', 'harmonic'])
def test_sigma_point_smoothers_at_the_eight_gpu_shard_size(kind):
    """BASELINE C3 / C5 sharded over 8 GPUs: 125 trials x 10 000 steps.  The default launch (time-split) against the
    one-wave-per-trial form of the same kernels on the engine's own filtering results."""
    import torch
    import bench
    from chirpgp_amd import filters_smoothers as fs
    wl = bench.make_workload(125, 10000, kind=kind)
    ys = torch.from_numpy(wl['ys']).cuda()
    f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
    auto = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'])
    whole = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'], flags=NO_TIME_SPLIT)
    for a, b in zip(auto, whole):
        cs.assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-10, f'{kind} auto vs whole')
    assert torch.isfinite(auto[0]).all()


def _small_batch_default_against_whole():
    """B = 125, T = 10 000 (BASELINE C3's shard on one of 8 GPUs): the default call against the whole-record walk; returns both times."""
    import torch
    import bench
    from chirpgp_amd import filters_smoothers as fs, _engine
    wl = bench.make_workload(125, 10000, kind='ekf')
    ys = torch.from_numpy(wl['ys']).cuda()
    f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)

    def timed(**kw):
        out = fs.eks(wl['disc'], f[0], f[1], wl['dt'], **kw)
        torch.cuda.synchronize()
        ev = _engine.kernel_events = []
        for _ in range(5):
            out = fs.eks(wl['disc'], f[0], f[1], wl['dt'], **kw)
        torch.cuda.synchronize()
        _engine.kernel_events = None
        return out, min(a.elapsed_time(b) for _, a, b in ev)
    auto, t_auto = timed()
    whole, t_whole = timed(flags=NO_TIME_SPLIT)
    print(f'eks B=125 T=10000: default {t_auto:.3f} ms, one wave per trial {t_whole:.3f} ms')
    for a, b in zip(auto, whole):
        cs.assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-11, 'auto vs whole')
    # the composed maps round differently from the step-by-step walk: bitwise equality would mean the default was NOT the split form
    assert not all(torch.equal(a, b) for a, b in zip(auto, whole))
    return t_auto, t_whole


def test_small_batch_takes_the_time_split_form_by_default():
    """The default call at a small batch is the time-split form: same results as the whole-record walk to rounding (1e-11), not bitwise."""
    _small_batch_default_against_whole()


@pytest.mark.perf
def test_small_batch_time_split_is_faster():
    """... and at least twice as fast (measured ~4x; the bound is loose on purpose).  A timing assertion: CGP_RUN_PERF=1 only."""
    t_auto, t_whole = _small_batch_default_against_whole()
    assert t_auto < 0.5 * t_whole, (t_auto, t_whole)


def test_linear_models_split_on_request_only():
    """`rts` on a caller-supplied F never takes the time-split form by default (its accuracy depends on the conditioning of products
    of the caller's gains: 1e-7 for the random F of test_random_record_lengths_split_equals_whole): at the small batch where `eks`
    splits, the default `rts` launch equals the one-wave-per-trial form bit for bit, and CGP_TIME_SPLIT still switches it on."""
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    from chirpgp_amd import models as pm
    d, B, T = 6, 16, 2000
    rng = np.random.default_rng(77)
    F = np.eye(d) * 0.97 + 0.02 * rng.standard_normal((d, d))
    S = np.eye(d) * 0.05
    lin = pm.linear_cond_m_cov(F, S)
    f = port.filter(port.F_EKF, lin, None, np.ones(d), 0.1, np.zeros(d), np.eye(d), 0., rng.standard_normal((B, T)))
    auto = fs.rts(F, S, f[0], f[1])
    whole = fs.rts(F, S, f[0], f[1], flags=NO_TIME_SPLIT)
    split = fs.rts(F, S, f[0], f[1], **SPLIT)
    for a, w, s in zip(auto, whole, split):
        assert np.array_equal(a, w)
        assert not np.array_equal(s, w)
        cs.assert_close(s, w, 1e-6, 'rts split on request')


def test_threads_sharing_a_context_and_a_stream():
    """the time-split forms keep their scratch in ONE buffer per (context, stream).  Host threads of one process share
    the context and, unless they set their own, torch's default stream; ctypes releases the GIL, so their compose / apply launches used to
    be free to interleave (compose_A, compose_B, apply_A -> A smoothed with B's maps).  A call's launches are now enqueued under a
    per-context lock: eight threads, different inputs and batch sizes (so the buffer also GROWS under them), each result equal to the
    single-threaded one bit for bit."""
    import threading
    import torch
    from chirpgp_amd import filters_smoothers as fs
    c = cs.chirp_case(T=1300, seed=47)
    jobs = []
    for i in range(8):
        f = _filter_inputs(c, 2 + 3 * i, 100 + i)
        jobs.append((torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()))
    ref = [tuple(t.clone() for t in fs.eks(c.disc, m, P, c.dt, **SPLIT)) for m, P in jobs]
    ys = [torch.from_numpy(c.ys[None, :] + 0.01 * np.random.default_rng(i).standard_normal((3 + i, c.ys.size))).cuda() for i in range(8)]
    fref = [fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, y, time_split=(4, 512)) for y in ys]
    torch.cuda.synchronize()
    out, fout, errs = [None] * 8, [None] * 8, []

    def work(i):
        try:
            for _ in range(6):
                out[i] = fs.eks(c.disc, jobs[i][0], jobs[i][1], c.dt, **SPLIT)
                fout[i] = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, ys[i], time_split=(4, 512))
        except Exception as e:                                       # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(8):
        for g, w in zip(out[i], ref[i]):
            assert torch.equal(g, w), i
        for g, w in zip(fout[i], fref[i]):
            assert torch.equal(g, w), i


def test_a_reserved_workspace_is_pinned():
    """a captured graph bakes the workspace pointer in, so a buffer sized by cgp_reserve_workspace is never freed or
    regrown by a launch -- a time-split filter that needs more fails loudly, a smoother takes its one-wavefront form -- until
    cgp_release_workspace (or a larger reserve)."""
    import torch
    from chirpgp_amd import filters_smoothers as fs, _engine
    c = cs.chirp_case(T=1300, seed=48)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        _engine.reserve_workspace(4096)                    # (rounded up to one MiB)
        big = torch.from_numpy(c.ys[None, :] + np.zeros((640, 1))).cuda()
        with pytest.raises(RuntimeError, match='workspace'):
            fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, big, time_split=(8, 256))      # 8 x 41 x 640 x 8 B = 1.6 MiB > the pinned MiB
        f = _filter_inputs(c, 640, 5)
        got = fs.eks(c.disc, f[0], f[1], c.dt, **SPLIT)                                  # falls back to the whole-record walk: same results
        _engine.release_workspace()
        ok = fs.ekf(c.disc, c.H, c.Xi, c.m0, c.P0, c.dt, big, time_split=(8, 256))
        whole = fs.eks(c.disc, f[0], f[1], c.dt, **WHOLE)
    s.synchronize()
    assert torch.isfinite(ok[0]).all()
    for g, w in zip(got, whole):
        cs.assert_close(g, w, 1e-11, 'pinned-workspace fallback')
