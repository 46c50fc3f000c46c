"""Time-split filters with burn-in (cgp_filter_time_split): several wavefronts per trial for batches that leave most SIMDs idle.
Not the sequential recursion -- a filter cannot be cut in time exactly -- but as close to it as the reported junction mismatch says:
checked here against the sequential launch of the same kernels, with the mismatch the launch itself reports as the yardstick, and
(test_time_split_filters_against_the_oracle_at_the_bench_shards) against the CPU oracle itself at the shard sizes bench.py reports."""
import numpy as np
import pytest

from tests import cases as cs

pytestmark = pytest.mark.gpu


def _noisy(c, B, seed):
    return c.ys[None, :] + 0.05 * np.random.default_rng(seed).standard_normal((B, c.ys.size))


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize('T,segments,burn_in', [(6000, 4, 3000), (5003, 3, 2500), (4100, 8, 4000)])
def test_sgp_filter_time_split_is_as_close_as_its_junctions_say(T, segments, burn_in):
    """Ragged T (the last segment shorter, T not a multiple of 64), a burn-in longer than a segment (clipped at the record's start):
    every output within a small multiple of the reported junction mismatch of the sequential filter, cumulative NLL continuous."""
    from chirpgp_amd import filters_smoothers as fs, _engine
    c = cs.chirp_case(T=3000, seed=81)
    c.ys = np.tile(c.ys, 3)[:T] if T > 3000 else c.ys[:T]
    ys = _noisy(c, 5, T)
    seq = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    got = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys, time_split=(segments, burn_in))
    err = float(_engine.last_junction_error.max())
    print(T, segments, burn_in, 'junction mismatch', err, [f'{_rel(g, s):.1e}' for g, s in zip(got, seq)])
    seg_len = -(-(-(-T // segments)) // 64) * 64
    exact = -(-burn_in // 64) * 64 >= (-(-T // seg_len) - 1) * seg_len       # every burn-in reaches the record's start: the sequential filter
    assert err == 0.0 if exact else 0 < err < 1e-4
    for g, s, n in zip(got, seq, ('mfs', 'Pfs', 'nll')):
        assert np.isfinite(g).all()
        assert _rel(g, s) <= 5 * err + 1e-14, (n, _rel(g, s), err)
    assert np.array_equal(got[0][:, :seg_len], seq[0][:, :seg_len])               # the first segment IS the sequential filter
    last = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys, time_split=(segments, burn_in), nll_final_only=True, want=(False, False, True))[2]
    assert _rel(last, seq[2][:, -1]) <= 5 * err + 1e-14


def test_one_segment_is_the_sequential_filter_and_unsupported_methods_say_so():
    from chirpgp_amd import filters_smoothers as fs, _engine
    c = cs.chirp_case(T=700, seed=82)
    ys = _noisy(c, 3, 1)
    seq = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    one = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys, time_split=(1, 640))
    assert all(np.array_equal(a, b) for a, b in zip(one, seq)) and float(_engine.last_junction_error.max()) == 0.0
    short = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys[:, :50], time_split=(4, 640))      # one chunk: nothing to split
    assert np.array_equal(short[0], seq[0][:, :50])
    with pytest.raises(RuntimeError, match='time-split'):
        fs.cd_ekf(c.drift, c.disp, c.H, c.Xi, c.m0, c.P0, c.dt, ys, time_split=(2, 128))
    with pytest.raises(RuntimeError, match='time-split'):                              # the generic kernels do not know segments
        fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys, time_split=(2, 128), flags=0x10)


def test_a_nan_at_a_junction_is_reported_and_the_tolerance_falls_back():
    """A NaN measurement in the first segment: the sequential filter is NaN from there on, a segment that starts afresh is not --
    the junction says so (inf), and with split_tol the call returns the sequential result."""
    from chirpgp_amd import filters_smoothers as fs, _engine
    c = cs.chirp_case(T=3000, seed=84)
    ys = _noisy(c, 4, 2)
    ys[2, 300] = np.nan                                                                      # before the second segment's burn-in starts (at 512)
    seq = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    got = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys, time_split=(3, 512))
    err = _engine.last_junction_error.cpu().numpy()
    assert np.isinf(err[2]) and np.isfinite(err[[0, 1, 3]]).all()
    assert np.isfinite(got[0][2, 1024:]).all() and np.isnan(seq[0][2, 300:]).all()          # which is why the junction matters
    safe = fs.sgp_filter(c.disc, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys, time_split=(3, 512), split_tol=1e-1)
    for a, b in zip(safe, seq):
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


@pytest.mark.parametrize('kind,B,T,segments,burn_in', [('ekf', 125, 10000, 8, 4096), ('sgp', 125, 10000, 8, 3008), ('cd_sgp', 512, 50000, 2, 3008)])
def test_time_split_filters_against_the_oracle_at_the_bench_shards(kind, B, T, segments, burn_in):
    """The shards `bench.py` reports under other_configs.time_split_filters (BASELINE C2 / C3 on one of 8 GPUs: 125 x 10^4; C4: 512 x
    5 10^4), EVERY trial against the C port of the reference's sequential recursion (filters_smoothers.py:263, 489, 581) -- not against
    the engine's own sequential launch: within max(1e-5, 5 x the junction mismatch the launch reports) of the port, relative to each
    output's largest entry (1e-5 is the north star's tolerance; the mismatch is the launch's own accuracy figure)."""
    import copy
    import torch
    import bench
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    wl = bench.make_workload(B, T, kind=kind)
    ys = torch.from_numpy(wl['ys']).cuda()
    a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'])
    kw = dict(time_split=(segments, burn_in), return_junction_error=True)
    if kind == 'ekf':
        got = fs.ekf(wl['disc'], *a, ys, **kw)
        want = port.filter(port.F_EKF, wl['disc'], None, *a, wl['ys'])
    elif kind == 'sgp':
        got = fs.sgp_filter(wl['disc'], wl['sgps'], *a, ys, **kw)
        want = port.filter(port.F_SGP, wl['disc'], wl['sgps'], *a, wl['ys'])
    else:
        dg = copy.copy(wl['drift'])
        dg.gamma = wl['disp'].outer()
        got = fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], *a, ys, **kw)
        want = port.filter(port.F_CD_SGP, dg, wl['sgps'], *a, wl['ys'])
    err = got[3].cpu().numpy()
    assert err.shape == (B,) and np.isfinite(err).all() and err.max() > 0
    rel = []
    for g, w, n in zip(got[:3], want, ('mfs', 'Pfs', 'nll')):
        g = g.cpu().numpy()
        per_trial = np.abs(g - w).reshape(B, -1).max(axis=1) / np.abs(w).max()
        gate = np.maximum(1e-5, 5 * err)
        assert (per_trial <= gate).all(), (kind, n, float(per_trial.max()), float(err.max()))
        rel.append(float(per_trial.max()))
    print(f'{kind} {B} x {T} ({segments}, {burn_in}): junction mismatch {err.max():.1e}; worst difference from the port mfs / Pfs / nll ' +
          ' / '.join(f'{v:.1e}' for v in rel))


@pytest.mark.perf
def test_small_batch_time_split_is_faster():
    """BASELINE C3's shard on one of 8 GPUs (125 x 10 000): eight segments with 3008 steps of burn-in against the sequential launch.
    (A timing assertion: CGP_RUN_PERF=1 only.)"""
    import torch
    import bench
    from chirpgp_amd import filters_smoothers as fs, _engine
    wl = bench.make_workload(125, 10000, kind='sgp')
    ys = torch.from_numpy(wl['ys']).cuda()
    a = (wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)

    def timed(**kw):
        out = fs.sgp_filter(*a, **kw); torch.cuda.synchronize()
        ev = _engine.kernel_events = []
        for _ in range(3):
            out = fs.sgp_filter(*a, **kw)
        torch.cuda.synchronize(); _engine.kernel_events = None
        return out, min(x.elapsed_time(y) for n, x, y in ev if n == 'filter')
    seq, t_seq = timed()
    got, t_split = timed(time_split=(8, 3008))
    err = float(_engine.last_junction_error.max())
    worst = max(float((g - s).abs().max() / s.abs().max()) for g, s in zip(got, seq))
    print(f'sgp_filter 125 x 10000: sequential {t_seq:.3f} ms, 8 segments + 3008 burn-in {t_split:.3f} ms, junction mismatch {err:.1e}, worst output difference {worst:.1e}')
    assert err < 1e-6 and worst <= 5 * err and t_split < 0.6 * t_seq


@pytest.mark.parametrize('method', ['ekf', 'cd_sgp_filter', 'sgp_filter_d6', 'sgp_filter_d8', 'ekf_general_H', 'lascala_ekf', 'lascala_sgp'])
def test_the_other_kernels_that_know_segments(method):
    """ekf (the bench kernel, also with a measurement vector that is not e_1), cd_sgp_filter (RK4) at d = 4, and the tile-layout
    sigma-point filter at d = 6 / 8: split into four segments with 3008 steps of burn-in, against their own sequential launch."""
    from chirpgp_amd import filters_smoothers as fs, _engine
    T, B = 5500, 4
    if method.startswith('sgp_filter_d'):
        nh = 2 if method.endswith('6') else 3
        c = cs.harmonic_case(T=3000, seed=90 + nh, nh=nh)
    elif method.startswith('lascala'):
        c = cs.lascala_case(T=3000, seed=95)
    else:
        c = cs.chirp_case(T=3000, seed=90)
    ys = np.tile(c.ys, 2)[None, :T] + 0.05 * np.random.default_rng(7).standard_normal((B, T))
    H = c.H if method != 'ekf_general_H' else np.array([0.2, 1.0, 0.0, 0.1])
    if method in ('ekf', 'ekf_general_H', 'lascala_ekf'):
        run = lambda **kw: fs.ekf(c.disc, H, c.Xi, c.m0, c.P0, c.dt, ys, **kw)
    elif method == 'cd_sgp_filter':
        run = lambda **kw: fs.cd_sgp_filter(c.drift, c.disp(None), c.sgps, H, c.Xi, c.m0, c.P0, c.dt, ys, **kw)
    else:
        run = lambda **kw: fs.sgp_filter(c.disc, c.sgps, H, c.Xi, c.m0, c.P0, c.dt, ys, **kw)
    seq = run()
    got = run(time_split=(4, 3008))
    err = float(_engine.last_junction_error.max())
    rel = [_rel(g, s_) for g, s_ in zip(got, seq)]
    print(method, 'junction mismatch', f'{err:.1e}', [f'{v:.1e}' for v in rel])
    assert max(rel) <= 5 * err + 1e-14
    assert np.array_equal(got[0][:, :1408], seq[0][:, :1408])                          # segment 0 is the sequential filter
    if method == 'ekf_general_H':
        # with this measurement vector the EKF does NOT forget its start within 3008 steps (another start settles on another
        # track): the junctions say so -- a mismatch of percents -- and a caller's tolerance sends the call to the sequential filter
        assert err > 1e-3
        safe = run(time_split=(4, 3008), split_tol=1e-5)
        assert all(np.array_equal(a, b) for a, b in zip(safe, seq))
        return
    # (the La Scala model forgets more slowly: 1e-4 after 3008 steps where the chirp model is at 4e-8 -- the junctions say so)
    assert 0 < err < (1e-3 if method.startswith('lascala') else 1e-4)
    last = run(time_split=(4, 3008), nll_final_only=True, want=(False, False, True))[2]
    seq_last = run(nll_final_only=True, want=(False, False, True))[2]
    assert _rel(last, seq_last) <= 5 * err + 1e-12


# ------------------------------------------------------------------------------------------------ the smoother's counterpart (round 6)
def _cd_filtered(c, B, T, seed):
    """Filtering rows of B noisy copies of the case's record from the C port (the smoothers are compared on identical inputs)."""
    import copy
    from oracle import port
    dg = copy.copy(c.drift)
    dg.gamma = c.disp.outer()
    ys = c.ys[None, :T] + 0.05 * np.random.default_rng(seed).standard_normal((B, T))
    return dg, port.filter(port.F_CD_SGP, dg, c.sgps, c.H, c.Xi, c.m0, c.P0, c.dt, ys)


@pytest.mark.parametrize('T,segments,burn_in', [(3000, 3, 640), (3000, 2, 1280), (1000, 4, 0), (130, 2, 64), (65, 5, 64)])
def test_cd_sgp_smoother_time_split_against_its_sequential_launch(T, segments, burn_in):
    """cgp_smoother_time_split (cd_sgp_smoother on the d = 4 matrix-core kernel): every output within 5 x the junction mismatch the launch
    reports of the sequential launch's, the segment that is last in time bit-identical, ragged T (a record shorter than the segments asked
    for), no burn-in at all (the junctions then report what that costs)."""
    from chirpgp_amd import filters_smoothers as fs
    c = cs.chirp_case(T=T, seed=91)
    dg, f = _cd_filtered(c, 5, T, 1)
    b = c.disp(None)
    seq = fs.cd_sgp_smoother(c.drift, b, c.sgps, f[0], f[1], c.dt)
    got = fs.cd_sgp_smoother(c.drift, b, c.sgps, f[0], f[1], c.dt, time_split=(segments, burn_in), return_junction_error=True)
    err = got[2].cpu().numpy() if hasattr(got[2], 'cpu') else np.asarray(got[2])
    assert err.shape == (5,) and np.isfinite(err).all()
    chunks = (T - 1 + 63) // 64
    cps = -(-chunks // segments)
    own = 64 * cps                                                    # rows of the segment that is last in time: no junction in front of it
    for g, s_, n in zip(got[:2], seq, ('mss', 'Pss')):
        np.testing.assert_array_equal(g[:, T - 1 - min(own, T - 1):], s_[:, T - 1 - min(own, T - 1):])
        worst = np.abs(g - s_).reshape(5, -1).max(axis=1) / np.abs(s_).max()
        assert (worst <= np.maximum(1e-12, 5 * err)).all(), (n, worst, err)
    if burn_in == 0 and chunks > cps:
        assert err.max() > 1e-3                                       # a segment that starts from the FILTERING row is far from the smoothed one
    print(f'T {T} segments {segments} burn-in {burn_in}: junction mismatch {err.max():.1e}')


def test_cd_sgp_smoother_time_split_against_the_oracle_and_refusals():
    """C4's shape in small (T = 12 000, two segments, 2048 steps of burn-in) against the C port's sequential smoother on identical filtering
    rows: within max(1e-7, 5 x junction); split_tol sends a launch with a NaN junction back to the sequential smoother; other methods
    and kernel variants are refused by name."""
    from chirpgp_amd import filters_smoothers as fs, _engine
    from oracle import port
    T = 12000
    c = cs.chirp_case(T=T, seed=92)
    dg, f = _cd_filtered(c, 4, T, 2)
    b = c.disp(None)
    want = port.smoother(port.S_CD_SGP, dg, c.sgps, c.dt, f[0], f[1])
    got = fs.cd_sgp_smoother(c.drift, b, c.sgps, f[0], f[1], c.dt, time_split=(2, 2048), return_junction_error=True)
    err = np.asarray(got[2].cpu())
    assert 0 < err.max() < 1e-6
    for g, w, n in zip(got[:2], want, ('mss', 'Pss')):
        worst = np.abs(g - w).reshape(4, -1).max(axis=1) / np.abs(w).max()
        assert (worst <= np.maximum(1e-7, 5 * err)).all(), (n, worst, err)
    print(f'cd_sgp_smoother 4 x {T} (2, 2048): junction mismatch {err.max():.1e}')
    # a NaN in the filtering rows just behind a junction: inf, and split_tol returns the sequential result
    f2 = (f[0].copy(), f[1].copy())
    chunks = (T - 1 + 63) // 64
    row = T - 1 - 64 * (-(-chunks // 2))                              # the junction row of segment 1
    f2[1][1, row + 100, 2, 2] = np.nan                                # inside segment 1's burn-in stretch
    seq = fs.cd_sgp_smoother(c.drift, b, c.sgps, f2[0], f2[1], c.dt)
    fs.cd_sgp_smoother(c.drift, b, c.sgps, f2[0], f2[1], c.dt, time_split=(2, 2048))
    e2 = _engine.last_junction_error.cpu().numpy()
    assert np.isinf(e2[1]) and np.isfinite(e2[[0, 2, 3]]).all()
    safe = fs.cd_sgp_smoother(c.drift, b, c.sgps, f2[0], f2[1], c.dt, time_split=(2, 2048), split_tol=1e-3)
    for a_, b_ in zip(safe, seq):
        assert np.array_equal(np.isnan(a_), np.isnan(b_)) and np.array_equal(a_[~np.isnan(a_)], b_[~np.isnan(b_)])
    with pytest.raises(RuntimeError, match='cd_sgp_smoother'):
        fs.eks(c.disc, f[0], f[1], c.dt, time_split=(2, 640))
    with pytest.raises(RuntimeError, match='cd_sgp_smoother'):
        fs.cd_sgp_smoother(c.drift, b, c.sgps, f[0], f[1], c.dt, time_split=(2, 640), flags=0x80)      # the LDS-reduced kernel knows no segments


@pytest.mark.parametrize('T,segments,burn_in', [(3000, 3, 640), (1000, 4, 0), (65, 5, 64)])
def test_cd_eks_time_split_against_its_sequential_launch_and_the_oracle(T, segments, burn_in):
    """The same cut for cd_eks (cdeks4_mfma_kernel<SPLIT>): within 5 x the reported junction mismatch of the sequential launch, the segment
    last in time bit-identical, and (through the sequential launch's own parity) of the C port on identical filtering rows."""
    import copy
    from chirpgp_amd import filters_smoothers as fs
    from oracle import port
    c = cs.chirp_case(T=T, seed=93)
    dg = copy.copy(c.drift)
    dg.gamma = c.disp.outer()
    ys = c.ys[None, :T] + 0.05 * np.random.default_rng(3).standard_normal((3, T))
    f = port.filter(port.F_CD_EKF, dg, None, c.H, c.Xi, c.m0, c.P0, c.dt, ys)
    seq = fs.cd_eks(c.drift, c.disp, f[0], f[1], c.dt)
    got = fs.cd_eks(c.drift, c.disp, f[0], f[1], c.dt, time_split=(segments, burn_in), return_junction_error=True)
    err = np.asarray(got[2].cpu()) if hasattr(got[2], 'cpu') else np.asarray(got[2])
    assert err.shape == (3,) and np.isfinite(err).all()
    chunks = (T - 1 + 63) // 64
    cps = -(-chunks // segments)
    own = min(64 * cps, T - 1)
    want = port.smoother(port.S_CD_EKS, dg, None, c.dt, f[0], f[1])
    for g, s_, w, n in zip(got[:2], seq, want, ('mss', 'Pss')):
        np.testing.assert_array_equal(g[:, T - 1 - own:], s_[:, T - 1 - own:])
        worst = np.abs(g - s_).reshape(3, -1).max(axis=1) / np.abs(s_).max()
        assert (worst <= np.maximum(1e-12, 5 * err)).all(), (n, worst, err)
        if burn_in:
            worst = np.abs(g - w).reshape(3, -1).max(axis=1) / np.abs(w).max()
            assert (worst <= np.maximum(1e-8, 5 * err)).all(), (n, worst, err)
    if burn_in == 0 and chunks > cps:
        assert err.max() > 1e-3
    print(f'cd_eks T {T} segments {segments} burn-in {burn_in}: junction mismatch {err.max():.1e}')
    with pytest.raises(RuntimeError, match='cd_eks'):
        fs.cd_eks(c.drift, c.disp, f[0], f[1], c.dt, time_split=(2, 640), flags=0x80)          # the LDS-reduced kernel knows no segments
