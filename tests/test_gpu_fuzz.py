"""Randomised records through the regime state machine of the d = 4 EKF kernels.

The headline filter (cgp_mfma4.hpp: ekf4_mfma_kernel) runs 64-step chunks speculatively on lean polynomials chosen by the frequency
state's range -- HIGH (>= 5), common (>= 1.5), LOW (<= -1.5), MID (|x| < 2), the branch-free WIDE step, the checked step -- and repeats a
chunk that left its regime; five counters and thresholds tuned on 45 benchmark sets decide what is tried next.  Hand-built crossing
records test it in test_gpu_parity.py; here 240 seeded random record sets do: frequency-state tracks anywhere in -2 .. 40 (0.1 to 40 Hz)
that cross the boundaries at random places and rates, measurement noise 1e-3 .. 3, dt 1e-3 or 1e-2, T 65 .. 3000, per-trial model
parameters and noise levels, NaN and inf measurements -- through the one-trial-per-wavefront kernel, the four-trials-per-wavefront
kernel and the one-lane-per-trial kernel, each against the C port, NaN and inf positions identical, and the regime counters of the
one-trial kernel accounting for every chunk.

Tolerance, per output array and state component: 1e-9 -- or, for a record set on which the recursion itself is ill-conditioned, 300 x the
amount by which the PORT's own results move when every measurement is perturbed by 1e-15 of itself (measured per set, two runs of the
port); the matrix-core kernels may also use their speculative tiers' designed 1e-11 times that conditioning, up to 1e-7.  Of the 240 sets 228 sit below
1e-10 on all three kernels; the worst (seed 29: dt = 1e-2, the filter 25 units of frequency state away from the signal) moves the port by
1.8e-4 under that perturbation and the kernels by 2.3e-4 .. 5.9e-4 -- the full-accuracy lane kernel as much as the lean-polynomial ones:
it is the recursion that amplifies, not the tiers (ratios to the port's own response: <= 150 over the whole fuzz)."""
import math

import numpy as np
import pytest

from tests import cases as cs

pytestmark = pytest.mark.gpu

ONE, FOUR, LANE = 0x2 | 0x400, 0x2 | 0x200, 0x4
N_SETS = 240


def _softplus(v):
    return np.logaddexp(0.0, v)


def make_set(seed):
    """One random record set: (B, T, dt, Xi (B,), params (B, 6), ys (B, T), tracks (B, T))."""
    rng = np.random.default_rng(90000 + seed)
    B = int(rng.integers(3, 9))
    T = int(rng.integers(65, 3001))
    dt = 1e-3 if rng.random() < 0.6 else 1e-2
    Xi = 10 ** rng.uniform(-3, math.log10(3.0), size=B)
    if rng.random() < 0.5:
        Xi[:] = Xi[0]
    ts = dt * np.arange(1, T + 1)
    vmax = 40.0 if dt == 1e-3 else 25.0                       # stay well below Nyquist at dt = 1e-2
    params, ys, tracks = np.empty((B, 6)), np.empty((B, T)), np.empty((B, T))
    for b in range(B):
        kind = rng.integers(0, 4)
        if kind == 0:                                         # anywhere to anywhere
            v0, v1 = rng.uniform(-2, vmax, size=2)
        elif kind == 1:                                       # around the common / MID / LOW boundaries
            v0, v1 = rng.uniform(-2.5, 3.0, size=2)
        elif kind == 2:                                       # around the HIGH boundary
            v0, v1 = rng.uniform(3.5, 7.0, size=2)
        else:                                                 # flat
            v0 = rng.uniform(-2, vmax)
            v1 = v0 + rng.uniform(-0.3, 0.3)
        cyc = rng.uniform(0.3, 2.5)
        v = v0 + (v1 - v0) * 0.5 * (1 - np.cos(2 * math.pi * cyc * np.arange(T) / T))
        tracks[b] = v
        phase = np.cumsum(_softplus(v)) * dt
        amp = rng.uniform(0.5, 2.0)
        ys[b] = amp * np.sin(2 * math.pi * phase + rng.uniform(0, 2 * math.pi)) + math.sqrt(Xi[b]) * rng.standard_normal(T)
        params[b] = np.array([0.1, rng.uniform(0.05, 0.6), 0.1, rng.uniform(0.3, 1.2), rng.uniform(0.8, 3.0), v0]) * np.r_[rng.uniform(0.7, 1.3, 5), 1.0]
        if rng.random() < 0.1:
            params[b, 0] = 0.0                                # the lam = 0 branch of models.py:302-308
    if rng.random() < 0.15:
        ys[rng.integers(0, B), rng.integers(0, T)] = np.nan
    if rng.random() < 0.08:
        ys[rng.integers(0, B), rng.integers(0, T)] = np.inf
    return B, T, dt, Xi, params, ys, tracks


def distance(g, w, n):
    """Largest error of an output array, component by component, each against ITS scale: the frequency state runs up to 40, the chirp's
    amplitude is ~1 -- one denominator for the whole vector would hide the small components, and an element-wise floor of 1e-3 of the global
    scale would gate them at 1e-12 of theirs.  Non-finite entries (compared separately) count as 0."""
    bad = ~(np.isfinite(w) & np.isfinite(g))
    gf, wf = np.where(bad, 0.0, g), np.where(bad, 0.0, w)
    if n == 'mfs':
        return max(cs.max_rel_err(gf[..., c], wf[..., c]) for c in range(4))
    if n == 'Pfs':
        sd = np.sqrt(np.maximum(np.abs(wf[..., np.arange(4), np.arange(4)]).reshape(-1, 4).max(axis=0), 1e-300))
        return max(float(np.abs(gf[..., i, j] - wf[..., i, j]).max() / (sd[i] * sd[j])) for i in range(4) for j in range(4))
    return cs.max_rel_err(gf, wf)


def test_random_records_through_the_regime_state_machine():
    from chirpgp_amd import filters_smoothers as fs, models as pm, _engine
    from oracle import port
    totals = dict.fromkeys(_engine.REGIME_COUNTERS, 0)
    worst = {ONE: 0.0, FOUR: 0.0, LANE: 0.0}
    ill = 0
    failures = []
    below = arrays = 0
    for seed in range(N_SETS):
        B, T, dt, Xi, params, ys, tracks = make_set(seed)
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        want = port.filter(port.F_EKF, disc, None, H, Xi, m0, P0, dt, ys)
        # the recursion's own conditioning on this set: how far the port moves when every measurement moves by 1e-15 of itself
        moved = port.filter(port.F_EKF, disc, None, H, Xi, m0, P0, dt, ys * (1 + 1e-15 * np.random.default_rng(seed).choice([-1., 1.], size=ys.shape)))
        delta = {n: distance(np.asarray(b_), np.asarray(a), n) for a, b_, n in zip(want, moved, ('mfs', 'Pfs', 'nll'))}
        # kappa = delta / 1e-15 is the set's conditioning.  The lane kernel evaluates every function to a few ulp: 300 delta covers it (<= 150
        # seen).  The speculative tiers of the matrix-core kernels inject 1e-11 (LOW: 1.6e-9) by design (cgp_mfma4.hpp) -- kappa x 2e-11 is
        # theirs to use, but never more than 1e-7 on that account (seen: 7.5e-9 at kappa = 1.7e4, seed 23)
        tols = {n: max(1e-9, 300.0 * d) for n, d in delta.items()}
        tols_lean = {n: max(tols[n], min(1e-7, 2e4 * d)) for n, d in delta.items()}
        ill += max(tols.values()) > 1e-9
        for flags in (ONE, FOUR, LANE):
            if flags == ONE:
                _engine.debug_set(_engine.DBG_COUNT_REGIMES, 1)
                _engine.debug_counters(reset=True)
            got = fs.ekf(disc, H, Xi, m0, P0, dt, ys, flags=flags)
            if flags == ONE:
                rg = _engine.debug_counters(reset=True)
                _engine.debug_set(_engine.DBG_COUNT_REGIMES, 0)
                chunks = B * ((T + 63) // 64)
                kept = rg['high'] + rg['common'] + rg['low'] + rg['mid'] + rg['redone'] + rg['wide'] + rg['checked']
                if kept != chunks:
                    failures.append((seed, 'counters', kept, chunks, rg))
                for k in totals:
                    totals[k] += rg[k]
            for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
                g, w = np.asarray(g), np.asarray(w)
                # non-finite entries (an inf measurement gives inf, then NaN) must be the SAME non-finite values; the rest is compared
                bad = ~np.isfinite(w)
                if not (np.array_equal(bad, ~np.isfinite(g)) and np.array_equal(np.isnan(w), np.isnan(g)) and np.array_equal(np.sign(w[bad & ~np.isnan(w)]), np.sign(g[bad & ~np.isnan(w)]))):
                    failures.append((seed, flags, n, 'non-finite entries differ', dict(B=B, T=T, dt=dt)))
                    continue
                e, tol = distance(g, w, n), (tols if flags == LANE else tols_lean)[n]
                if not e <= tol:
                    failures.append((seed, flags, n, f'{e:.3e} > {tol:.1e}', dict(B=B, T=T, dt=dt, delta=delta[n])))
                worst[flags] = max(worst[flags], e)
                below += e < 1e-10
                arrays += 1
                if e > 1e-9:
                    print(f'  seed {seed} flags {flags:#x} {n}: {e:.2e}  (B {B} T {T} dt {dt}; the port under a 1e-15 perturbation: {delta[n]:.2e})')
    print(f'{N_SETS} sets ({ill} ill-conditioned: tolerance above 1e-9); {below} of {arrays} output arrays below 1e-10; worst relative error one-trial {worst[ONE]:.2e}, four-trials {worst[FOUR]:.2e}, lane {worst[LANE]:.2e}')
    print('chunks by regime over the fuzz:', totals)
    assert not failures, failures[:6]
    # the fuzz reached every tier of the state machine
    for k in ('high', 'common', 'low', 'mid', 'wide', 'redone'):
        assert totals[k] > 0, (k, totals)


# ------------------------------------------------------------------------------------------------ the sigma-point filters' tiers
def _dist_any(g, w, n, d):
    bad = ~(np.isfinite(w) & np.isfinite(g))
    gf, wf = np.where(bad, 0.0, g), np.where(bad, 0.0, w)
    if n == 'mfs':
        return max(cs.max_rel_err(gf[..., c], wf[..., c]) for c in range(d))
    if n == 'Pfs':
        sd = np.sqrt(np.maximum(np.abs(wf[..., np.arange(d), np.arange(d)]).reshape(-1, d).max(axis=0), 1e-300))
        return max(float(np.abs(gf[..., i, j] - wf[..., i, j]).max() / (sd[i] * sd[j])) for i in range(d) for j in range(d))
    return cs.max_rel_err(gf, wf)


@pytest.mark.parametrize('config', ['gh3_d4', 'cubature_d8'])
def test_random_records_through_the_sigma_point_filters(config):
    """The sigma-point filters speculate too (cgp_mfma4_sigma.hpp, cgp_coop8.hpp, cgp_lane4.hpp: a lean fan where every point's frequency state
    sits in the common regime, a branch-free full-accuracy fan behind it, the checked fan beyond 700): the same random record sets -- tracks
    that cross the bands anywhere, per-trial parameters and noise levels, NaN / inf measurements -- through sgp_filter with Gauss-Hermite order 3
    on the d = 4 chirp model (one wavefront per trial AND one lane per trial) and with the cubature rule on the three-harmonic model (d = 8, tile
    layout), against the C port under the conditioning-based gate of the EKF fuzz."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    d = 4 if config == 'gh3_d4' else 8
    sgps = SigmaPoints.gauss_hermite(4, 3) if config == 'gh3_d4' else SigmaPoints.cubature(8)
    shapes = ((0x2, 'wave'), (0x4, 'lane')) if config == 'gh3_d4' else ((0x2, 'wave'),)
    failures, worst, ill, below, arrays = [], dict.fromkeys([s for _, s in shapes], 0.0), 0, 0, 0
    n_sets = 60
    for seed in range(n_sets):
        B, T, dt, Xi, params, ys, _ = make_set(1000 + seed)
        T = min(T, 1200)
        ys = np.ascontiguousarray(ys[:, :T])
        if config == 'gh3_d4':
            _, _, disc, m0, P0, H = pm.build_chirp_model(params)
        else:
            _, _, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, num_harmonics=3)
        want = port.filter(port.F_SGP, disc, sgps, H, Xi, m0, P0, dt, ys)
        moved = port.filter(port.F_SGP, disc, sgps, H, Xi, m0, P0, dt, ys * (1 + 1e-15 * np.random.default_rng(seed).choice([-1., 1.], size=ys.shape)))
        delta = {n: _dist_any(np.asarray(b_), np.asarray(a), n, d) for a, b_, n in zip(want, moved, ('mfs', 'Pfs', 'nll'))}
        tols = {n: max(1e-9, 300.0 * v) for n, v in delta.items()}
        tols_lean = {n: max(tols[n], min(1e-7, 2e4 * v)) for n, v in delta.items()}
        ill += max(tols.values()) > 1e-9
        for flags, shape in shapes:
            got = fs.sgp_filter(disc, sgps, H, Xi, m0, P0, dt, ys, flags=flags)
            for g, w, n in zip(got, want, ('mfs', 'Pfs', 'nll')):
                g, w = np.asarray(g), np.asarray(w)
                bad = ~np.isfinite(w)
                if not (np.array_equal(bad, ~np.isfinite(g)) and np.array_equal(np.isnan(w), np.isnan(g))):
                    failures.append((seed, shape, n, 'non-finite entries differ', dict(B=B, T=T, dt=dt)))
                    continue
                e, tol = _dist_any(g, w, n, d), tols_lean[n]
                if not e <= tol:
                    failures.append((seed, shape, n, f'{e:.3e} > {tol:.1e}', dict(B=B, T=T, dt=dt, delta=delta[n])))
                worst[shape] = max(worst[shape], e)
                below += e < 1e-10
                arrays += 1
                if e > 1e-9:
                    print(f'  seed {seed} {shape} {n}: {e:.2e}  (B {B} T {T} dt {dt}; the port under a 1e-15 perturbation: {delta[n]:.2e})')
    print(f'{config}: {n_sets} sets ({ill} ill-conditioned); {below} of {arrays} output arrays below 1e-10; worst ' + ', '.join(f'{k} {v:.2e}' for k, v in worst.items()))
    assert not failures, failures[:6]


def test_random_records_through_the_smoothers_and_the_continuous_discrete_filters():
    """The rest of the d = 4 chirp kernels on the random record sets: eks and cd_eks in both launch shapes (the walk / the matrix-core kernel, and one
    lane per trial), sgp_smoother and cd_sgp_smoother (Gauss-Hermite order 3) on the port's filtering rows, cd_ekf and cd_sgp_filter on the
    records -- each against the C port under the conditioning-based gate (the smoothers are run on IDENTICAL filtering rows, so their gate is the
    plain 1e-9 unless the port's smoother itself moves under a 1e-15 perturbation of those rows)."""
    import copy
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    gh3 = SigmaPoints.gauss_hermite(4, 3)
    failures, worst, arrays, below, broken, overflowed = [], {}, 0, 0, 0, 0

    def broken_from(want):
        """Per trial, the first step at which the port's filter has broken down -- a negative variance or a NaN likelihood (RK4 at dt = 1e-2 on a 20 Hz
        chirp loses the covariance's definiteness; the reference returns NaN from there on).  What any float64 implementation computes behind that
        point is the rounding noise of differences of 1e80-sized numbers: the port's own sample of it is reproduced only by a kernel with the port's
        operation order (the DPP kernels do, to 1e-10; the matrix-core ones do not), and nothing is compared there."""
        Pd = np.asarray(want[1])[..., np.arange(4), np.arange(4)]
        ok = np.isfinite(np.asarray(want[2])) & (Pd > 0).all(axis=-1) & np.isfinite(np.asarray(want[0])).all(axis=-1)
        return np.where(ok.all(axis=1), ok.shape[1], np.argmin(ok, axis=1))

    def check(tag, seed, got, want, moved, info, until=None, valid=None):
        nonlocal arrays, below
        for i, (g, w, mv) in enumerate(zip(got, want, moved)):
            g, w, mv = np.array(g), np.array(w), np.array(mv)
            if until is not None:
                for b, t in enumerate(until):
                    g[b, t:], w[b, t:], mv[b, t:] = 0.0, 0.0, 0.0
            if valid is not None:
                g[~valid], w[~valid], mv[~valid] = 0.0, 0.0, 0.0
            n = ('mfs', 'Pfs', 'nll')[i]
            bad = ~np.isfinite(w)
            if not (np.array_equal(bad, ~np.isfinite(g)) and np.array_equal(np.isnan(w), np.isnan(g))):
                failures.append((seed, tag, n, 'non-finite entries differ', info))
                continue
            delta = _dist_any(mv, w, n, 4)
            tol = max(1e-9, 300.0 * delta, min(1e-7, 2e4 * delta))
            e = _dist_any(g, w, n, 4)
            if not e <= tol:
                failures.append((seed, tag, n, f'{e:.3e} > {tol:.1e}', dict(info, delta=delta)))
            worst[tag] = max(worst.get(tag, 0.0), e)
            arrays += 1
            below += e < 1e-10

    for seed in range(40):
        B, T, dt, Xi, params, ys, _ = make_set(2000 + seed)
        T = min(T, 800)
        ys = np.ascontiguousarray(ys[:, :T])
        info = dict(B=B, T=T, dt=dt)
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        dg = copy.copy(drift)
        dg.gamma = disp.outer()
        sign = np.random.default_rng(seed).choice([-1., 1.], size=ys.shape)
        ys_moved = ys * (1 + 1e-15 * sign)
        # ---- continuous-discrete filters on the records
        for tag, method, sg, run in (('cd_ekf', port.F_CD_EKF, None, lambda: fs.cd_ekf(drift, disp, H, Xi, m0, P0, dt, ys)),
                                     ('cd_sgp_filter', port.F_CD_SGP, gh3, lambda: fs.cd_sgp_filter(drift, disp(None), gh3, H, Xi, m0, P0, dt, ys))):
            want = port.filter(method, dg, sg, H, Xi, m0, P0, dt, ys)
            moved = port.filter(method, dg, sg, H, Xi, m0, P0, dt, ys_moved)
            until = broken_from(want)
            broken += int((until < T).sum())
            check(tag, seed, run(), want, moved, info, until)
        # ---- smoothers on the port's filtering rows (finite records only: a NaN row makes every earlier smoothed row NaN, nothing to compare)
        fd = port.filter(port.F_EKF, disc, None, H, Xi, m0, P0, dt, ys)
        fc = port.filter(port.F_CD_EKF, dg, None, H, Xi, m0, P0, dt, ys)
        # (filtering rows a smoother can be asked to smooth: finite and positive definite throughout, from either filter)
        finite = np.isfinite(ys).all(axis=1) & (broken_from(fd) == T) & (broken_from(fc) == T)
        if not finite.any():
            continue
        sel = np.flatnonzero(finite)
        prm = params[sel]
        drift_s, disp_s, disc_s, _, _, _ = pm.build_chirp_model(prm)
        dg_s = copy.copy(drift_s)
        dg_s.gamma = disp_s.outer()
        for tag, method, model, sg, f, runs in (
                ('eks', port.S_EKS, disc_s, None, fd, {'wave': lambda m, P: fs.eks(disc_s, m, P, dt, flags=0x2), 'lane': lambda m, P: fs.eks(disc_s, m, P, dt, flags=0x4)}),
                ('sgp_smoother', port.S_SGP, disc_s, gh3, fd, {'wave': lambda m, P: fs.sgp_smoother(disc_s, gh3, m, P, dt, flags=0x2)}),
                ('cd_eks', port.S_CD_EKS, dg_s, None, fc, {'wave': lambda m, P: fs.cd_eks(drift_s, disp_s, m, P, dt, flags=0x2), 'lane': lambda m, P: fs.cd_eks(drift_s, disp_s, m, P, dt, flags=0x4)}),
                ('cd_sgp_smoother', port.S_CD_SGP, dg_s, gh3, fc, {'wave': lambda m, P: fs.cd_sgp_smoother(drift_s, disp_s(None), gh3, m, P, dt, flags=0x2)})):
            m, P = np.ascontiguousarray(f[0][sel]), np.ascontiguousarray(f[1][sel])
            if not (np.isfinite(m).all() and np.isfinite(P).all()):
                continue
            want = port.smoother(method, model, sg, dt, m, P)
            moved = port.smoother(method, model, sg, dt, m * (1 + 1e-15), P)
            # the backward RK4 can overflow as well (dt = 1e-2): rows are compared from the end of the record down to the port's first non-finite
            # one; behind it the implementations differ in where inf turns into NaN, which is nobody's result
            ok = np.isfinite(want[0]).all(axis=-1) & np.isfinite(want[1]).all(axis=(-1, -2))
            valid = np.flip(np.logical_and.accumulate(np.flip(ok, axis=1), axis=1), axis=1)
            overflowed += int((~valid.all(axis=1)).sum())
            for shape, run in runs.items():
                check(f'{tag} {shape}', seed, run(m, P), want, moved, info, valid=valid)
    print(f'{broken} (trial, filter) pairs broke down and {overflowed} (trial, smoother) pairs overflowed (compared up to there); {below} of {arrays} output arrays below 1e-10; worst by kernel: ' + ', '.join(f'{k} {v:.1e}' for k, v in sorted(worst.items())))
    assert not failures, failures[:6]


def _gate(g, w, mv, n, d):
    """(error, tolerance) of one output array under the fuzz's conditioning-based gate; None if the non-finite patterns differ."""
    g, w, mv = np.asarray(g), np.asarray(w), np.asarray(mv)
    bad = ~np.isfinite(w)
    if not (np.array_equal(bad, ~np.isfinite(g)) and np.array_equal(np.isnan(w), np.isnan(g))):
        return None
    delta = _dist_any(mv, w, n, d)
    return _dist_any(g, w, n, d), max(1e-9, 300.0 * delta, min(1e-7, 2e4 * delta))


def test_random_models_through_the_other_kernels():
    """What the chirp fuzz does not reach: the linear kernels at every dimension 1 .. 8 (kf + rts on random stable F, Sigma: generic, d = 4 matrix-core /
    walk and d >= 5 tile-layout kernels, whole-record and time-split smoothers), the La Scala model (ekf, eks, sgp_filter: the chirp kernels on
    its parameter layout), the harmonic models at d = 6 and 8 (ekf + eks in the tile layout and per lane), and ekf_for_kpt at d = 3, 4, 5 -- random
    parameters, noise levels, record lengths and NaN measurements, against the C port under the conditioning-based gate."""
    from chirpgp_amd import filters_smoothers as fs, models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    from oracle import port
    failures, worst, arrays, below = [], {}, 0, 0

    def compare(tag, seed, got, want, moved, d, names=('mfs', 'Pfs', 'nll')):
        nonlocal arrays, below
        for g, w, mv, n in zip(got, want, moved, names):
            r = _gate(g, w, mv, n, d)
            if r is None:
                failures.append((seed, tag, n, 'non-finite entries differ'))
                continue
            e, tol = r
            if not e <= tol:
                failures.append((seed, tag, n, f'{e:.3e} > {tol:.1e}'))
            worst[tag] = max(worst.get(tag, 0.0), e)
            arrays += 1
            below += e < 1e-10

    for seed in range(48):
        rng = np.random.default_rng(70000 + seed)
        B, T = int(rng.integers(2, 7)), int(rng.integers(30, 700))
        sign = rng.choice([-1., 1.], size=(B, T))
        # ---- linear: kf + rts, d = 1 .. 8
        d = 1 + seed % 8
        A = rng.standard_normal((d, d))
        F = 0.97 * A / max(np.abs(np.linalg.eigvals(A)).max(), 1e-3) if d > 1 else np.array([[0.9]])
        L = 0.3 * rng.standard_normal((d, d))
        Sigma = L @ L.T + 0.01 * np.eye(d)
        H = rng.standard_normal(d)
        Xi = 10 ** rng.uniform(-2, 0)
        m0, P0 = rng.standard_normal(d), np.eye(d) * rng.uniform(0.2, 2.0)
        ys = rng.standard_normal((B, T))
        if rng.random() < 0.3:
            ys[rng.integers(0, B), rng.integers(0, T)] = np.nan
        lin = pm.linear_cond_m_cov(F, Sigma)
        want = port.filter(port.F_EKF, lin, None, H, Xi, m0, P0, 0.0, ys)
        moved = port.filter(port.F_EKF, lin, None, H, Xi, m0, P0, 0.0, ys * (1 + 1e-15 * sign))
        for name, fl in (('wave', 0x2), ('lane', 0x4), ('generic', 0x12)):
            compare(f'kf d{d} {name}', seed, fs.kf(F, Sigma, H, Xi, m0, P0, ys, flags=fl), want, moved, d)
        ok = np.isfinite(ys).all(axis=1)
        if ok.any():
            m, P = np.ascontiguousarray(want[0][ok]), np.ascontiguousarray(want[1][ok])
            ws = port.smoother(port.S_EKS, lin, None, 0.0, m, P)
            ms = port.smoother(port.S_EKS, lin, None, 0.0, m * (1 + 1e-15), P)
            for name, fl in (('wave', 0x2), ('lane', 0x4), ('generic', 0x12), ('time-split', 0x2 | 0x800)):
                compare(f'rts d{d} {name}', seed, fs.rts(F, Sigma, m, P, flags=fl), ws, ms, d, ('mfs', 'Pfs'))
        # ---- La Scala (d = 4), harmonic (d = 6, 8), KPT (d = 3, 4, 5): a toy chirp per trial
        dt = 1e-3
        ts = dt * np.arange(1, T + 1)
        f0 = rng.uniform(2.0, 12.0, size=(B, 1))
        ysc = np.sin(2 * np.pi * (f0 * ts + 0.5 * rng.uniform(-3, 3, size=(B, 1)) * ts ** 2)) + np.sqrt(0.1) * rng.standard_normal((B, T))
        kind = seed % 3
        if kind == 0:
            prm = np.array([0.1, 1.0, 1.0, 7.0]) * rng.uniform(0.6, 1.5, size=(B, 4))
            _, _, disc, m0c, P0c, Hc = pm.build_lascala_model(prm)
            gh3 = SigmaPoints.gauss_hermite(4, 3)
            for tag, method, sg, run in (('lascala ekf', port.F_EKF, None, lambda fl: fs.ekf(disc, Hc, 0.1, m0c, P0c, dt, ysc, flags=fl)),
                                         ('lascala sgp_filter', port.F_SGP, gh3, lambda fl: fs.sgp_filter(disc, gh3, Hc, 0.1, m0c, P0c, dt, ysc, flags=fl))):
                want = port.filter(method, disc, sg, Hc, 0.1, m0c, P0c, dt, ysc)
                moved = port.filter(method, disc, sg, Hc, 0.1, m0c, P0c, dt, ysc * (1 + 1e-15 * sign))
                for name, fl in (('wave', 0x2), ('lane', 0x4)):
                    compare(f'{tag} {name}', seed, run(fl), want, moved, 4)
                if sg is None:
                    ws = port.smoother(port.S_EKS, disc, None, dt, want[0], want[1])
                    ms = port.smoother(port.S_EKS, disc, None, dt, want[0] * (1 + 1e-15), want[1])
                    for name, fl in (('wave', 0x2), ('lane', 0x4), ('time-split', 0x2 | 0x800)):
                        compare(f'lascala eks {name}', seed, fs.eks(disc, want[0], want[1], dt, flags=fl), ws, ms, 4, ('mfs', 'Pfs'))
        elif kind == 1:
            nh = 2 + (seed // 3) % 2
            dd = 2 * nh + 2
            prm = np.array([0.1, 0.1, 0.1, 1.0, 1.0, 7.0]) * rng.uniform(0.6, 1.5, size=(B, 6))
            _, _, disc, m0c, P0c, Hc = pm.build_harmonic_chirp_model(prm, num_harmonics=nh)
            want = port.filter(port.F_EKF, disc, None, Hc, 0.1, m0c, P0c, dt, ysc)
            moved = port.filter(port.F_EKF, disc, None, Hc, 0.1, m0c, P0c, dt, ysc * (1 + 1e-15 * sign))
            for name, fl in (('wave', 0x2), ('lane', 0x4)):
                compare(f'harmonic{nh} ekf {name}', seed, fs.ekf(disc, Hc, 0.1, m0c, P0c, dt, ysc, flags=fl), want, moved, dd)
            ws = port.smoother(port.S_EKS, disc, None, dt, want[0], want[1])
            ms = port.smoother(port.S_EKS, disc, None, dt, want[0] * (1 + 1e-15), want[1])
            for name, fl in (('wave', 0x2), ('lane', 0x4), ('time-split', 0x2 | 0x800)):
                compare(f'harmonic{nh} eks {name}', seed, fs.eks(disc, want[0], want[1], dt, flags=fl), ws, ms, dd, ('mfs', 'Pfs'))
        else:
            nh = 1 + (seed // 3) % 3
            c = cs.kpt_case(T=max(T, 2), seed=seed, nh=nh, params=tuple(np.array([0.5, 1e-4, 0.1, 8., 1.]) * rng.uniform(0.7, 1.4, size=5)))
            yk = c.ys[None, :T] + 0.05 * rng.standard_normal((B, T))
            spec = pm.linear_cond_m_cov(c.F, c.Sigma)
            spec.model_id, spec.n_harm = pm.M_KPT, nh
            want = port.filter(port.F_EKF_KPT, spec, None, None, c.Xi, c.m0, c.P0, c.dt, yk)
            moved = port.filter(port.F_EKF_KPT, spec, None, None, c.Xi, c.m0, c.P0, c.dt, yk * (1 + 1e-15 * sign))
            for name, fl in (('wave', 0x2), ('lane', 0x4), ('generic', 0x12)):
                compare(f'kpt{nh} {name}', seed, fs.ekf_for_kpt(c.F, c.Sigma, c.h, c.Xi, c.m0, c.P0, c.dt, yk, flags=fl), want, moved, nh + 2)
    print(f'{below} of {arrays} output arrays below 1e-10; worst by kernel family: ' +
          ', '.join(f'{k} {v:.1e}' for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]))
    assert not failures, failures[:8]
