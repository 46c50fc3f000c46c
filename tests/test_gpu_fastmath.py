"""Accuracy of the engine's in-kernel float64 elementary functions (csrc/cgp_fastmath.hpp) against a 200-bit
reference (mpmath), through the C-ABI test hook cgp_debug_math.  Bar: a few ulp -- 1e-14 relative here, nine orders
inside the path's 1e-5 gate."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulp_err(got, want_mp):
    import mpmath as mp
    errs = []
    for g, w in zip(got, want_mp):
        if w == 0:
            errs.append(abs(g))
        else:
            errs.append(float(abs((mp.mpf(float(g)) - w) / w)))
    return max(errs)


def test_exp():
    import mpmath as mp
    from chirpgp_amd import _engine as E
    mp.mp.prec = 200
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 40, 3000), rng.uniform(-700, 700, 500), [0., 1., -1., 709.7, -708., 1e-300]])
    got, _ = E.debug_math(0, x)
    assert _ulp_err(got, [mp.exp(mp.mpf(float(v))) for v in x]) < 1e-14
    sp, _ = E.debug_math(0, np.array([710., 1e4, np.inf, -746., -np.inf, np.nan]))
    assert sp[0] == np.inf and sp[1] == np.inf and sp[2] == np.inf and sp[3] == 0. and sp[4] == 0. and np.isnan(sp[5])
    np.testing.assert_array_equal(sp[:5], np.exp(np.array([710., 1e4, np.inf, -746., -np.inf])))


def test_log_ge1_and_softplus():
    import mpmath as mp
    from chirpgp_amd import _engine as E
    mp.mp.prec = 200
    rng = np.random.default_rng(1)
    z = np.concatenate([1 + np.exp(rng.uniform(-36, 40, 3000)), rng.uniform(1, 4, 500), 10 ** rng.uniform(0, 300, 500),
                        [1., 1. + 2 ** -52, 2., np.sqrt(2.), 1.7976931348623157e308]])
    got, _ = E.debug_math(1, z)
    assert _ulp_err(got, [mp.log(mp.mpf(float(v))) for v in z]) < 1e-14
    inf, _ = E.debug_math(1, np.array([np.inf, np.nan]))
    assert inf[0] == np.inf and np.isnan(inf[1])
    # softplus pair in the reference's naive form, incl. its overflow behaviour (models.py:50)
    x = np.concatenate([rng.uniform(-30, 60, 2000), [7., 0., 708., 710., 800., -800.]])
    sp, dsp = E.debug_math(4, x)
    with np.errstate(over='ignore', invalid='ignore'):
        e = np.exp(x)
        want_sp, want_d = np.log(e + 1.), e / (e + 1.)
    assert np.array_equal(np.isnan(dsp), np.isnan(want_d)) and np.array_equal(np.isinf(sp), np.isinf(want_sp))
    ok = np.isfinite(want_sp) & (want_sp > 0)
    assert np.max(np.abs(sp[ok] - want_sp[ok]) / want_sp[ok]) < 1e-14
    okd = np.isfinite(want_d)
    assert np.max(np.abs(dsp[okd] - want_d[okd]) / np.maximum(want_d[okd], 1e-300)) < 1e-14


def test_uniform_variants():
    """The wave-uniform softplus (regime split at x = 6: x + log1p(exp(-x))) and sincos used by the cooperative EKF."""
    import mpmath as mp
    from chirpgp_amd import _engine as E
    mp.mp.prec = 200
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-30, 60, 1500), rng.uniform(5.9, 6.1, 200), rng.uniform(690, 712, 100),
                        [6., 5.999999999, 7., 36., 37., 699.99, 700., 708., 710., 800., -800., np.inf, -np.inf, np.nan]])
    sp, dsp = E.debug_math(5, x)
    with np.errstate(over='ignore', invalid='ignore'):
        e = np.exp(x)
        ref_sp, ref_d = np.log(e + 1.), e / (e + 1.)
    assert np.array_equal(np.isnan(dsp), np.isnan(ref_d)) and np.array_equal(np.isinf(sp), np.isinf(ref_sp))
    assert np.array_equal(np.isnan(sp), np.isnan(ref_sp))
    # against the naive form as NumPy evaluates it (for very negative x the naive form itself is inaccurate: models.py:50)
    fin = np.isfinite(ref_sp) & np.isfinite(x) & (ref_sp > 0)
    # (a 1-ulp difference in exp(x) moves 1 + exp(x) by an ulp of 1, i.e. log(.) by 2e-16 / sp relative: 1e-13 bound)
    assert np.max(np.abs(sp[fin] - ref_sp[fin]) / ref_sp[fin]) < 1e-13
    okd = fin & np.isfinite(ref_d)
    assert np.max(np.abs(dsp[okd] - ref_d[okd]) / ref_d[okd]) < 1e-14
    # and against the exact value where the naive form is accurate (x >= 0), which covers the regime-split branch
    pos = fin & (x >= 0)
    assert _ulp_err(sp[pos], [mp.log(mp.exp(mp.mpf(float(v))) + 1) for v in x[pos]]) < 1e-15
    posd = okd & (x >= 0)
    assert _ulp_err(dsp[posd], [1 / (1 + mp.exp(-mp.mpf(float(v)))) for v in x[posd]]) < 1e-15
    xs = np.concatenate([rng.uniform(-10, 10, 1500), rng.uniform(-9e4, 9e4, 500), [0., 0.044, 99999.9, 1.0e5, 3.3e7, np.inf, np.nan]])
    sn, cs = E.debug_math(6, xs)
    f = np.isfinite(xs)
    assert max(abs(float(mp.mpf(float(g)) - mp.sin(mp.mpf(float(v))))) for g, v in zip(sn[f], xs[f])) < 3e-16
    assert max(abs(float(mp.mpf(float(g)) - mp.cos(mp.mpf(float(v))))) for g, v in zip(cs[f], xs[f])) < 3e-16
    assert np.all(np.isnan(sn[~f])) and np.all(np.isnan(cs[~f]))


def test_wide_softplus_forms():
    """The common-regime softplus x + t q(t), t = exp(-x), q = log1p(t) / t with the lean degree-7 polynomials (7e-12): per
    lane with the naive fallback for lanes outside [1.5, 700) (mixed in the same wavefronts), and the speculative step's form."""
    import mpmath as mp
    from chirpgp_amd import _engine as E
    mp.mp.prec = 200
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0.6, 60, 3000), rng.uniform(-30, 1.6, 800), rng.uniform(1.49, 1.51, 200), rng.uniform(690, 712, 100),
                        [1.5, 1.4999999999, 0.6931471805599453, 699.99, 700., 710., 800., -800., np.inf, -np.inf, np.nan]])
    rng.shuffle(x)                                  # common and uncommon lanes side by side in every wavefront
    sp, dsp = E.debug_math(7, x)
    with np.errstate(over='ignore', invalid='ignore'):
        e = np.exp(x)
        ref_sp, ref_d = np.log(e + 1.), e / (e + 1.)
    assert np.array_equal(np.isnan(dsp), np.isnan(ref_d)) and np.array_equal(np.isnan(sp), np.isnan(ref_sp))
    assert np.array_equal(np.isinf(sp), np.isinf(ref_sp))
    fin = np.isfinite(ref_sp) & np.isfinite(x) & (ref_sp > 0)
    assert np.max(np.abs(sp[fin] - ref_sp[fin]) / ref_sp[fin]) < 1e-11
    okd = fin & np.isfinite(ref_d)
    assert np.max(np.abs(dsp[okd] - ref_d[okd]) / ref_d[okd]) < 2e-11
    pos = fin & (x >= 0)
    assert _ulp_err(sp[pos], [mp.log(mp.exp(mp.mpf(float(v))) + 1) for v in x[pos]]) < 1e-11
    assert _ulp_err(dsp[pos & okd], [1 / (1 + mp.exp(-mp.mpf(float(v)))) for v in x[pos & okd]]) < 2e-11
    out = fin & ~((x >= 1.5) & (x < 700))            # outside the lean regime: the full-precision naive form
    assert np.max(np.abs(sp[out] - ref_sp[out]) / ref_sp[out]) < 1e-13
    # the speculative step's LEAN pair inside its regime [1.5, 700): degree-7 polynomials, 7e-12 on the softplus (the chain
    # of the bench kernel trades five orders of unused accuracy for two Estrin levels; cgp_fastmath.hpp)
    xr = np.concatenate([rng.uniform(1.5, 60, 2000), rng.uniform(1.5, 3.0, 1000), rng.uniform(60, 699, 200), [1.5, 699.999]])
    sp, dsp = E.debug_math(8, xr)
    assert _ulp_err(sp, [mp.log(mp.exp(mp.mpf(float(v))) + 1) for v in xr]) < 1e-11
    assert _ulp_err(dsp, [1 / (1 + mp.exp(-mp.mpf(float(v)))) for v in xr]) < 2e-11
    d = np.concatenate([10 ** rng.uniform(-8, 8, 2000) * rng.choice([-1, 1], 2000)])
    r1, _ = E.debug_math(9, d)
    assert np.max(np.abs(r1 * d - 1)) < 5e-15


def test_sincos():
    import mpmath as mp
    from chirpgp_amd import _engine as E
    mp.mp.prec = 300
    rng = np.random.default_rng(2)
    k = rng.integers(-60000, 60000, 400)
    x = np.concatenate([rng.uniform(-10, 10, 3000), rng.uniform(-9e4, 9e4, 1500), k * (np.pi / 2) + rng.uniform(-1e-6, 1e-6, 400),
                        [0., 0.044, np.pi / 4, -np.pi / 4, 99999.9, 1.0e5, 3.3e7, 1e15]])
    sn, cs = E.debug_math(2, x)
    want_s = [mp.sin(mp.mpf(float(v))) for v in x]
    want_c = [mp.cos(mp.mpf(float(v))) for v in x]
    # absolute error (the values feed rotations: what matters is |error| vs 1), plus relative where not tiny
    assert max(abs(float(mp.mpf(float(g)) - w)) for g, w in zip(sn, want_s)) < 3e-16
    assert max(abs(float(mp.mpf(float(g)) - w)) for g, w in zip(cs, want_c)) < 3e-16
    sp, cp = E.debug_math(2, np.array([np.inf, np.nan]))
    assert np.all(np.isnan(sp)) and np.all(np.isnan(cp))


def test_rcp():
    from chirpgp_amd import _engine as E
    rng = np.random.default_rng(3)
    x = np.concatenate([10 ** rng.uniform(-200, 200, 3000) * rng.choice([-1, 1], 3000), [1., 0.1, 3.]])
    got, _ = E.debug_math(3, x)
    assert np.max(np.abs(got * x - 1.0)) < 4e-16


def test_branch_free_softplus_pair_for_any_argument():
    """softplus_pair_any (cgp_fastmath.hpp; the WIDE step of the matrix-core EKF): log(exp(x) + 1) and its derivative for every finite
    |x| < 700 -- negative arguments, the crossover at 0, the region where exp(-|x|) vanishes against 1 -- to a few ulp of a 200-bit
    reference; |x| >= 700, inf and NaN are reported invalid (the kernel then takes the reference's naive form)."""
    import mpmath as mp
    from chirpgp_amd import _engine as E
    mp.mp.prec = 200
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-50, 50, 4000), rng.uniform(-699, 699, 1000), rng.uniform(-2, 2, 1000),
                        [0., 1e-300, -1e-300, 1.5, -1.5, 36.7, -36.7, 40., -40., 699.9, -699.9, np.log(np.sqrt(2.) - 1.), np.log(np.sqrt(2.) - 1.) + 1e-12]])
    sp, dsp = E.debug_math(10, x)
    assert _ulp_err(sp, [mp.log1p(mp.exp(mp.mpf(float(v)))) for v in x]) < 2e-15          # (log1p: at 200 bits 1 + exp(-300) is 1)
    assert _ulp_err(dsp, [1 / (1 + mp.exp(-mp.mpf(float(v)))) for v in x]) < 2e-15
    bad, _ = E.debug_math(10, np.array([700., -700., 1e4, np.inf, -np.inf, np.nan]))
    assert np.isnan(bad).all()


def test_mid_band_softplus_pair_as_polynomials():
    """softplus_pair_mid (cgp_fastmath.hpp; the MID regime of the matrix-core EKF's speculative step): for |x| <= 2 the softplus is
    x / 2 + g(x^2) and its derivative 1 / 2 + x h(x^2), g and h degree-14 polynomials -- no exp, no log, no reciprocal -- to a few ulp
    of a 200-bit reference (coefficients: tools/gen_math_constants.py)."""
    import mpmath as mp
    from chirpgp_amd import _engine as E
    mp.mp.prec = 200
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(-2, 2, 6000), [0., 1e-300, -1e-300, 2., -2., 1.75, -1.75, 1e-8, -1e-8]])
    sp, dsp = E.debug_math(11, x)
    # ABSOLUTE accuracy is what the filter needs of them (the rotation angle is a multiple of the softplus, the Jacobian column of the
    # derivative): a few 1e-16 (two ulp).  Relative to a softplus that is itself small -- 0.127 at x = -2, where x / 2 = -1 and g = 1.127 cancel --
    # that is up to 3e-15.
    want_sp = [mp.log1p(mp.exp(mp.mpf(float(v)))) for v in x]
    want_d = [1 / (1 + mp.exp(-mp.mpf(float(v)))) for v in x]
    assert max(abs(float(mp.mpf(float(g)) - w)) for g, w in zip(sp, want_sp)) < 1e-15          # (2 ulp at 2.1)
    assert max(abs(float(mp.mpf(float(g)) - w)) for g, w in zip(dsp, want_d)) < 5e-16
    assert _ulp_err(sp, want_sp) < 4e-15 and _ulp_err(dsp, want_d) < 3e-15
