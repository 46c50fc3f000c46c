"""Regenerates the split constants of chirpgp_amd/csrc/cgp_fastmath.hpp at 200 bits (mpmath):
ln 2 = hi (32 bits) + lo, pi/2 = p1 (33 bits) + p2 (33 bits) + p3, log2(e), 2/pi, sqrt(1/2)."""
import mpmath as mp

mp.mp.prec = 200


def split(x, bits):
    m, e = mp.frexp(x)
    hi = mp.ldexp(mp.floor(m * 2 ** bits + mp.mpf(1) / 2), e - bits)
    return float(hi), x - hi


ln2 = mp.log(2)
hi, lo = split(ln2, 32)
print('kLn2Hi     =', repr(hi))
print('kLn2Lo     =', repr(float(lo)))
print('kLog2e     =', repr(float(1 / ln2)))
p1, r1 = split(mp.pi / 2, 33)
p2, r2 = split(r1, 33)
print('kPio2_1    =', repr(p1))
print('kPio2_2    =', repr(p2))
print('kPio2_3    =', repr(float(r2)))
print('kTwoOverPi =', repr(float(2 / mp.pi)))
print('kSqrtHalf  =', repr(float(mp.sqrt(mp.mpf(1) / 2))))


# ---- the lean degree-7 polynomials of the latency-bound kernels (cgp_fastmath.hpp: kExpLean, kLog1pOverTLean) --------------
# Interpolation at the Chebyshev nodes of the interval (near-minimax), solved at 300 bits, coefficients rounded to float64:
#     exp(r) on |r| <= ln 2 / 2 (x 1.0001): relative error 5.5e-11;   log1p(t) / t on [0, exp(-1.5)]: relative error 1.0e-11
# (log1p has its singularity at t = -1: the Chebyshev series converges like 19.9^-n on [0, 0.223]).
def cheb_fit(f, a, b, deg):
    mp.mp.prec = 300
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    V = mp.matrix(n, n)
    for i, x in enumerate(xs):
        for j in range(n):
            V[i, j] = x ** j
    c = mp.lu_solve(V, mp.matrix([f(x) for x in xs]))
    return [float(c[j]) for j in range(n)]


def max_rel_err(f, c, a, b, n=4000):
    return max(abs(sum(cj * (a + (b - a) * mp.mpf(k) / n) ** j for j, cj in enumerate(c)) / f(a + (b - a) * mp.mpf(k) / n) - 1) for k in range(n + 1))


ln2h = mp.log(2) / 2
ex = cheb_fit(mp.exp, -ln2h * mp.mpf('1.0001'), ln2h * mp.mpf('1.0001'), 7)
q = lambda t: mp.log1p(t) / t if t != 0 else mp.mpf(1)
lq = cheb_fit(q, mp.mpf(0), mp.exp(mp.mpf('-1.5')), 7)
print('kExpLean[8] = {' + ', '.join(repr(v) for v in ex) + '};   // max rel err', mp.nstr(max_rel_err(mp.exp, ex, -ln2h, ln2h), 3))
print('kLog1pOverTLean[8] = {' + ', '.join(repr(v) for v in lq) + '};   // max rel err', mp.nstr(max_rel_err(q, lq, mp.mpf(0), mp.exp(mp.mpf('-1.5'))), 3))


# ---- the HIGH-regime polynomials of the d = 4 EKF step (cgp_fastmath.hpp: kExpHigh, kLog1pOverTHigh, kSigmoidHigh) ---------
# Frequency state u2 >= 5 (23 Hz and up at the demos' scaling): t = exp(-u2) <= 6.74e-3.  The degrees were settled against the
# C port on records at the regime's lower edge (tests/test_gpu_parity.py: test_high_frequency_regime_of_the_matrix_core_ekf):
# the angle dt 2 pi fs (u2 + log1p(t)) to ~ 1e-13 rad costs 2.6e-11 in the filtering means at worst; 1 / (1 + t) needs ~ 1e-13
# (the weakly observed frequency-rate state amplifies a bias of the Jacobian's softplus derivative: 1.6e-11 showed as 7e-10).
tmax = mp.exp(mp.mpf(-5))
exh = cheb_fit(mp.exp, -ln2h * mp.mpf('1.0001'), ln2h * mp.mpf('1.0001'), 6)
lqh = cheb_fit(q, mp.mpf(0), tmax, 2)
sgh = cheb_fit(lambda t: 1 / (1 + t), mp.mpf(0), tmax, 4)
print('kExpHigh[7] = {' + ', '.join(repr(v) for v in exh) + '};   // max rel err', mp.nstr(max_rel_err(mp.exp, exh, -ln2h, ln2h), 3))
print('kLog1pOverTHigh[3] = {' + ', '.join(repr(v) for v in lqh) + '};   // max rel err', mp.nstr(max_rel_err(q, lqh, mp.mpf(0), tmax), 3))
print('kSigmoidHigh[5] = {' + ', '.join(repr(v) for v in sgh) + '};   // max rel err', mp.nstr(max_rel_err(lambda t: 1 / (1 + t), sgh, mp.mpf(0), tmax), 3))


# ---- the MID regime of the d = 4 matrix-core EKF (round 5; cgp_fastmath.hpp: kSoftplusMidG / kSoftplusMidH) -------------------------
# For |x| <= 2:  softplus(x) = x / 2 + g(x^2),  g(w) = log(2 cosh(sqrt(w) / 2));   softplus'(x) = 1 / 2 + x h(x^2),  h(w) = tanh(sqrt(w) / 2) / (2 sqrt(w)).
# g and h are analytic in w up to w = -pi^2: interpolation at the Chebyshev nodes of [0, 4.0008] converges like 7.7^-n; degree 14 leaves
# 2.5e-16 (softplus) and 4.3e-16 (derivative) relative, i.e. the rounding of the coefficients.
def fit_mid():
    def g(w):
        return mp.log(2 * mp.cosh(mp.sqrt(w) / 2)) if w > 0 else mp.log(2)

    def h(w):
        return mp.tanh(mp.sqrt(w) / 2) / (2 * mp.sqrt(w)) if w > 0 else mp.mpf(1) / 4

    mp.mp.prec = 400
    deg, a, b = 14, mp.mpf(0), mp.mpf(4) * mp.mpf('1.0002')
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    V = mp.matrix(n, n)
    for i, x in enumerate(xs):
        for j in range(n):
            V[i, j] = x ** j
    for name, f in (('kSoftplusMidG', g), ('kSoftplusMidH', h)):
        c = mp.lu_solve(V, mp.matrix([f(x) for x in xs]))
        print(f'constexpr double {name}[15] = {{' + ', '.join(repr(float(v)) for v in c) + '};')


fit_mid()
