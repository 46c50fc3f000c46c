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


# ---- log1p(t) / t on [0, 1/2] as a polynomial of degree 15 (softplus tail of the speculative EKF step) ----------------
# Interpolation at the Chebyshev nodes of the interval, solved at 300 bits, coefficients rounded to float64; the maximum
# relative error of the exact-coefficient polynomial is 1.7e-17 (log1p has its singularity at t = -1: the Chebyshev
# series converges like 5.8^-n on [0, 1] and 9.9^-n on [0, 1/2]).
def log1p_over_t_coefficients(b=0.5, n=15):
    mp.mp.prec = 300
    f = lambda t: mp.mpf(1) if t == 0 else mp.log1p(t) / t
    nodes = [mp.mpf(b) / 2 * (1 + mp.cos(mp.pi * (2 * i + 1) / (2 * (n + 1)))) for i in range(n + 1)]
    V = mp.matrix(n + 1, n + 1)
    for i, t in enumerate(nodes):
        for j in range(n + 1):
            V[i, j] = t ** j
    c = mp.lu_solve(V, mp.matrix([f(t) for t in nodes]))
    return [float(c[j]) for j in range(n + 1)]


print('kLog1pOverT[16] = {')
for v in log1p_over_t_coefficients():
    print('    %r,' % v)
print('};')
