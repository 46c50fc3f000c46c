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
