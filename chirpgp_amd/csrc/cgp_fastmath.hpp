// cgp_fastmath.hpp -- lean float64 elementary functions for the serial part of the filters.
//
// A filter step is a strict dependency chain through exp -> log -> sincos (softplus frequency, rotation angle) and a
// reciprocal (Kalman gain).  Measured on MI355X (tools/ubench/f64_issue.hip), one wave per SIMD, dependent chain:
//     v_fma_f64 7 cycles | ocml exp 123 | ocml log 412 | ocml sincos 364 | f64 divide 74 | sqrt 107
// so the library calls alone cost ~1000 of the ~3700 cycles of a v1 EKF step.  The versions below are branch-free
// (one rarely taken fallback in sincos), have no denormal / errno / special-case paths beyond what the reference's
// arithmetic can produce (+inf from exp overflow, NaN), and are accurate to a few ulp -- far inside the 1e-5 gate
// and checked against a 200-bit reference in tests/test_gpu_fastmath.py.
//
// Constants were generated with mpmath at 200 bits (tools/gen_math_constants.py).
#pragma once
#include "cgp_math.hpp"

namespace cgp {

constexpr double kLog2e = 1.4426950408889634;
constexpr double kLn2Hi = 0.6931471806019545;        // 32 significant bits: k * kLn2Hi is exact for |k| < 2^21
constexpr double kLn2Lo = -4.2009150726810846e-11;
constexpr double kTwoOverPi = 0.6366197723675814;
constexpr double kPio2_1 = 1.5707963267341256;       // 33 bits
constexpr double kPio2_2 = 6.077100506303966e-11;    // 33 bits
constexpr double kPio2_3 = 2.0222662487959506e-21;
constexpr double kSqrtHalf = 0.70710678118654752440;

// p * x + c as a three-address v_fma_f64 (a translation unit whose coefficients are all pinned in registers may define
// CGP_HORNER_PLAIN and take the compiler's own fma: cgp_inst_ekf4.hip).  hipcc selects the two-address v_fmac_f64 for fma(p, x, CONSTANT) and then has to
// copy the loop-invariant constant into the destination first (v_mov_b64 + v_fmac_f64 per Horner step, seen in the
// ISA of the v2 kernels); the explicit form keeps every coefficient in its own VGPR pair and issues one instruction.
CGP_DEV double horner(double p, double x, double c) {
#ifdef CGP_HORNER_PLAIN
    return fma(p, x, c);
#else
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(x), "v"(c));
    return d;
#endif
}

// A Horner step whose addend is a compile-time CONSTANT.  CGP_HORNER_SGPR (the large-batch lane kernels, cgp_inst_lane4.hip): the constant as
// the instruction's one SCALAR operand -- three-address, no register pair pinned per coefficient, and none of the v_mov_b64 the compiler
// puts in front of a two-address v_fmac_f64 to load the constant into the destination (58 of them per step in the lane EKF; the scalar
// moves that materialise the constants issue beside the other wavefront's vector instructions).
CGP_DEV double horner_c(double p, double x, double c) {
#ifdef CGP_HORNER_SGPR
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(x), "s"(c));
    return d;
#else
    return horner(p, x, c);
#endif
}

// (A whole chain as ONE assembly statement, every coefficient moved into VCC as two literals in front of its v_fma_f64, was measured in
// round 5: no hoisted constants, scalar spills 128 -> 38, hazard padding 195 -> 26 -- and 7 % slower: the scalar moves take the wavefront's
// issue slots on its dependent chain.)
// 1 / d by v_rcp_f64 and two Newton steps (full double accuracy for normal d); 0 -> NaN, inf -> NaN, NaN -> NaN.
CGP_DEV double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}

// 1 / d by v_rcp_f64 (relative error 4.6e-8 on MI355X, tools/ubench/rcp_accuracy.hip) and ONE Newton step: 2.2e-15.
// For quantities on a filter's serial chain whose own conditioning is far worse than that (the reciprocal innovation
// variance, the softplus derivative): two dependent FMAs less than rcp_nr.
CGP_DEV double rcp_nr1(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    return fma(r, fma(-d, r, 1.0), r);
}

// n / d with one residual correction (<= 1 ulp).
CGP_DEV double div_nr(double n, double d) {
    const double r = rcp_nr(d);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}
// the same from the one-step reciprocal: its 2.2e-15 goes into the quotient squared by the residual correction (<= 1 ulp), two FMAs less
CGP_DEV double div_nr1(double n, double d) {
    const double r = rcp_nr1(d);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// sum_{i <= 13} r^i / i!, Horner.  (Estrin's scheme was measured here too: neutral in the sigma-point and lane-per-trial
// kernels, which have other work to overlap; it pays only in the cooperative EKF, see softplus_pair_uniform below.)
constexpr double kExpTaylor[14] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                                   1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
CGP_DEV double exp_poly(double r) {
    double p = kExpTaylor[0];
    CGP_UNROLL for (int i = 1; i < 14; i++) p = horner_c(p, r, kExpTaylor[i]);
    return p;
}

// exp(x): x = k ln2 + r, Taylor of degree 13 on |r| <= ln2 / 2 (truncation 4e-18), v_ldexp_f64.
// Overflows to +inf above 709.78 like libm; NaN in -> NaN out.
CGP_DEV double fast_exp(double x) {
    const double k = __builtin_rint(x * kLog2e);
    double r = fma(-k, kLn2Hi, x);
    r = fma(-k, kLn2Lo, r);
    const double p = exp_poly(r);
    double y = __builtin_amdgcn_ldexp(p, (int)k);
    y = (x > 709.782712893384) ? __builtin_inf() : y;
    y = (x < -745.2) ? 0.0 : y;
    return y;
}

// 1/3 + s2/5 + ... + s2^9/21 (the atanh series of fast_log_ge1 without its leading 1), Horner
constexpr double kAtanhOdd[10] = {1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0};
CGP_DEV double atanh_tail_poly(double s2) {
    double p = kAtanhOdd[0];
    CGP_UNROLL for (int i = 1; i < 10; i++) p = horner_c(p, s2, kAtanhOdd[i]);
    return p;
}

// log(z) for z in [1, +inf] (the softplus argument exp(x) + 1): z = 2^k m, m in [sqrt(1/2), sqrt(2)),
// s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + s^4/5 + ... + s^20/21)   (|s| <= 0.1716, truncation 2e-17).
CGP_DEV double fast_log_ge1(double z) {
    double m = __builtin_amdgcn_frexp_mant(z);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(z);
    const bool small = m < kSqrtHalf;
    m = small ? m + m : m;
    e = small ? e - 1 : e;
    const double k = (double)e;
    const double f = m - 1.0;
    const double s = div_nr(f, m + 1.0);
    const double s2 = s * s;
    const double p = atanh_tail_poly(s2);
    const double two_s = s + s;
    double lm = fma(two_s, p * s2, two_s);
    lm = fma(k, kLn2Lo, lm);
    double y = fma(k, kLn2Hi, lm);
    return (z == __builtin_inf()) ? z : y;
}

// fast_log_ge1 for a FINITE z >= 1 with the exponent taken so that the mantissa lands in [sqrt(1/2), sqrt(2)) directly: k = exponent of
// z sqrt(2), m = z 2^-k (exact) -- five instructions where frexp + the compare-and-select adjustment of fast_log_ge1 take nine.  (The
// rounding of z sqrt(2) can leave m an ulp outside the interval at its ends: the series does not care.)
CGP_DEV double fast_log_ge1_finite(double z) {
    const int e = __builtin_amdgcn_frexp_exp(z * 1.4142135623730951) - 1;
    const double m = __builtin_amdgcn_ldexp(z, -e);
    const double k = (double)e;
    const double s = div_nr1(m - 1.0, m + 1.0);
    const double s2 = s * s;
    const double p = atanh_tail_poly(s2);
    const double two_s = s + s;
    double lm = fma(two_s, p * s2, two_s);
    lm = fma(k, kLn2Lo, lm);
    return fma(k, kLn2Hi, lm);
}

// sin(r), cos(r) on the reduced range |r| <= pi/4: Taylor to r^17 / r^16 (truncation 5e-17), two independent Horner chains.
constexpr double kSinTaylor[8] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0, 1.0 / 39916800.0, -1.0 / 362880.0, 1.0 / 5040.0,
                                  -1.0 / 120.0, 1.0 / 6.0};                  // -1/17! ... 1/3! (sign folded below)
constexpr double kCosTaylor[7] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0};
CGP_DEV void sincos_reduced(double r, double& s0, double& c0) {
    const double r2 = r * r;
    double ps = kSinTaylor[0], pc = kCosTaylor[0];
    CGP_UNROLL for (int i = 1; i < 8; i++) ps = horner_c(ps, r2, kSinTaylor[i]);
    CGP_UNROLL for (int i = 1; i < 7; i++) pc = horner_c(pc, r2, kCosTaylor[i]);
    s0 = fma(-(r * r2), ps, r);                       // r - r^3 (1/3! - r^2/5! + ...)
    c0 = fma(r2 * r2, pc, fma(-0.5, r2, 1.0));        // 1 - r^2/2 + r^4 (1/4! - ...)
}
constexpr double kPiOver4 = 0.78539816339744830962;

// sin(x), cos(x): 3-term Cody-Waite reduction by pi/2 for |x| < 1e5 (n < 2^16: n * kPio2_1, n * kPio2_2 exact),
// Taylor to r^17 / r^16 on |r| <= pi/4 (truncation 5e-17).  Larger |x|, inf and NaN take the library's values.
// The reduced-range evaluation runs unconditionally; only if some active lane is outside the range does the wavefront
// also call the library and those lanes take its result -- one wave-uniform branch at the END (a branch in front would
// cut the caller's dependent chain into basic blocks that cannot be interleaved; see cgp_mfma4.hpp).  Signs are flipped
// on the high word instead of negate-and-select.
CGP_DEV void fast_sincos(double x, double& sn, double& cs) {
    const bool common = fabs(x) < 1.0e5;               // false for inf and NaN
    const double n = __builtin_rint(x * kTwoOverPi);
    double r = fma(-n, kPio2_1, x);
    r = fma(-n, kPio2_2, r);
    r = fma(-n, kPio2_3, r);
    double s0, c0;
    sincos_reduced(r, s0, c0);
    const int q = (int)n;
    const bool swap = (q & 1) != 0;
    const double a = swap ? c0 : s0, b = swap ? s0 : c0;
    sn = __hiloint2double(__double2hiint(a) ^ ((q & 2) << 30), __double2loint(a));
    cs = __hiloint2double(__double2hiint(b) ^ (((q + 1) & 2) << 30), __double2loint(b));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!common) != 0, 0)) {
        double sl, cl;
        sincos(x, &sl, &cl);
        sn = common ? sn : sl;
        cs = common ? cs : cl;
    }
}

// fast_sincos for kernels whose angles are small throughout (one lane per trial at dt = 0.01: the rotation angle is dt 2 pi g(u) ~ 0.06 g):
// if every active lane's |x| <= pi / 4 the Cody-Waite reduction is the identity (n = 0, r = x, no swap, no sign flip) and is skipped -- the
// same bits, ~ 18 instructions less; one wave-uniform branch, otherwise fast_sincos as is (NaN and inf take that way).
// ... and while every lane's |x| <= 1/4 (a frequency below 4 Hz at that step) the series end four terms earlier: sin to x^11, cos to x^12
// (truncation 2.4e-18 / 4e-20), five instructions less.
CGP_DEV void sincos_quarter(double r, double& s0, double& c0) {
    const double r2 = r * r;
    double ps = kSinTaylor[3], pc = kCosTaylor[2];                       // 1/11!, 1/12!
    CGP_UNROLL for (int i = 4; i < 8; i++) ps = horner_c(ps, r2, kSinTaylor[i]);
    CGP_UNROLL for (int i = 3; i < 7; i++) pc = horner_c(pc, r2, kCosTaylor[i]);
    s0 = fma(-(r * r2), ps, r);
    c0 = fma(r2 * r2, pc, fma(-0.5, r2, 1.0));
}
CGP_DEV void fast_sincos_small(double x, double& sn, double& cs) {
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(fabs(x) <= 0.25)) == 0, 1)) { sincos_quarter(x, sn, cs); return; }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(fabs(x) <= kPiOver4)) == 0, 1)) { sincos_reduced(x, sn, cs); return; }
    fast_sincos(x, sn, cs);
}

// Softplus log(exp(x) + 1) and its derivative for ANY finite |x| < 700 without a branch, in Estrin form (round 5: the middle tier of the
// matrix-core EKF, cgp_mfma4.hpp -- records whose frequency state leaves the lean regime x >= 1.5 used to fall all the way to the naive
// exp + log with their regime branches, three times the step): with t = exp(-|x|) in (0, 1], z = 1 + t in (1, 2],
//     log(exp(x) + 1) = max(x, 0) + log1p(t),      exp(x) / (exp(x) + 1) = (x > 0 ? 1 : t) / z,
// log z by the atanh series of fast_log_ge1 (no exponent extraction: z or z / 2 lies in (1 / sqrt 2, sqrt 2]), log1p(t) = log z + (t - (z - 1)) / z
// (the rounding of 1 + t put back: exact to the last bits also where t vanishes against 1).  ~1e-16 like the naive form; ok = false
// for |x| >= 700, inf and NaN, where the reference's form overflows (the caller repeats those with the checked step).
CGP_DEV void softplus_pair_any(double x, double& sp, double& dsp, bool& ok) {
    const double a = fabs(x);
    ok = a < 700.0;
    const double na = -a;
    const double k = __builtin_rint(na * kLog2e);
    double r = fma(-k, kLn2Hi, na);
    r = fma(-k, kLn2Lo, r);
    // exp(r), |r| <= ln2 / 2: sum r^i / i!, i <= 13, Estrin
    const double r2 = r * r;
    const double e0 = fma(r, 1.0, 1.0), e1 = fma(r, 1.0 / 6.0, 0.5), e2 = fma(r, 1.0 / 120.0, 1.0 / 24.0), e3 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
    const double e4 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0), e5 = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0), e6 = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    const double r4 = r2 * r2;
    const double f0 = fma(e1, r2, e0), f1 = fma(e3, r2, e2), f2 = fma(e5, r2, e4);
    const double r8 = r4 * r4;
    const double g0 = fma(f1, r4, f0), g1 = fma(e6, r4, f2);
    const double t = __builtin_amdgcn_ldexp(fma(g1, r8, g0), (int)k);
    const double z = 1.0 + t;
    const double rz = rcp_nr(z);
    const bool big = z > 1.4142135623730951;
    const double m = big ? 0.5 * z : z;
    const double s = (m - 1.0) * rcp_nr(m + 1.0);
    const double sc = fma(fma(-(m + 1.0), s, m - 1.0), rcp_nr1(m + 1.0), s);          // one residual correction of the quotient (<= 1 ulp)
    const double s2 = sc * sc;
    // 1/3 + s2/5 + ... + s2^9/21, Estrin
    const double p0 = fma(s2, 1.0 / 5.0, 1.0 / 3.0), p1 = fma(s2, 1.0 / 9.0, 1.0 / 7.0), p2 = fma(s2, 1.0 / 13.0, 1.0 / 11.0);
    const double p3 = fma(s2, 1.0 / 17.0, 1.0 / 15.0), p4 = fma(s2, 1.0 / 21.0, 1.0 / 19.0);
    const double s4 = s2 * s2;
    const double q0 = fma(p1, s4, p0), q1 = fma(p3, s4, p2);
    const double s8 = s4 * s4;
    const double pl = fma(fma(p4, s8, q1), s8, q0);
    const double two_s = sc + sc;
    double lz = fma(two_s, pl * s2, two_s);
    lz = big ? fma(1.0, kLn2Lo, lz) + kLn2Hi : lz;
    const double l1p = fma(t - (z - 1.0), rz, lz);
    sp = (x > 0.0 ? x : 0.0) + l1p;
    dsp = (x > 0.0 ? 1.0 : t) * rz;
}

// fast_sincos without its fallback branch: ok = false for |x| >= 1e5, inf and NaN (the caller repeats with fast_sincos).
CGP_DEV void fast_sincos_spec(double x, double& sn, double& cs, bool& ok) {
    ok = fabs(x) < 1.0e5;
    const double n = __builtin_rint(x * kTwoOverPi);
    double r = fma(-n, kPio2_1, x);
    r = fma(-n, kPio2_2, r);
    r = fma(-n, kPio2_3, r);
    double s0, c0;
    sincos_reduced(r, s0, c0);
    const int q = (int)n;
    const bool swap = (q & 1) != 0;
    const double a = swap ? c0 : s0, b = swap ? s0 : c0;
    sn = __hiloint2double(__double2hiint(a) ^ ((q & 2) << 30), __double2loint(a));
    cs = __hiloint2double(__double2hiint(b) ^ (((q + 1) & 2) << 30), __double2loint(b));
}

// exp(x) without the overflow / underflow / NaN handling, for arguments known to lie in (-700, 700).
CGP_DEV double fast_exp_core(double x) {
    const double k = __builtin_rint(x * kLog2e);
    double r = fma(-k, kLn2Hi, x);
    r = fma(-k, kLn2Lo, r);
    const double p = exp_poly(r);
    return __builtin_amdgcn_ldexp(p, (int)k);
}

// sin / cos for a WAVE-UNIFORM argument: the fallback test is one scalar compare on the exponent bits.
CGP_DEV void fast_sincos_uniform(double x, double& sn, double& cs) {
    const int hx = __builtin_amdgcn_readfirstlane(__double2hiint(x)) & 0x7fffffff;
    if (__builtin_expect(hx >= 0x40F86A00, 0)) {            // |x| >= 1e5, inf, NaN (out of line: rare)
        sincos(x, &sn, &cs);
        return;
    }
    const double n = __builtin_rint(x * kTwoOverPi);
    double r = fma(-n, kPio2_1, x);
    r = fma(-n, kPio2_2, r);
    r = fma(-n, kPio2_3, r);
    const double r2 = r * r;
    double ps = -1.0 / 355687428096000.0;
    ps = horner(ps, r2, 1.0 / 1307674368000.0);
    ps = horner(ps, r2, -1.0 / 6227020800.0);
    ps = horner(ps, r2, 1.0 / 39916800.0);
    ps = horner(ps, r2, -1.0 / 362880.0);
    ps = horner(ps, r2, 1.0 / 5040.0);
    ps = horner(ps, r2, -1.0 / 120.0);
    ps = horner(ps, r2, 1.0 / 6.0);
    double pc = 1.0 / 20922789888000.0;
    pc = horner(pc, r2, -1.0 / 87178291200.0);
    pc = horner(pc, r2, 1.0 / 479001600.0);
    pc = horner(pc, r2, -1.0 / 3628800.0);
    pc = horner(pc, r2, 1.0 / 40320.0);
    pc = horner(pc, r2, -1.0 / 720.0);
    pc = horner(pc, r2, 1.0 / 24.0);
    const double s0 = fma(-(r * r2), ps, r);
    const double c0 = fma(r2 * r2, pc, fma(-0.5, r2, 1.0));
    // quadrant: swap by selects, signs by flipping the sign bit with a wave-uniform mask
    const int q = __builtin_amdgcn_readfirstlane((int)n);
    const bool swap = (q & 1) != 0;
    const double a = swap ? c0 : s0, b = swap ? s0 : c0;
    const int sa = (q & 2) << 30, sb = ((q + 1) & 2) << 30;
    sn = __hiloint2double(__double2hiint(a) ^ sa, __double2loint(a));
    cs = __hiloint2double(__double2hiint(b) ^ sb, __double2loint(b));
}

// softplus log(exp(x) + 1) and its derivative exp(x) / (exp(x) + 1) for a WAVE-UNIFORM x (one scalar branch).
// For 6 <= x < 700 -- every frequency above g(6) = 6.0025 Hz -- it uses the algebraically identical
//     log(exp(x) + 1) = x + log1p(t),  exp(x) / (exp(x) + 1) = 1 / (1 + t),  t = exp(-x) <= 2.5e-3,
// with the 6-term series of log1p (truncation t^7 / 7 < 1e-19): one exp and one reciprocal instead of exp, a full
// log and a reciprocal.  Both forms are within an ulp or two of the exact value; elsewhere (and for inf / NaN) the
// naive form of models.py:50 is evaluated as is, overflow behaviour included.
CGP_DEV void softplus_pair_uniform(double x, double& sp, double& dsp) {
    const int hx = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    // 40 <= x < 700: t = exp(-x) < 4.3e-18 is below half an ulp of x and of 1, so the form below returns exactly (x, 1) -- without
    // evaluating it.  (The KPT measurement's argument is frequency + accumulated PHASE, models.py:575-578: beyond 40 after a few hundred
    // steps of every record; the exp / reciprocal chain was a third of that filter's step.)
    if (hx >= 0x40440000 && hx < 0x4085E000) { sp = x; dsp = 1.0; return; }
    if (__builtin_expect(hx >= 0x40180000 && hx < 0x4085E000, 1)) {   // 6.0 <= x < 700.0 (positive doubles order like their bits)
        const double t = fast_exp_core(-x);
        double p = -1.0 / 6.0;
        p = horner(p, t, 0.2);
        p = horner(p, t, -0.25);
        p = horner(p, t, 1.0 / 3.0);
        p = horner(p, t, -0.5);
        p = horner(p, t, 1.0);
        sp = fma(p, t, x);
        dsp = rcp_nr(1.0 + t);
        return;
    }
    const double e = fast_exp(x);
    const double z = e + 1.0;
    sp = fast_log_ge1(z);
    dsp = e * rcp_nr(z);
}

// The polynomial coefficients of the wave-uniform softplus / sincos pinned in VGPRs for the lifetime of a kernel.
// Floating-point immediates are "free to rematerialise" for the compiler, so inside the step loop it rebuilds each
// coefficient with s_mov + v_mov pairs right before the v_fma that uses it (18 v_mov + 9 hazard nops per step in the
// cooperative EKF); passing every constant once through an empty asm makes it an opaque loop-invariant value that
// simply stays in its register pair (46 pairs).
struct FastMathRegs {
    double ex[14];      // 1/13! ... 1/2!, 1, 1     (exp Taylor, Horner order)
    double lp[6];       // -1/6, 1/5, -1/4, 1/3, -1/2, 1   (log1p)
    double sn[8];       // -1/17!, 1/15!, ..., 1/3!
    double cs[7];       // 1/16!, -1/14!, ..., 1/4!
    double log2e, ln2hi, ln2lo, two_over_pi, pio2_1, pio2_2, pio2_3;
    CGP_DEV static double pin(double x) { asm volatile("" : "+v"(x)); return x; }
    CGP_DEV void init() {
        const double ex_[14] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
                                1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
        const double lp_[6] = {-1.0 / 6.0, 0.2, -0.25, 1.0 / 3.0, -0.5, 1.0};
        const double sn_[8] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0, 1.0 / 39916800.0,
                               -1.0 / 362880.0, 1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0};
        const double cs_[7] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0,
                               1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0};
        CGP_UNROLL for (int i = 0; i < 14; i++) ex[i] = pin(ex_[i]);
        CGP_UNROLL for (int i = 0; i < 6; i++) lp[i] = pin(lp_[i]);
        CGP_UNROLL for (int i = 0; i < 8; i++) sn[i] = pin(sn_[i]);
        CGP_UNROLL for (int i = 0; i < 7; i++) cs[i] = pin(cs_[i]);
        log2e = pin(kLog2e); ln2hi = pin(kLn2Hi); ln2lo = pin(kLn2Lo);
        two_over_pi = pin(kTwoOverPi); pio2_1 = pin(kPio2_1); pio2_2 = pin(kPio2_2); pio2_3 = pin(kPio2_3);
    }
};

// Same call shape with the coefficients left as literals, for kernels with no registers to spare.
struct FastMathImm {};
CGP_DEV void softplus_pair_uniform(const FastMathImm&, double x, double& sp, double& dsp) { softplus_pair_uniform(x, sp, dsp); }
CGP_DEV void fast_sincos_uniform(const FastMathImm&, double x, double& sn, double& cs) { fast_sincos_uniform(x, sn, cs); }

// softplus_pair_uniform / fast_sincos_uniform with pinned coefficients, arranged for LATENCY: these two calls are the
// head of every EKF step's dependent chain (680 of a step's 1200 cycles in the first MFMA kernel; per-op latencies on
// MI355X from tools/ubench/f64_ops.hip: dependent v_fma_f64 8 cycles, readfirstlane -> scalar compare -> branch 43,
// v_cmp -> v_cndmask 20, v_rcp_f64 19).  So:
//   * the common regime is evaluated unconditionally and the regime test is a scalar compare issued at the top and
//     consumed by a rarely-taken forward branch at the bottom (the fix-up recomputes in the reference's naive form);
//   * every polynomial is in Estrin form (3-4 dependent levels instead of 6-13);
//   * the sin / cos quadrant logic stays on the vector ALU, off the critical path, instead of a round trip through
//     the scalar unit.
// Even a never-taken branch splits a step into basic blocks that the instruction scheduler cannot interleave (measured:
// 167 cycles per step for these two), which is why the matrix-core EKF goes one step further and runs whole 64-step
// chunks with no checks at all (SpecRegs below, cgp_mfma4.hpp); the versions here serve the DPP cooperative kernels.
// exp(-x) for |x| < 700 (no overflow / underflow / NaN handling): x = -(k ln2 + r), Estrin on r, v_ldexp_f64.
template <class Regs>
CGP_DEV double exp_neg_common(const Regs& R, double x) {
    const double nx = -x;
    const double k = __builtin_rint(nx * R.log2e);
    double r = fma(-k, R.ln2hi, nx);
    r = fma(-k, R.ln2lo, r);
    // c_i = ex[13 - i]
    const double r2 = r * r;
    const double a0 = horner(R.ex[12], r, R.ex[13]), a1 = horner(R.ex[10], r, R.ex[11]), a2 = horner(R.ex[8], r, R.ex[9]);
    const double a3 = horner(R.ex[6], r, R.ex[7]), a4 = horner(R.ex[4], r, R.ex[5]), a5 = horner(R.ex[2], r, R.ex[3]);
    const double a6 = horner(R.ex[0], r, R.ex[1]);
    const double r4 = r2 * r2;
    const double b0 = horner(a1, r2, a0), b1 = horner(a3, r2, a2), b2 = horner(a5, r2, a4);
    const double r8 = r4 * r4;
    const double d0 = horner(b1, r4, b0), d1 = horner(a6, r4, b2);
    return __builtin_amdgcn_ldexp(horner(d1, r8, d0), (int)k);
}
// softplus and its derivative from t = exp(-x), 6 <= x < 700: x + log1p(t), 1 / (1 + t).
CGP_DEV void softplus_from_exp_neg(const FastMathRegs& R, double x, double t, double& sp, double& dsp) {
    // log1p(t) / t = 1 - t/2 + t^2/3 - t^3/4 + t^4/5 - t^5/6
    const double t2 = t * t;
    const double l0 = horner(R.lp[4], t, R.lp[5]), l1 = horner(R.lp[2], t, R.lp[3]), l2 = horner(R.lp[0], t, R.lp[1]);
    const double q = horner(horner(l2, t2, l1), t2, l0);
    sp = fma(q, t, x);
    dsp = rcp_nr(1.0 + t);
}
// 6.0 <= x < 700.0 as one unsigned compare on the high word of a wave-uniform x (false for NaN and negatives).
CGP_DEV bool softplus_common_regime(double x) {
    const unsigned hx = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(x));
    return (hx - 0x40180000u) < (0x4085E000u - 0x40180000u);
}

// ---- the speculative EKF step's softplus (cgp_mfma4.hpp) ------------------------------------------------------------------
// That step is one dependent chain, half of it these two polynomials, and its results sat at 1e-13 of the CPU checker against
// a gate of 1e-5 (1e-9 in the full-size test): the chain carries near-minimax polynomials of degree 7 (three Estrin levels,
// 9 instructions each) instead of degree 13 / 15 (four levels, 17 / 18 instructions):
//     exp(r),        |r| <= ln 2 / 2:        relative error 5.5e-11        (kExpLean)
//     log1p(t) / t,  0 <= t <= exp(-1.5):    relative error 1.0e-11        (kLog1pOverTLean)
// (Chebyshev-node interpolants, tools/gen_math_constants.py).  With t = exp(-x) <= 0.223 the softplus x + t q(t) is then
// good to 1.4e-11 absolute (7e-12 relative: x >= 1.5) and the rotation angle dt 2 pi fs g(x) to 1e-13 rad at the demos'
// dt = 1e-3; the derivative 1 / (1 + t) to 1.2e-11.  The error is a fixed function of x -- a softplus perturbed by 1e-11 --
// so it does not accumulate from step to step.  Outside [1.5, 700) the chunk is repeated with the checked step, which
// uses the full-precision functions (and the reference's naive form where that is what the reference evaluates).
constexpr double kExpLean[8] = {0.9999999999595294, 0.9999999999955055, 0.5000000107793876, 0.1666666678638044,
                                0.041666218139710595, 0.008333283518768105, 0.0013948590286863383, 0.00019907582591325134};
constexpr double kLog1pOverTLean[8] = {0.999999999990116, -0.49999999432118947, 0.3333327946277929, -0.24998038614728868,
                                       0.19964561452797533, -0.16312923302145096, 0.12263546195074254, -0.059855823415835646};

// Constants of the speculative step pinned in VGPRs (see FastMathRegs).  lq is stored pre-multiplied by `scale` (the
// step wants the rotation angle scale * softplus, so the scale rides in the coefficients: one multiply less on the chain).
struct SpecRegs {
    double ex[8], lq[8];
    double log2e, ln2hi, ln2lo, ln2;
    double s3, c4;               // -1/6, 1/24: the rotation increment's sin d = d (1 + s3 d^2), tan(d / 2) = d (1/2 + c4 d^2)
    // PIN = false leaves the values as ordinary constants the compiler may rematerialise: fewer live registers, a few
    // more moves -- the trade for a kernel that wants two waves per SIMD rather than the shortest chain
    template <bool PIN = true> CGP_DEV void init(double scale = 1.0) {
        auto keep = [](double v) { return PIN ? FastMathRegs::pin(v) : v; };
        CGP_UNROLL for (int i = 0; i < 8; i++) ex[i] = keep(kExpLean[i]);
        CGP_UNROLL for (int i = 0; i < 8; i++) lq[i] = keep(kLog1pOverTLean[i] * scale);
        log2e = keep(kLog2e); ln2hi = keep(kLn2Hi); ln2lo = keep(kLn2Lo); ln2 = keep(kLn2Hi + kLn2Lo);
        s3 = keep(-1.0 / 6.0); c4 = keep(1.0 / 24.0);
    }
};
// exp(-x) for |x| < 700 with the lean polynomial: x = -(k ln2 + r), three Estrin levels, v_ldexp_f64.
template <class Regs>
CGP_DEV double exp_neg_lean(const Regs& R, double x) {
    const double nx = -x;
    const double k = __builtin_rint(nx * R.log2e);
    double r = fma(-k, R.ln2hi, nx);
    r = fma(-k, R.ln2lo, r);
    const double r2 = r * r;
    const double a0 = horner(R.ex[1], r, R.ex[0]), a1 = horner(R.ex[3], r, R.ex[2]);
    const double a2 = horner(R.ex[5], r, R.ex[4]), a3 = horner(R.ex[7], r, R.ex[6]);
    const double r4 = r2 * r2;
    const double b0 = horner(a1, r2, a0), b1 = horner(a3, r2, a2);
    return __builtin_amdgcn_ldexp(horner(b1, r4, b0), (int)k);
}
// The same with a ONE-constant range reduction, r = -x - k ln2 with ln2 rounded to double: the reduction's error is
// k ulp(ln2) / 2 <= 6e-14 absolute for x < 700 (k <= 1010), 2e-15 for the x < 20 of a chirp's frequency state -- far inside
// the lean polynomial's own 1e-10.  One vector instruction less on a step that is bound by instruction issue.
template <class Regs>
CGP_DEV double exp_neg_lean1(const Regs& R, double x) {
    const double nx = -x;
    const double k = __builtin_rint(nx * R.log2e);
    const double r = fma(-k, R.ln2, nx);
    const double r2 = r * r;
    const double a0 = horner(R.ex[1], r, R.ex[0]), a1 = horner(R.ex[3], r, R.ex[2]);
    const double a2 = horner(R.ex[5], r, R.ex[4]), a3 = horner(R.ex[7], r, R.ex[6]);
    const double r4 = r2 * r2;
    const double b0 = horner(a1, r2, a0), b1 = horner(a3, r2, a2);
    return __builtin_amdgcn_ldexp(horner(b1, r4, b0), (int)k);
}
// scale * log1p(t) / t  (scale folded into R.lq) and the softplus derivative 1 / (1 + t), t = exp(-x) <= exp(-1.5).
template <class Regs>
CGP_DEV void softplus_tail_lean(const Regs& R, double t, double& q_scaled, double& dsp) {
    const double t2 = t * t;
    const double a0 = horner(R.lq[1], t, R.lq[0]), a1 = horner(R.lq[3], t, R.lq[2]);
    const double a2 = horner(R.lq[5], t, R.lq[4]), a3 = horner(R.lq[7], t, R.lq[6]);
    const double t4 = t2 * t2;
    const double b0 = horner(a1, t2, a0), b1 = horner(a3, t2, a2);
    q_scaled = horner(b1, t4, b0);
    dsp = rcp_nr1(1.0 + t);
}

// ---- the same step in its HIGH regime: frequency state x >= 5 (t = exp(-x) <= 6.74e-3; 23 Hz and up at the demos' scaling) ----
// What the step needs of the softplus there takes far shorter polynomials than on t <= 0.223 (tools/gen_math_constants.py,
// Chebyshev-node fits):
//     exp(r), |r| <= ln 2 / 2, degree 6:    relative error 2.5e-9   (x t x the angle scale: 1.1e-13 rad of rotation angle)
//     log1p(t) / t on [0, e^-5], degree 2:  relative error 2.4e-9   (1.0e-13 rad)
//     1 / (1 + t)  on [0, e^-5], degree 4:  relative error 2.7e-14  (no reciprocal, no Newton step)
// -- 7 vector operations and a v_rcp_f64 less than the lean forms above, on a step that is bound by instruction issue.  The
// degrees are set by the filter's sensitivity, measured against the C port on records that live at the regime's lower edge:
// 1 / (1 + t) to degree 3 (1.6e-11, a constant bias of the Jacobian's softplus derivative) showed as 7e-10 in the weakly
// observed frequency-rate state -- too close to the 1e-9 gate of the full-size tests -- hence degree 4; the two angle
// polynomials cost 2.6e-11 at worst where the common regime's own lean polynomials cost 1.4e-10 (degree 7 / 3 instead: 1.2e-12).
constexpr double kExpHigh[7] = {1.0, 1.0000000377388714, 0.5000000047146057, 0.16666415414041177, 0.04166635277111221,
                                0.008375134774624326, 0.0013941118895837};
constexpr double kLog1pOverTHigh[3] = {0.9999999976293401, -0.4999936650358292, 0.33082183854737557};
constexpr double kSigmoidHigh[5] = {0.9999999999999734, -0.9999999998020426, 0.999999764797864, -0.9999021072027222, 0.9833379103966127};
struct SpecRegsHigh {
    double ex[7], lq[3], sg[5];
    // `scale` rides in the coefficients of log1p(t) / t, `dscale` in those of 1 / (1 + t) (the EKF wants the derivative times a
    // per-lane factor of its Jacobian: one multiply less)
    CGP_DEV void init(double scale, double dscale = 1.0) {
        CGP_UNROLL for (int i = 0; i < 7; i++) ex[i] = FastMathRegs::pin(kExpHigh[i]);
        CGP_UNROLL for (int i = 0; i < 3; i++) lq[i] = FastMathRegs::pin(kLog1pOverTHigh[i] * scale);
        CGP_UNROLL for (int i = 0; i < 5; i++) sg[i] = FastMathRegs::pin(kSigmoidHigh[i] * dscale);
    }
};
// exp(-x) with the degree-6 polynomial (log2e and ln2 from the lean set R)
template <class Regs>
CGP_DEV double exp_neg_high(const Regs& R, const SpecRegsHigh& H, double x) {
    const double nx = -x;
    const double k = __builtin_rint(nx * R.log2e);
    const double r = fma(-k, R.ln2, nx);
    const double r2 = r * r;
    const double a0 = horner(H.ex[1], r, H.ex[0]), a1 = horner(H.ex[3], r, H.ex[2]), a2 = horner(H.ex[5], r, H.ex[4]);
    const double r4 = r2 * r2;
    const double b0 = horner(a1, r2, a0), b1 = horner(H.ex[6], r2, a2);
    return __builtin_amdgcn_ldexp(horner(b1, r4, b0), (int)k);
}
// scale * log1p(t) / t and dscale / (1 + t) for t <= exp(-5)
CGP_DEV void softplus_tail_high(const SpecRegsHigh& H, double t, double& q_scaled, double& dsp) {
    q_scaled = horner(horner(H.lq[2], t, H.lq[1]), t, H.lq[0]);
    dsp = horner(horner(horner(horner(H.sg[4], t, H.sg[3]), t, H.sg[2]), t, H.sg[1]), t, H.sg[0]);
}

// The MID regime of the speculative EKF step (round 5): for |x| <= 2, between the two lean regimes,
//     softplus(x) = x / 2 + g(x^2),   g(w) = log(2 cosh(sqrt(w) / 2));      softplus'(x) = 1 / 2 + x h(x^2),   h(w) = tanh(sqrt(w) / 2) / (2 sqrt(w))
// with g and h as degree-14 polynomials in w = x^2 (both analytic up to w = -pi^2: Chebyshev interpolation on [0, 4.0008] converges like
// 7.7^-n; relative error 2.5e-16 / 4.3e-16, the rounding of the coefficients -- tools/gen_math_constants.py).  No exp, no log, no
// reciprocal: two Estrin evaluations of five dependent levels, where the branch-free full-range form (softplus_pair_any) is a chain of ~ 33.
// `scale` rides in g's coefficients (and the caller's x / 2 term), `dscale` in h's.
constexpr double kSoftplusMidG[15] = {0.6931471805599453, 0.1249999999999985, -0.005208333333305166, 0.0003472222220131582, -2.6351685692261904e-05,
                                      2.1356903149955178e-06, -1.803200077700644e-07, 1.5657248276684575e-08, -1.385960812967277e-09,
                                      1.2369325852697208e-10, -1.0850756248916867e-11, 8.834323634734408e-13, -6.018381605146707e-14,
                                      2.903148270175286e-15, -7.119528411200814e-17};
constexpr double kSoftplusMidH[15] = {0.24999999999999997, -0.02083333333332944, 0.0020833333332603114, -0.00021081349152117946, 2.1356920280814414e-05,
                                      -2.163870857832512e-06, 2.1923833378278747e-07, -2.2205950615139193e-08, 2.2444102551603243e-09,
                                      -2.2450104536606832e-10, 2.1645992007368752e-11, -1.8954433287798848e-12, 1.3596116122481806e-13,
                                      -6.787790406146025e-15, 1.7022413849506242e-16};
struct SpecRegsMid {
    double g[15], h[15];
    CGP_DEV void init(double scale = 1.0, double dscale = 1.0) {
        CGP_UNROLL for (int i = 0; i < 15; i++) g[i] = FastMathRegs::pin(kSoftplusMidG[i] * scale);
        CGP_UNROLL for (int i = 0; i < 15; i++) h[i] = FastMathRegs::pin(kSoftplusMidH[i] * dscale);
    }
};
// sum_{i < 15} c[i] w^i, Estrin: five dependent levels behind w
CGP_DEV double estrin15(const double (&c)[15], double w) {
    const double w2 = w * w;
    const double a0 = horner(c[1], w, c[0]), a1 = horner(c[3], w, c[2]), a2 = horner(c[5], w, c[4]), a3 = horner(c[7], w, c[6]);
    const double a4 = horner(c[9], w, c[8]), a5 = horner(c[11], w, c[10]), a6 = horner(c[13], w, c[12]);
    const double w4 = w2 * w2;
    const double b0 = horner(a1, w2, a0), b1 = horner(a3, w2, a2), b2 = horner(a5, w2, a4), b3 = horner(c[14], w2, a6);
    const double w8 = w4 * w4;
    const double d0 = horner(b1, w4, b0), d1 = horner(b3, w4, b2);
    return horner(d1, w8, d0);
}
// scale * (softplus(x) - x / 2) and dscale * (softplus'(x) - 1 / 2) / x ... as the two polynomial values: the caller adds the linear terms
CGP_DEV void softplus_mid_polys(const SpecRegsMid& M, double x, double& g_scaled, double& h_scaled) {
    const double w = x * x;
    g_scaled = estrin15(M.g, w);
    h_scaled = estrin15(M.h, w);
}
// the pair itself (scale = dscale = 1): tests, cgp_debug_math
CGP_DEV void softplus_pair_mid(double x, double& sp, double& dsp) {
    SpecRegsMid M;
    M.init();
    double g, h;
    softplus_mid_polys(M, x, g, h);
    sp = fma(0.5, x, g);
    dsp = fma(x, h, 0.5);
}

// Pinned coefficients for a per-lane sigma-point fan that is evaluated without regime branches (cgp_mfma4_sigma.hpp,
// cgp_mfma4_cd.hpp): the lean softplus above (SoftplusRegs) and, for the discrete model's rotation, sin / cos on the
// reduced range |r| <= pi/4 (Taylor to r^17 / r^16) in Estrin form -- four dependent levels instead of the eight of
// sincos_reduced's Horner chains (FanRegs).
struct SoftplusRegs {
    double ex[8], lq[8], log2e, ln2hi, ln2lo;
    CGP_DEV void init() {
        CGP_UNROLL for (int i = 0; i < 8; i++) { ex[i] = FastMathRegs::pin(kExpLean[i]); lq[i] = FastMathRegs::pin(kLog1pOverTLean[i]); }
        log2e = FastMathRegs::pin(kLog2e); ln2hi = FastMathRegs::pin(kLn2Hi); ln2lo = FastMathRegs::pin(kLn2Lo);
    }
};
struct FanRegs : SoftplusRegs {
    double sn[8];       // -1/17!, 1/15!, ..., 1/3!
    double cs[7];       // 1/16!, -1/14!, ..., 1/4!
    CGP_DEV void init() {
        const double sn_[8] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0, 1.0 / 39916800.0,
                               -1.0 / 362880.0, 1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0};
        const double cs_[7] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0,
                               1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0};
        SoftplusRegs::init();
        CGP_UNROLL for (int i = 0; i < 8; i++) sn[i] = FastMathRegs::pin(sn_[i]);
        CGP_UNROLL for (int i = 0; i < 7; i++) cs[i] = FastMathRegs::pin(cs_[i]);
    }
};
CGP_DEV void sincos_reduced(const FanRegs& R, double r, double& s0, double& c0) {
    const double z = r * r, z2 = z * z;
    const double sa0 = horner(R.sn[6], z, R.sn[7]), sa1 = horner(R.sn[4], z, R.sn[5]), sa2 = horner(R.sn[2], z, R.sn[3]);
    const double sa3 = horner(R.sn[0], z, R.sn[1]);
    const double ca0 = horner(R.cs[5], z, R.cs[6]), ca1 = horner(R.cs[3], z, R.cs[4]), ca2 = horner(R.cs[1], z, R.cs[2]);
    const double z4 = z2 * z2;
    const double sb0 = horner(sa1, z2, sa0), sb1 = horner(sa3, z2, sa2);
    const double cb0 = horner(ca1, z2, ca0), cb1 = horner(R.cs[0], z2, ca2);
    const double ps = horner(sb1, z4, sb0), pc = horner(cb1, z4, cb0);
    s0 = fma(-(r * z), ps, r);
    c0 = fma(z2, pc, fma(-0.5, z, 1.0));
}

// The constants of softplus_pair_any and fast_sincos_spec pinned in registers for the duration of a chunk on the ANY / WIDE tiers of the
// matrix-core EKF (round 5): that translation unit is built without machine-level loop-invariant code motion (Makefile), so literal
// operands would be materialised again in every iteration of those tiers' loops -- 138 scalar moves per eight steps, measured.  ~ 90
// vector moves per 64-step chunk instead.  The softplus is the literal form's operation for operation; sin / cos take the fans' Estrin form.
struct WideRegs : FanRegs {
    double ex[11];      // 1/6, 1/24, ..., 1/13!  (exp Taylor from the cubic term on; 1, 1, 1/2 are inline constants)
    double lg[10];      // 1/3, 1/5, ..., 1/21
    double sqrt2, two_over_pi, pio2_1, pio2_2, pio2_3;
    CGP_DEV void init() {
        FanRegs::init();
        const double ex_[11] = {1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0,
                                1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0};
        const double lg_[10] = {1.0 / 3.0, 1.0 / 5.0, 1.0 / 7.0, 1.0 / 9.0, 1.0 / 11.0, 1.0 / 13.0, 1.0 / 15.0, 1.0 / 17.0, 1.0 / 19.0, 1.0 / 21.0};
        CGP_UNROLL for (int i = 0; i < 11; i++) ex[i] = FastMathRegs::pin(ex_[i]);
        CGP_UNROLL for (int i = 0; i < 10; i++) lg[i] = FastMathRegs::pin(lg_[i]);
        sqrt2 = FastMathRegs::pin(1.4142135623730951);
        two_over_pi = FastMathRegs::pin(kTwoOverPi);
        pio2_1 = FastMathRegs::pin(kPio2_1); pio2_2 = FastMathRegs::pin(kPio2_2); pio2_3 = FastMathRegs::pin(kPio2_3);
    }
};
// softplus_pair_any with its constants from W (the same operations in the same order)
CGP_DEV void softplus_pair_any(const WideRegs& W, double x, double& sp, double& dsp, bool& ok) {
    const double a = fabs(x);
    ok = a < 700.0;
    const double na = -a;
    const double k = __builtin_rint(na * W.log2e);
    double r = fma(-k, W.ln2hi, na);
    r = fma(-k, W.ln2lo, r);
    const double r2 = r * r;
    const double e0 = fma(r, 1.0, 1.0), e1 = fma(r, W.ex[0], 0.5), e2 = fma(r, W.ex[2], W.ex[1]), e3 = fma(r, W.ex[4], W.ex[3]);
    const double e4 = fma(r, W.ex[6], W.ex[5]), e5 = fma(r, W.ex[8], W.ex[7]), e6 = fma(r, W.ex[10], W.ex[9]);
    const double r4 = r2 * r2;
    const double f0 = fma(e1, r2, e0), f1 = fma(e3, r2, e2), f2 = fma(e5, r2, e4);
    const double r8 = r4 * r4;
    const double g0 = fma(f1, r4, f0), g1 = fma(e6, r4, f2);
    const double t = __builtin_amdgcn_ldexp(fma(g1, r8, g0), (int)k);
    const double z = 1.0 + t;
    const double rz = rcp_nr(z);
    const bool big = z > W.sqrt2;
    const double m = big ? 0.5 * z : z;
    const double s = (m - 1.0) * rcp_nr(m + 1.0);
    const double sc = fma(fma(-(m + 1.0), s, m - 1.0), rcp_nr1(m + 1.0), s);
    const double s2 = sc * sc;
    const double p0 = fma(s2, W.lg[1], W.lg[0]), p1 = fma(s2, W.lg[3], W.lg[2]), p2 = fma(s2, W.lg[5], W.lg[4]);
    const double p3 = fma(s2, W.lg[7], W.lg[6]), p4 = fma(s2, W.lg[9], W.lg[8]);
    const double s4 = s2 * s2;
    const double q0 = fma(p1, s4, p0), q1 = fma(p3, s4, p2);
    const double s8 = s4 * s4;
    const double pl = fma(fma(p4, s8, q1), s8, q0);
    const double two_s = sc + sc;
    double lz = fma(two_s, pl * s2, two_s);
    lz = big ? fma(1.0, W.ln2lo, lz) + W.ln2hi : lz;
    const double l1p = fma(t - (z - 1.0), rz, lz);
    sp = (x > 0.0 ? x : 0.0) + l1p;
    dsp = (x > 0.0 ? 1.0 : t) * rz;
}
// fast_sincos_spec with its constants from W (the reduced-range polynomials in the Estrin form of the fans: four dependent levels)
CGP_DEV void fast_sincos_spec(const WideRegs& W, double x, double& sn, double& cs, bool& ok) {
    ok = fabs(x) < 1.0e5;
    const double n = __builtin_rint(x * W.two_over_pi);
    double r = fma(-n, W.pio2_1, x);
    r = fma(-n, W.pio2_2, r);
    r = fma(-n, W.pio2_3, r);
    double s0, c0;
    sincos_reduced(static_cast<const FanRegs&>(W), r, s0, c0);
    const int q = (int)n;
    const bool swap = (q & 1) != 0;
    const double a = swap ? c0 : s0, b = swap ? s0 : c0;
    sn = __hiloint2double(__double2hiint(a) ^ ((q & 2) << 30), __double2loint(a));
    cs = __hiloint2double(__double2hiint(b) ^ (((q + 1) & 2) << 30), __double2loint(b));
}

// The wave-uniform pair with the lean polynomials (regime [1.5, 700), 7e-12 / 1.2e-11; see "the speculative EKF step's
// softplus" above): evaluated unconditionally, the regime test is a scalar compare consumed by a rarely-taken branch at the end.
template <class Regs>
CGP_DEV void softplus_pair_uniform_lean(const Regs& R, double x, double& sp, double& dsp) {
    const unsigned hx = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(x));
    const bool common = (hx - 0x3FF80000u) < (0x4085E000u - 0x3FF80000u);
    const double t = exp_neg_lean(R, x);
    double q;
    softplus_tail_lean(R, t, q, dsp);
    sp = fma(q, t, x);
    if (__builtin_expect(!common, 0)) {            // elsewhere, and for inf / NaN: the naive form of models.py:50 as is
        const double e = fast_exp(x);
        const double z = e + 1.0;
        sp = fast_log_ge1(z);
        dsp = e * rcp_nr(z);
    }
}
CGP_DEV void softplus_pair_uniform(const SpecRegs& R, double x, double& sp, double& dsp) { softplus_pair_uniform_lean(R, x, sp, dsp); }
CGP_DEV void softplus_pair_uniform(const SoftplusRegs& R, double x, double& sp, double& dsp) { softplus_pair_uniform_lean(R, x, sp, dsp); }
CGP_DEV void softplus_pair_uniform(const FastMathRegs& R, double x, double& sp, double& dsp) {
    const bool common = softplus_common_regime(x);
    softplus_from_exp_neg(R, x, exp_neg_common(R, x), sp, dsp);
    if (__builtin_expect(!common, 0)) {            // elsewhere, and for inf / NaN: the naive form of models.py:50 as is
        const double e = fast_exp(x);
        const double z = e + 1.0;
        sp = fast_log_ge1(z);
        dsp = e * rcp_nr(z);
    }
}
CGP_DEV void fast_sincos_uniform(const FastMathRegs& R, double x, double& sn, double& cs) {
    const unsigned hx = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(x)) & 0x7fffffffu;
    const bool common = hx < 0x40F86A00u;                                          // |x| < 1e5 (not inf, not NaN)
    const double n = __builtin_rint(x * R.two_over_pi);
    double r = fma(-n, R.pio2_1, x);
    r = fma(-n, R.pio2_2, r);
    r = fma(-n, R.pio2_3, r);
    // quadrant: swap mask and sign bits from the integer n, per lane (same in every lane), ready long before the
    // polynomials are
    const int qi = (int)n;
    const bool swap = (qi & 1) != 0;
    const int sa = (qi & 2) << 30, sb = ((qi + 1) & 2) << 30;
    const double z = r * r;
    const double z2 = z * z;
    // sin: r - r^3 (c0 + c1 z + ... + c7 z^7), c_i = sn[7 - i];  cos: 1 - z/2 + z^2 (c0 + ... + c6 z^6), c_i = cs[6 - i]
    const double sa0 = horner(R.sn[6], z, R.sn[7]), sa1 = horner(R.sn[4], z, R.sn[5]), sa2 = horner(R.sn[2], z, R.sn[3]);
    const double sa3 = horner(R.sn[0], z, R.sn[1]);
    const double ca0 = horner(R.cs[5], z, R.cs[6]), ca1 = horner(R.cs[3], z, R.cs[4]), ca2 = horner(R.cs[1], z, R.cs[2]);
    const double z4 = z2 * z2;
    const double sb0 = horner(sa1, z2, sa0), sb1 = horner(sa3, z2, sa2);
    const double cb0 = horner(ca1, z2, ca0), cb1 = horner(R.cs[0], z2, ca2);
    const double ps = horner(sb1, z4, sb0), pc = horner(cb1, z4, cb0);
    const double s0 = fma(-(r * z), ps, r);
    const double c0 = fma(z2, pc, fma(-0.5, z, 1.0));
    const double a = swap ? c0 : s0, b = swap ? s0 : c0;
    sn = __hiloint2double(__double2hiint(a) ^ sa, __double2loint(a));
    cs = __hiloint2double(__double2hiint(b) ^ sb, __double2loint(b));
    if (__builtin_expect(!common, 0)) sincos(x, &sn, &cs);                          // out of line: rare
}

// Negative log-likelihood increment of a scalar Gaussian measurement, in the arithmetic of
// jax.scipy.stats.norm.logpdf(y, pred, sqrt(S)) (filters_smoothers.py:44-45, 68).
// Round 4: with the engine's own square root, logarithm and reciprocal root (5e-15, 1 ulp) instead of the library's sqrt, log and
// divide -- 593 cycles of dependent latency and ~ 100 instructions per call (tools/ubench/f64_issue.hip), paid once per trial-step
// by the lane-per-trial kernels and once per 64-step chunk, on the critical path, by the wave-per-trial ones.  S <= 0 or NaN -> NaN.
CGP_DEV double nll_increment(double S, double innov) {
    double sc, isc;
    sqrt_rsqrt(S, sc, isc);                            // scale = sqrt(S), as the reference passes it to logpdf
    const double z = innov * isc;
    return 0.5 * (fast_log_ge1(kTwoPi * (sc * sc)) + z * z);
}

}  // namespace cgp
