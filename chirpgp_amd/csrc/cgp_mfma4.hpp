// cgp_mfma4.hpp -- the d = 4 EKF step of BASELINE config C2 on the float64 matrix cores.
//
// v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4 x 4 blocks per wavefront, one element of A, B and C / D per
// lane.  Lane layout on gfx950, found with tools/ubench/mfma_f64_layout.hip:
//
//     lane = 16 r + 4 b + q     b = block;   A holds A_b[q][r],   B holds B_b[r][q],   C / D hold D_b[r][q]
//
// so a register that carries X[r][q] ("natural" layout) is X as the B operand or the result, and X^T as the A operand.
// The covariance therefore lives at lane (r, q) = (lane >> 4, lane & 3) -- matrix row = DPP row, column = lane of the
// quad -- replicated over the four blocks, and one register RJT holding J[q][r] serves twice:
//
//     Q  = P J^T            mfma(A = P (symmetric), B = RJT)
//     Pp = J Q + Sigma      mfma(A = RJT,           B = Q,   C = Sigma)
//     PH[r] (per row)       mfma(A = Pp,            B = H[r])                the measurement vector is a per-lane constant
//     PH[q] (per column)    mfma(A = H[r],          B = Pp)
//     S = H . PH + Xi       mfma(A = H[r],          B = PH[r], C = Xi)       lands in every lane
//
// Five matrix instructions replace the 85 DPP moves and FMAs of the cooperative kernel's covariance algebra
// (cgp_coop4.hpp), three more carry the mean (ekf4_mfma_finish); what stays on the vector ALU is the scalar chain
// softplus -> rotation and the rank-one update.  This is not a GEMM-shaped workload being forced onto MFMA: the products ARE 4 x 4 x 4, and the instruction
// is used for its latency (one issue slot, 4 passes) on a T-serial chain.
//
// Speculation.  The step is one dependent chain, and a branch anywhere in it costs far more than its own cycles: it
// cuts the step into basic blocks that cannot be interleaved (tools/ekf_variants.py: 167 cycles of a 1070-cycle step
// for the two never-taken regime checks of softplus and sincos).  So a chunk of 64 steps first runs with NO branches --
// the common-regime formulas evaluated blindly (ekf4_mfma_step_spec), the verdicts ORed into a scalar -- and only if
// some step left the regime (frequency state outside its band, a jump of the rotation angle beyond the increment bound,
// inf, NaN) is the chunk repeated from its saved state on the next tier.  Since round 5 the regime of a chunk is chosen from
// the state at its start -- HIGH (u >= 5), COMMON (u >= 1.5), LOW (u <= -1.5), MID (|u| < 2), all branch-free with short
// polynomials (Ekf4Verdict below) -- and the tiers behind them are ANY (the same step with the full-accuracy softplus for
// any |u| < 700), WIDE (a fresh sincos per step: a jump of the angle) and the CHECKED step, which evaluates the full
// functions with their regime branches and reproduces the reference's naive arithmetic (overflow beyond 700, NaN).  After
// a chunk left the COMMON regime the kernel stays off it for kCheckedChunks chunks (chunks that start in the LOW or the
// MID band are tried there all the same), so a record that hovers around its end pays at most 1/16 extra.
#pragma once
#include "cgp_coop4.hpp"

namespace cgp {


CGP_DEV double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

constexpr int kCheckedChunks = 16;

struct Ekf4MfmaConst {
    double M0, M1, M2, M3, rho, ang;           // M32 block, exp(-lam dt), dt 2 pi fs
    double Hr, Xi, Sig;
    double kc, ks, kj, kk;                     // J[q][r] = kc c + ks s + kk + kj dth f[q ^ 1]  (kj = -1 at (0, 2), +1 at (1, 2))
    double SigHq, c0;                          // H = e_1 form of the update: Sigma[1][q], Sigma[1][1] + Xi
    double kcr, ksr, kja;                      // kc rho, ks rho, kj ang: the damping and the angle scale ride in the per-lane coefficients
    double angm;                               // ang in the lanes of the rotation block, 0 elsewhere (see Ekf4Anchor)
    CGP_DEV void fold() { kcr = kc * rho; ksr = ks * rho; kja = kj * ang; angm = (kc != 0.0 || ks != 0.0) ? ang : 0.0; }
};
// The mean is distributed like the covariance: ur = u[r] (row layout) and uq = u[q] (column layout) at lane (r, q); the
// frequency state u[2] every lane needs is one quad broadcast of uq.
struct Ekf4State {
    double P, ur, uq;
    CGP_DEV double u2() const { return dpp_f64<kQuadBcast2>(uq); }
    CGP_DEV double u2_replicated() const { return row_bcast_f64<2>(uq); }      // one trial per wavefront: the four blocks are replicas
};

// Everything of a step after the rotation (c1, s1) = (cos, sin)(theta) and the softplus derivative dsp are known -- the
// mean on the matrix cores as well.  With J0 = blockdiag(rho Rot(theta), M32) held as J0[q][r] (one register, like RJT):
//     f = J0 u        by row:    mfma(A = J0,   B = u[r])        by column:  mfma(A = u[r], B = J0)
//     H . f           everywhere: mfma(A = H[r], B = f by row)
//     mf = f + PH g   one FMA per layout with PH by row / by column (which the covariance update needs anyway)
// so the per-lane scalar copies of u0..u3 and f0..f3, the quad broadcasts of PH and the FMA chains for f and H . f are
// gone; the Jacobian column d f / d u2 = dth (-f1, f0) is a quad swap of f by column times a per-lane sign.
//
// E1 != 0 (round 3): the measurement vector is the unit vector e_1 -- every chirp / La Scala model of the reference has
// H = [0, 1, 0, 0] (models.py:118).  Then H picks entries of Q = P J^T and Pp = J Q + Sigma instead of contracting with them:
//     P a = P J^T e_1 = column 1 of Q                  one quad broadcast of Q            (a = J^T H = row 1 of J)
//     (Pp H)[r] = Pp[r][1]  by row                      one quad broadcast of Pp
//     (H Pp)[q] = Pp[1][q]  by column                   mfma(A = P a, B = RJT, C = Sigma[1][q]) = (J P a)[q] + Sigma[1][q]: a lane
//                                                        cannot fetch another ROW cheaply; issued beside Pp, not behind it
//     S = Pp[1][1] + Xi                                 E1 == 2 (one trial per wavefront): entry 1 of (H Pp) by column, one row broadcast;
//                                                        E1 == 1 (one trial per MFMA block, the x4 kernel): mfma(A = a, B = P a, C = Sigma_11 + Xi)
//     H f = f[1]                                        one quad broadcast of f by column
// -- FIVE matrix instructions a step (f by row, f by column, Q, Pp, H Pp) where the general form has eight.  What this kernel
// is bound by was measured in round 3's last microbenchmarks: INSTRUCTION ISSUE first, its dependent chain second.  A wavefront
// that has its SIMD to itself issues nothing in the shadow of its own v_mfma_f64_4x4x4 (tools/ubench/mfma_valu_overlap.hip:
// M matrix + N vector instructions take 20 M + 5 N cycles, not the maximum of the two); a v_fma_f64 / v_mul_f64 / v_add_f64
// occupies it for 5 cycles, a v_mov_dpp 4, a v_readlane 12, v_rcp_f64 16 (tools/ubench/issue_costs.hip); the step's
// operations as straight-line code take about the sum of these costs in any reasonable order (tools/ubench/ekf4_parts.hip:
// 477 cycles in source order, 508 in a latency-driven list schedule; the dependent chain alone is ~ 310).  So a quad
// broadcast (two v_mov_dpp, 8 cycles) beats the matrix instruction (20) that would put the same numbers into the same
// lanes.  The chain still counts in the step's tail: H Pp as mfma(A = H, B = Pp) + S as its quad broadcast (no P a, no
// v_readlane) is three instructions shorter but one matrix instruction deeper, and measured 2.64 against 2.60 ms -- the
// linear filter, whose step is the tail alone, takes that form and gains 4 % (kf4_mfma_trial).
template <int E1 = 0>
// kjd = (kj ang) x the softplus derivative: this lane's factor of the Jacobian column d f / d u2
CGP_DEV void ekf4_mfma_finish_j(const Ekf4MfmaConst& K, double y, double J0T, double kjd, Ekf4State& x, double& S, double& innov) {
    const double f_r = mfma4(J0T, x.ur, 0.0), f_q = mfma4(x.ur, J0T, 0.0);
    const double RJT = fma(kjd, dpp_f64<kQuadSwap1>(f_q), J0T);
    double Pp, PHr, PHq;
    if constexpr (E1 != 0) {
        // E1 == 2: one trial per wavefront, the four blocks are replicas -- a 64-bit row broadcast (one instruction) does
        // what the quad broadcast (two) does when every block carries its own trial
        auto bcast1 = [](double v) { if constexpr (E1 == 2) return row_bcast_f64<1>(v); else return dpp_f64<kQuadBcast1>(v); };
        const double Q = mfma4(x.P, RJT, 0.0);
        const double Pa = bcast1(Q);                                   // (P a)[r] = Q[r][1], a = J^T H = row 1 of J
        PHq = mfma4(Pa, RJT, K.SigHq);                                 // (J P a)[q] + Sigma[1][q] = Pp[1][q], beside Pp
        Pp = mfma4(RJT, Q, K.Sig);
        PHr = bcast1(Pp);                                              // Pp[r][1]
        if constexpr (E1 == 1) {
            const double a = dpp_f64<kQuadBcast1>(RJT);                // a[r] = J[1][r]
            S = mfma4(a, Pa, K.c0);                                    // a . (P a) + Sigma_11 + Xi, per MFMA block
        } else {
            // Pp[1][1] is entry 1 of (H Pp) by column: ONE row broadcast (round 4; a v_readlane pair of lane (1, 1) of Pp
            // before: 16 issue cycles against 5 -- 2.19 -> 2.16 ms)
            S = bcast1(PHq) + K.Xi;
        }
        innov = y - bcast1(f_q);                                       // H . f = f[1]
    } else {
        // ---- predict: Pp = J P J^T + Sigma
        const double Q = mfma4(x.P, RJT, 0.0);
        Pp = mfma4(RJT, Q, K.Sig);
        // ---- update (filters_smoothers.py:55-68)
        PHr = mfma4(Pp, K.Hr, 0.0);
        PHq = mfma4(K.Hr, Pp, 0.0);
        S = mfma4(K.Hr, PHr, K.Xi);
        innov = y - mfma4(K.Hr, f_r, 0.0);
    }
    const double rS = rcp_nr1(S);                               // 2e-15 (one Newton step): two FMAs less on the chain
    // (v_fmac_f64 is VOP2 and takes its first source through the 64-bit DPP: "+= rowbcast1(Pp) * x" as one instruction would
    // spare the move that forms Pp H by row.  The compiler does not form it; as inline assembly it measured 2.48 against 2.40 ms.)
    x.P = fma(-(PHr * rS), PHq, Pp);                            // Pf = Pp - K (Pp H)^T
    const double g = rS * innov;
    x.ur = fma(PHr, g, f_r);                                    // mf = mp + K innov, in both layouts
    x.uq = fma(PHq, g, f_q);
}
template <int E1 = 0>
CGP_DEV void ekf4_mfma_finish(const Ekf4MfmaConst& K, double y, double c1, double s1, double dsp, Ekf4State& x, double& S, double& innov) {
    const double J0T = fma(K.kcr, c1, fma(K.ksr, s1, K.kk));           // rho (kc cos + ks sin) + kk: rho rides in kcr, ksr
    ekf4_mfma_finish_j<E1>(K, y, J0T, K.kja * dsp, x, S, innov);
}

// The WIDE step (round 5): full-accuracy softplus and sincos WITHOUT regime branches, valid for any |u2| < 700 and |theta| < 1e5
// (cgp_fastmath.hpp: softplus_pair_any, fast_sincos_spec) -- what a chunk outside the lean regime runs on; `bad` collects the steps outside
// even that (the chunk is then repeated with the checked step).  Records whose frequency state wanders below 1.5 -- low signal-to-noise,
// low or high chirp rates: two thirds of bench.py's C2_spread combinations -- ran 3 x slower on the checked step before.
template <int E1 = 0>
CGP_DEV void ekf4_mfma_step_wide(const Ekf4MfmaConst& K, const WideRegs& W, double y, Ekf4State& x, double& S, double& innov, bool& bad) {
    double sp, dsp, s1, c1;
    bool ok1, ok2;
    softplus_pair_any(W, x.u2(), sp, dsp, ok1);
    fast_sincos_spec(W, K.ang * sp, s1, c1, ok2);
    bad = bad || !(ok1 && ok2);
    ekf4_mfma_finish<E1>(K, y, c1, s1, dsp, x, S, innov);
}

// The checked step: full softplus and sincos, regime branches and all (the reference's naive arithmetic anywhere).
template <int E1 = 0>
CGP_DEV void ekf4_mfma_step_checked(const Ekf4MfmaConst& K, double y, Ekf4State& x, double& S, double& innov) {
    double sp, dsp, s1, c1;
    softplus_pair_uniform(x.u2(), sp, dsp);
    // theta = dt 2 pi g(u2) fs as ONE multiply by the constant dt 2 pi fs (the reference rounds three times,
    // models.py:296-297: a relative 1e-16 on an angle of ~0.05 rad)
    fast_sincos_uniform(K.ang * sp, s1, c1);
    ekf4_mfma_finish<E1>(K, y, c1, s1, dsp, x, S, innov);
}

// The speculative step.  Softplus: t = exp(-u2), then theta = ang (u2 + t q(t)) with the lean degree-7 polynomials of
// cgp_fastmath.hpp (ang rides in q's coefficients), valid for u2 >= 1.5.  Rotation: (cos, sin)(theta) is advanced
// INCREMENTALLY from the previous step's,
//     (cos, sin)(theta) = rotation of (cos, sin)(theta_prev) by d = theta - theta_prev,
// as three shears with tan(d / 2), sin d to d^3 (remainders < 5.8e-14 d while |d| <= 1.5 x 2^-8): 6 dependent operations instead of the 13 of a
// fresh sincos.  d is formed as ang q t + (ang u2 - theta_prev), whose second term does not wait for the polynomials; what
// accumulates is one rounding per step in the rotation, and the pair is re-anchored with the full sincos at the start
// of every 64-step chunk (relative error <= 64 x 2e-16).  A step with u2 outside [1.5, 700) or |d| >= 1.5 x 2^-8 (or NaN) marks
// the chunk's verdict (Ekf4Verdict) and the whole chunk is repeated with the checked step.
//
// What is rotated is not (cos, sin) but this lane's entry of J0 itself and its quarter-turn partner,
//     A = rho (kc cos + ks sin) + kk = J0[q][r],     B = rho (ks cos - kc sin):     A' = cd A + sd B,   B' = cd B - sd A
// gives the next J0 entry directly, where rotating (cos, sin) and then forming the entry took two operations more.  In the
// lanes outside the rotation block (A = kk or 0, B = 0) the increment d is ZERO, hence sd = 0, cd = 1 and A' = A: the angle
// scale rides in per-lane coefficients (K.angm and the polynomial of SpecRegs::init(K.angm)) that vanish there, at no cost.
// (The verdict on |d| is taken over the wavefront, so the lanes of the rotation block speak for it.)
struct Ekf4Anchor {
    double th, A, B;
    CGP_DEV void set(const Ekf4MfmaConst& K, double sp, double c1, double s1) {      // sp = softplus(u2): theta = ang sp
        th = K.angm * sp;
        A = fma(K.kcr, c1, fma(K.ksr, s1, K.kk));
        B = fma(K.ksr, c1, -(K.kcr * s1));
    }
};

CGP_DEV void ekf4_anchor(const Ekf4MfmaConst& K, double u2, Ekf4Anchor& a) {
    double sp, dsp, s1, c1;
    softplus_pair_uniform(u2, sp, dsp);
    fast_sincos_uniform(K.ang * sp, s1, c1);
    a.set(K, sp, c1, s1);
}

// Round 3 made the step five vector instructions shorter:
//   * the increment's sine and cosine to d^3 / d^4 only (round 2: d^5 / d^6): while |d| <= 2^-7 the dropped terms are below d^5 / 120 = 2.4e-13
//     relative and d^6 / 720 = 3e-16 (the bench records' largest increment, 2.7e-3, gives 1.2e-15), and the pair is re-anchored
//     every 64 steps;
//   * the regime verdicts from the high words of u2 and d alone (integer range checks; the chunk's verdict is one ballot pair).
// The regime verdicts of a chunk as two running maxima in vector registers (one integer add / and + one v_max_u32 each, which the
// compiler pairs into v_max3_u32 across steps; round 3 first had them as v_cmp + s_or_b64 pairs: five instructions a step):
// u = the largest offset of u2's high word from that of 1.5 (NaN, inf, negative and out-of-range values land above the
// offset of 700 as unsigned numbers), d = the largest high word of |d| (NaN above everything).
// Round 4: four running extremes of RAW high words (v_max3_u32 / v_min3_u32 / v_max3_i32 pair two steps each): no per-step offset or mask
//   umax / umin   unsigned extremes of u2's high word: a negative, NaN or infinite u2 lands above 700's word
//   d, di         unsigned and signed maximum of d's high word: the unsigned one is ruled by the negative increments (sign bit),
//                 the signed one by the positive ones; masked with 0x7FFFFFFF each is the magnitude of the largest of its sign
// The increment bound: 1.5 x 2^-8 = 5.9e-3 (the high word resolves it).  The kernel's time is that of its SLOWEST wavefront, and a
// chunk that fails a verdict is repeated with the checked step: with 2^-8 eleven of the bench's 1000 records (largest increment
// 4.8e-3) repeated a chunk -- and, then still sticky, ran sixteen more on the checked step: 2.10 -> 2.33 ms for the whole launch.
constexpr unsigned kIncrementBound = 0x3F780000u;
// The regimes of the speculative step: COMMON u2 in [1.5, 700), HIGH u2 in [5, 700) (shorter polynomials), and -- round 5 -- LOW, u2 in
// (-700, -1.5]: records the filter has lost (a chirp it never locks on, heavy noise) sit at a NEGATIVE frequency state most of the time
// (tools/state_histogram.py: 44 - 78 % of the steps of C2_spread's slow record sets below -1.5, 6 - 25 % within +-1.5), where
//     softplus(u2) = log1p(t),  t = exp(u2) <= 0.22:  theta = ang t q(t),  d softplus / d u2 = t / (1 + t)
// are the SAME lean polynomials in the same t-range as the common regime's (there t = exp(-u2), theta = ang (u2 + t q(t)), 1 / (1 + t)).
// MID, |u2| <= 2: softplus and its derivative as two polynomials in u2^2 (cgp_fastmath.hpp: SpecRegsMid) -- no exp, no log, no reciprocal.
// ANY: the same step with the full-accuracy branch-free softplus for any |u2| < 700 (softplus_pair_any): what a chunk that crosses from one
// band into the next runs on.
constexpr int kRegCommon = 0, kRegHigh = 1, kRegLow = 2, kRegAny = 3, kRegMid = 4;
struct Ekf4Verdict {
    unsigned u = 0u, umin = 0xFFFFFFFFu, d = 0u;
    int di = 0;
    bool any_bad = false;                                                   // kRegAny: a step at |u2| >= 700 (or NaN)
    unsigned uabs = 0u;                                                     // kRegMid: the largest high word of |u2| (NaN above everything)
    template <int REG = kRegCommon> CGP_DEV bool state_in_regime() const {
        if constexpr (REG == kRegAny) return !any_bad;
        if constexpr (REG == kRegMid) return uabs < 0x40000000u;             // |u2| < 2
        // (raw high words as unsigned numbers: the negative doubles order above the positive ones and by magnitude among themselves)
        if constexpr (REG == kRegLow) return umin >= 0xBFF80000u && u <= 0xC085DFFFu;           // -1.5 >= u2 > -700 (NaN words lie outside)
        constexpr unsigned lo = REG == kRegHigh ? 0x40140000u : 0x3FF80000u;
        return u <= 0x4085DFFFu && umin >= lo;
    }
    // 1: the frequency state left the regime, 2: an increment beyond the bound (bits of a wave-uniform code; 0 = the chunk stands)
    template <int REG = kRegCommon> CGP_DEV unsigned code() const {
        const bool state = __builtin_amdgcn_ballot_w64(!state_in_regime<REG>()) != 0;
        const bool jump = __builtin_amdgcn_ballot_w64((d & 0x7FFFFFFFu) >= kIncrementBound || ((unsigned)di & 0x7FFFFFFFu) >= kIncrementBound) != 0;
        return (state ? 1u : 0u) | (jump ? 2u : 0u);
    }
    template <int REG = kRegCommon> CGP_DEV unsigned long long uncommon() const {
        return __builtin_amdgcn_ballot_w64(!state_in_regime<REG>()) |                                                      // u2 outside [lo, 700)
               __builtin_amdgcn_ballot_w64((d & 0x7FFFFFFFu) >= kIncrementBound || ((unsigned)di & 0x7FFFFFFFu) >= kIncrementBound);   // |d| >= 1.5 2^-8
    }
};
template <int E1, int REG = kRegCommon>
CGP_DEV void ekf4_mfma_step_spec1(const Ekf4MfmaConst& K, const SpecRegs& R, const SpecRegsHigh& RH, double y, Ekf4State& x, Ekf4Anchor& a,
                                  double& S, double& innov, Ekf4Verdict& verdict, const SpecRegsMid* RM = nullptr, const WideRegs* RW = nullptr) {
    const double u2 = (E1 == 2) ? x.u2_replicated() : x.u2();
    constexpr bool HIGH = REG == kRegHigh, LOW = REG == kRegLow, ANY = REG == kRegAny, MID = REG == kRegMid;
    double d, jfac;                                                                  // the angle's increment; (kj ang) x the softplus derivative
    if constexpr (MID) {
        double ga, hj;                                                               // ang g(u2^2), kja h(u2^2): the scales ride in the coefficients
        softplus_mid_polys(*RM, u2, ga, hj);
        d = ga + fma(0.5 * K.angm, u2, -a.th);                                       // (the second term does not wait for the polynomial)
        jfac = fma(u2, hj, 0.5 * K.kja);
    } else if constexpr (ANY) {
        double sp, dspf; bool ok;
        if (RW) softplus_pair_any(*RW, u2, sp, dspf, ok);                              // (constants pinned for the chunk: WideRegs)
        else softplus_pair_any(u2, sp, dspf, ok);
        verdict.any_bad = verdict.any_bad || !ok;
        d = fma(K.angm, sp, -a.th);
        jfac = K.kja * dspf;
    } else {
        const double t = HIGH ? exp_neg_high(R, RH, u2) : exp_neg_lean1(R, LOW ? -u2 : u2);     // LOW: t = exp(u2)
        const double lin = LOW ? -a.th : fma(K.angm, u2, -a.th);                     // off the chain: needs u2 only (LOW: theta = ang t q(t))
        double qa, dsp;
        if constexpr (HIGH) softplus_tail_high(RH, t, qa, dsp);
        else softplus_tail_lean(R, t, qa, dsp);                                      // qa = ang log1p(t) / t
        d = fma(qa, t, lin);
        jfac = HIGH ? dsp : LOW ? (K.kja * t) * dsp : K.kja * dsp;                    // (HIGH: K.kja rides in the polynomial, SpecRegsHigh::init; LOW: t / (1 + t))
    }
    // The rotation by d as three shears (round 4) -- exact for tau = tan(d / 2), s = sin d, and of determinant 1 for ANY tau, s:
    //     A1 = A + tau B,   B' = B - s A1,   A' = A1 + tau B',        tau = d (1/2 + d^2 / 24),   s = d (1 - d^2 / 6)
    // eight operations where cos d, sin d and the four products of the plain rotation took ten (2.19 -> 2.15 ms).  Dropped terms:
    // d^5 / 240 in tau, d^5 / 120 in s -- below 5.8e-14 relative while |d| <= 1.5 x 2^-8, the verdict's bound (2^-7 in round 3:
    // 2.4e-13 a step; the bench records' largest increment is 4.8e-3, whose dropped terms are 2e-14).
    const double d2 = d * d;
    const double tau = d * fma(d2, R.c4, 0.5);
    const double sn = d * fma(d2, R.s3, 1.0);
    const double A1 = fma(tau, a.B, a.A);
    const double B = fma(-sn, A1, a.B);
    const double A = fma(tau, B, A1);
    const unsigned hx = (unsigned)__double2hiint(u2), hd = (unsigned)__double2hiint(d);
    if constexpr (MID) {
        const unsigned ha = hx & 0x7FFFFFFFu;
        verdict.uabs = verdict.uabs > ha ? verdict.uabs : ha;
    } else if constexpr (!ANY) {
        verdict.u = verdict.u > hx ? verdict.u : hx;
        verdict.umin = verdict.umin < hx ? verdict.umin : hx;
    }
    verdict.d = verdict.d > hd ? verdict.d : hd;
    verdict.di = verdict.di > (int)hd ? verdict.di : (int)hd;
    a.th += d; a.A = A; a.B = B;
    ekf4_mfma_finish_j<E1>(K, y, A, jfac, x, S, innov);
}

// Tried with it and dropped (all measured on the bench configuration, same box, A/B): a third, FLAT regime for chunks that start
// at u2 >= 38 (exp(-u2) < 2^-54: softplus is the identity to the last bit -- no exp, no polynomial; 2.38 -> 2.22 ms on a 30 - 75 Hz
// sweep, but the third unrolled variant cost the bench records, which never get there, 2.5 %: 2.16 -> 2.21 ms), the max-ILP
// scheduling strategy for this
// kernel once it is unrolled (2.72 against 2.65 ms; the other matrix-core kernels keep it; iterative-ilp / -minreg / -maxocc and
// no post-RA scheduler: 2.64 - 2.73 against 2.59), unrolling by 2 / 3 / 8 / 16 (2.70 / 2.70 / 2.65 / 2.68 ms), the four steps as
// one block in the order of a latency-driven list scheduler, pinned with scheduling barriers (tools/sched/ekf4_sched.py: 2.69
// against 2.60), both lean polynomials in Horner form (four operations fewer, eight levels deeper: no change), the measurement
// as ONE broadcast ds_read per step issued a step ahead (2.69 against 2.65; two ds_read_b128 per FOUR steps, a group ahead, are
// what the kernel uses now: 2.58 -> 2.50).  Two chain-shortening variants that ADD instructions measured slower (2.57 ms then):
// the rotation entries of the Jacobian as a quartic in the angle increment with per-lane coefficients from the previous
// rotation pair (four dependent operations fewer, nine instructions more: 2.77 ms), and the exponent argument -log2(e) u[2]
// formed beside the mean update instead of behind its quad broadcast (two fewer, six more: 2.73 ms).  All of it fits one
// picture: the step costs about the sum of its instructions' issue costs, and the order hardly matters.
constexpr int kEkf4Unroll = 4;

#ifdef CGP_EKF4_KERNELS      // the kernels are instantiated by cgp_inst_ekf4.hip alone; other units take the step functions
template <bool E1>
CGP_DEV void ekf4_mfma_trial(const FilterIO& io, const ModelArgs& ma) {
    const int lane = threadIdx.x;
    const int r = lane >> 4, q = lane & 3;
    const FilterSpan span = filter_span(io, blockIdx.x);                 // (a time-split launch: one SEGMENT of the trial's record)
    const int64_t trial = span.trial;

    HarmonicLCD<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    Ekf4MfmaConst K;
    K.M0 = model.M[0]; K.M1 = model.M[1]; K.M2 = model.M[2]; K.M3 = model.M[3];
    K.rho = model.rho;
    K.ang = (model.dt * kTwoPi) * model.fs;
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    K.Hr = Hp[r];
    K.Xi = io.Xi[trial * io.Xi_stride];
    // Sigma[r][q] (models.py:302-308)
    K.Sig = 0.0;
    if (r == q) K.Sig = (r < 2) ? model.q : (r == 2 ? model.MS[0] : model.MS[2]);
    else if (r + q == 5) K.Sig = model.MS[1];
    // J = [c -s jv0 0; s c jv1 0; 0 0 M0 M1; 0 0 M2 M3], this lane holds J[q][r]
    K.SigHq = (q == 1) ? model.q : 0.0;                                 // Sigma[1][q]: the chirp block of Sigma is q I
    K.c0 = model.q + K.Xi;
    K.kc = ((q == 0 && r == 0) || (q == 1 && r == 1)) ? 1.0 : 0.0;
    K.ks = (q == 0 && r == 1) ? -1.0 : ((q == 1 && r == 0) ? 1.0 : 0.0);
    K.kj = (r == 2 && q == 0) ? -1.0 : ((r == 2 && q == 1) ? 1.0 : 0.0);
    K.kk = (q == 2) ? (r == 2 ? K.M0 : (r == 3 ? K.M1 : 0.0)) : ((q == 3) ? (r == 2 ? K.M2 : (r == 3 ? K.M3 : 0.0)) : 0.0);
    K.fold();

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    Ekf4State x;
    x.ur = m0p[r]; x.uq = m0p[q];
    x.P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, r, q);

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    OobWindow mfs, Pfs, mnull, Pnull;
    mfs.init(io.mfs ? io.mfs + trial * T * 4 : nullptr, T * 32);
    Pfs.init(io.Pfs ? io.Pfs + trial * T * 16 : nullptr, T * 128);
    mnull.init(nullptr, 0); Pnull.init(nullptr, 0);                      // the burn-in chunks of a time-split segment store through these
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;
    // Output rows leave through buffer windows: the per-lane byte offset is a CONSTANT (block 0 stores the 16 entries of Pf,
    // lanes 0..3 the mean; every other lane carries an out-of-range offset and is dropped by the hardware), the step's
    // row offset t * 128 / t * 32 rides in the instruction's scalar offset -- no vector instruction per step for addressing.
    const unsigned p_off = (((lane >> 2) & 3) == 0) ? 8u * (4 * r + q) : kOobOffset;
    const unsigned m_off = (lane < 4) ? 8u * lane : kOobOffset;

    SpecRegs R;
    R.init(K.angm);
    SpecRegsHigh RH;
    RH.init(K.angm, K.kja);
    // (S, innovation) of each step are parked in LDS -- every lane writes the same pair to the step's slot, a plain
    // fire-and-forget ds_write -- and picked up per lane at the 64-step NLL flush (no compare / select on the chain)
    constexpr int kParkStride = 2;                                          // 32-byte slots (measured against 16: 3.35 against 3.39 ms a pass)
    __shared__ double2 park[64 * kParkStride];
    __shared__ double ybuf[64 + 16];                                        // the chunk's measurements (+ the read-ahead of the last group)
    double cum = 0.0;
    // sticky counts: off the common regime / off every speculative regime.  (The score starts below zero: two failed passes at a record's
    // start, before any chunk was kept, stop the trying at once -- the CRLB jobs' records are eight chunks long.)
    int checked_left = 0, spec_off = 0, jump_streak = 0, spec_score = -4;
    unsigned n_high = 0, n_common = 0, n_redo = 0, n_checked = 0, n_high_left = 0, n_wide = 0, n_low = 0, n_mid = 0;      // chunks by regime (scalars; cgp_debug_counters)
    // The rotation pair is re-anchored with the full softplus and sincos (a dependent chain of ~ 70 operations) every FOURTH
    // accepted chunk only (round 4): an accepted chunk hands its last (theta, A, B) to the next one -- one rounding per step in the
    // rotation, 256 steps at most: 3e-14.
    Ekf4Anchor anchor_live;
    int anchor_age = -1;                                                    // < 0: no anchor carried over
    // 64 measurements with one coalesced 512-B load, requested ONE CHUNK AHEAD: the wait for a load issued at the chunk's own
    // start exposes the whole memory latency (and, vmcnt counting in order, the drain of every store still in flight) once per
    // 64 steps -- 2.7 us of a 16 us chunk
    const int64_t Te = span.t_end;
    double ynext = (span.t_begin + lane < Te) ? ys[span.t_begin + lane] : 0.0;
    for (int64_t t0 = span.t_begin; t0 < Te; t0 += 64) {
        wave_lds_fence();                                                   // the previous chunk's NLL flush has read its slots
        // a segment's burn-in chunks write nothing (whole chunks: t_out is a multiple of 64); at the junction the state goes on record
        // (which WINDOW a chunk stores through is a scalar choice of the buffer descriptor: an empty window drops every store, and the
        // per-lane offsets stay the loop-invariant constants they were -- as per-chunk offsets they cost two vector adds a step)
        const bool burn = t0 < span.t_out;
        const OobWindow Pw = burn ? Pnull : Pfs, mw = burn ? mnull : mfs;
        if (span.state && span.seg > 0 && t0 == span.t_out) {
            if (lane < 4) span.state[lane] = x.uq;
            if (((lane >> 2) & 3) == 0) span.state[4 + 4 * r + q] = x.P;
        }
        // The empty asm consumes the loaded register here, so the compiler's s_waitcnt for it sits in this outer loop and
        // not in front of every step's v_readlane
        double ychunk = ynext;
        asm volatile("" : "+v"(ychunk));
        ynext = (t0 + 64 + lane < Te) ? ys[t0 + 64 + lane] : 0.0;
        const int nsteps = (Te - t0 < 64) ? (int)(Te - t0) : 64;
        const Ekf4State x0 = x;
        const unsigned hx0 = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(x.u2()));
        const bool low = !nll_final && hx0 - 0xBFFC0000u <= 0xC085DFFFu - 0xBFFC0000u;      // -1.75 >= u2 > -700 at the chunk's start
        const bool mid = !nll_final && (hx0 & 0x7FFFFFFFu) < 0x3FF80000u;                   // |u2| < 1.5: the MID regime's chunk (it holds while |u2| < 2)
        // 1.5 <= |u2| < 1.75: within a quarter of a band's end (with 2: the records that sit at -2 -- the 20 Hz chirp never locked on --
        // lose the LOW regime, 2.42 -> 2.68 ms; the noisy ones gain 2 %)
        const bool edge = !nll_final && !mid && (hx0 & 0x7FFFFFFFu) < 0x3FFC0000u;
        // The chunk's measurements go to LDS once, and every group of steps reads its own with broadcast ds_read_b128 one group ahead
        // -- a v_readlane pair per step costs 24 issue cycles (tools/ubench/issue_costs.hip)
        ybuf[lane] = ychunk;
        wave_lds_fence();
        // the rotation pair at the chunk's first state: carried over from the last accepted chunk, or re-anchored; every attempt of
        // this chunk starts from it
        Ekf4Anchor anchor0, anchor_end;
        // (not for a chunk that goes straight to the wide step: the anchor is a dependent chain of ~ 70 operations)
        const bool spec_allowed = spec_off == 0;
        if (!spec_allowed) anchor0 = anchor_live;
        else if (anchor_age < 0 || anchor_age >= 4) { ekf4_anchor(K, x.u2(), anchor0); anchor_age = 0; }
        else anchor0 = anchor_live;
        // One speculative pass over the chunk in regime REG: kRegCommon, kRegHigh (the short polynomials of u2 >= 5: SpecRegsHigh),
        // kRegLow, kRegAny.
        auto chunk = [&](auto reg_c) {
            constexpr int REG = decltype(reg_c)::value;
            // (the MID regime's 30 coefficients are pinned for the duration of ITS chunks only -- sixty moves and thirty products a chunk:
            // held for the whole kernel they cost the HIGH chunks of the bench records 3.5 %, measured)
            SpecRegsMid RM;
            if constexpr (REG == kRegMid) RM.init(K.angm, K.kja);
            WideRegs RW;
            if constexpr (REG == kRegAny) RW.init();
            Ekf4Anchor anchor = anchor0;
            Ekf4Verdict verdict;
            // step `k` of the group that starts at `slot`: the group's row offset rides in the stores' scalar offset, k in
            // a vector offset of its own (hoisted out of the loop) -- no scalar add per step
            auto one = [&](int slot, unsigned k, double y) {
                double S, innov;
                ekf4_mfma_step_spec1<E1 ? 2 : 0, REG>(K, R, RH, y, x, anchor, S, innov, verdict, &RM, REG == kRegAny ? &RW : nullptr);
                park[(slot + k) * kParkStride] = make_double2(S, innov);
                const unsigned t = (unsigned)(t0 + slot);
                Pw.store_s(x.P, p_off + k * 128u, t * 128u);
                mw.store_s(x.uq, m_off + k * 32u, t * 32u);
            };
            int slot = 0;
            // eight steps as straight-line code, then groups of four (a taken loop branch costs ~ 30 cycles: 2.38 -> 2.34 ms;
            // the same loop written generically over the group size -- arrays of read-ahead registers, one lambda for both
            // group sizes -- compiled to a schedule that gained nothing, with 8 or with 16 steps)
            double2 ya = *reinterpret_cast<const double2*>(ybuf), yb = *reinterpret_cast<const double2*>(ybuf + 2);
            double2 yc = *reinterpret_cast<const double2*>(ybuf + 4), yd = *reinterpret_cast<const double2*>(ybuf + 6);
            for (; slot + 8 <= nsteps; slot += 8) {
                const double2 na = *reinterpret_cast<const double2*>(ybuf + slot + 8), nb = *reinterpret_cast<const double2*>(ybuf + slot + 10);
                const double2 nc = *reinterpret_cast<const double2*>(ybuf + slot + 12), nd = *reinterpret_cast<const double2*>(ybuf + slot + 14);
                one(slot, 0u, ya.x); one(slot, 1u, ya.y); one(slot, 2u, yb.x); one(slot, 3u, yb.y);
                one(slot, 4u, yc.x); one(slot, 5u, yc.y); one(slot, 6u, yd.x); one(slot, 7u, yd.y);
                ya = na; yb = nb; yc = nc; yd = nd;
            }
            for (; slot + 4 <= nsteps; slot += 4) {
                const double2 na = *reinterpret_cast<const double2*>(ybuf + slot + 4), nb = *reinterpret_cast<const double2*>(ybuf + slot + 6);
                one(slot, 0u, ya.x); one(slot, 1u, ya.y); one(slot, 2u, yb.x); one(slot, 3u, yb.y);
                ya = na; yb = nb;
            }
            for (; slot < nsteps; slot++) one(slot, 0u, readlane_f64(ychunk, slot));
            anchor_end = anchor;
            return verdict.template code<REG>();
        };
        unsigned uncommon = 1;                                              // Ekf4Verdict::code of the chunk's last speculative pass
        // (where the state sits at the chunk's start is known: LOW and MID chunks are tried in their regime also while the sticky count runs;
        // a chunk that starts at the edge of a band goes to the ANY regime at once -- a pass that fails is a pass wasted, and the kernel's
        // time is that of its slowest wavefront)
        // Every speculative regime rotates by INCREMENTS; records whose angle jumps beyond the increment bound in many chunks (the CRLB jobs'
        // at dt = 0.01, where the ANY regime behind a failed pass failed the same way: 927 ns a step) back off: after a jump the speculative
        // regimes are off for 2, 4, 8, 16 chunks (doubling with every jump, one doubling back with every chunk kept in a regime), and those
        // chunks go straight to the wide step (spec_off; the score and the first-chunk rule below feed the same count): 453 - 491 ns a step
        // there, 432 - 455 for the library before the LOW / MID / ANY regimes, whose sticky count sent every such chunk to the wide step.
        // (Backing off after ANY failed pass, band exits too, gave the same there and cost the lost-track record sets 5 %.)
        const bool lean_tried = spec_allowed && (low || mid || (checked_left == 0 && !edge));
        if (lean_tried) {
            // A chunk that starts at u2 >= 6.5 is tried in the HIGH regime first (the bench records: 75 % of the chunks, none of
            // which falls out of it; with 5.5, round 3's threshold, 77 % and 1.7 % repeated: 1.4 % more time in all); one that
            // leaves it is repeated from its saved state in the common regime.
            // (Not for an NLL-only launch: that is the objective of a maximum-likelihood fit, differentiated by finite differences --
            // a chunk that changes regime between two probes would put a 1e-13 step into it, and the optimiser's path with it.)
            bool high = false;
            if constexpr (E1) high = !nll_final && hx0 - 0x401A0000u < 0x4085DFFFu - 0x401A0000u;
            if (low) {
                // (round 5) a chunk that starts at u2 <= -1.75 is tried in the LOW regime: records that have left the common regime sit
                // there most of the time
                uncommon = chunk(std::integral_constant<int, kRegLow>{});
                if (uncommon == 0) n_low++;
            } else if (mid) {
                uncommon = chunk(std::integral_constant<int, kRegMid>{});
                if (uncommon == 0) n_mid++;
            } else {
                if (high) {
                    uncommon = chunk(std::integral_constant<int, kRegHigh>{});
                    if (uncommon != 0) { x = x0; n_high_left++; } else n_high++;
                }
                if (uncommon != 0) {
                    uncommon = chunk(std::integral_constant<int, kRegCommon>{});
                    if (uncommon == 0) n_common++;
                }
            }
        }
        if (uncommon != 0) {
            // a record whose frequency state left the lean regimes probably stays outside: the next chunks go straight to the tiers
            // below; a single jump of the angle (code 2 alone) says nothing about the next chunk
            // (a chunk that left the LOW or the MID band says where the NEXT one starts, and that decides its regime: nothing sticky)
            if (lean_tried) { x = x0; checked_left = (!(low || mid) && (uncommon & 1u)) ? kCheckedChunks : 1; n_redo++; }
            // (round 5) first in the ANY regime -- the speculative step with the branch-free full-accuracy softplus for any |u2| < 700
            // (cgp_fastmath.hpp: softplus_pair_any) and the SAME incremental rotation: 6 dependent operations where the wide step below
            // takes a fresh sincos --, then ...
            // ... unless what failed was the increment bound (code 2): the ANY regime rotates by the same increments
            auto jumped = [&]() { jump_streak = jump_streak < 4 ? jump_streak + 1 : 4; if (spec_off < (1 << jump_streak)) spec_off = 1 << jump_streak; };
            // ... and a wavefront whose passes fail more often than they hold (score: +1 a chunk kept, -2 a failed pass, within [-16, 8])
            // stops trying for 32 chunks at a time: the CRLB jobs' records through this kernel, 6 % kept against 17 % failed
            // (a record whose very FIRST chunk fails -- the CRLB jobs' records are eight chunks long -- does not try again for eight)
            auto failed = [&]() {
                spec_score = spec_score > -14 ? spec_score - 2 : -16;
                if (spec_score <= -8 && spec_off < 32) spec_off = 32;
                if (t0 == span.t_begin && spec_off < 8) spec_off = 8;
            };
            if (lean_tried) { failed(); if ((uncommon & 2u) != 0) jumped(); }
            if (spec_allowed && spec_off == 0) {
                uncommon = chunk(std::integral_constant<int, kRegAny>{});
                if (uncommon == 0) { if (!lean_tried) n_wide++; }
                else { failed(); if ((uncommon & 2u) != 0) jumped(); }
            }
        }
        if (uncommon == 0) {
            anchor_live = anchor_end; anchor_age++;
            if (jump_streak > 0) jump_streak--;                              // (a chunk kept takes one doubling back, not all of them)
            if (spec_score < 8) spec_score++;
            if (checked_left > 0) checked_left--;
        } else {
            if (!spec_allowed) spec_off--;
            // ... on the wide step (full sincos: a jump of the angle beyond the increment bound); a chunk that leaves even that is repeated
            // with the checked step
            anchor_age = -1;
            x = x0;
            const bool redo = lean_tried;
            bool bad = false;
            WideRegs W;                                                     // (the wide step's constants, pinned for this chunk)
            W.init();
            auto wide_one = [&](int slot, double y) {
                double S, innov;
                ekf4_mfma_step_wide<E1 ? 2 : 0>(K, W, y, x, S, innov, bad);
                park[slot * kParkStride] = make_double2(S, innov);
                const unsigned t = (unsigned)(t0 + slot);
                Pw.store_s(x.P, p_off, t * 128u);
                mw.store_s(x.uq, m_off, t * 32u);
            };
            // four steps as straight-line code with their measurements read from LDS (a v_readlane pair per step: 24 issue cycles)
            int wslot = 0;
            for (; wslot + 4 <= nsteps; wslot += 4) {
                const double2 ya = *reinterpret_cast<const double2*>(ybuf + wslot), yb = *reinterpret_cast<const double2*>(ybuf + wslot + 2);
                wide_one(wslot, ya.x); wide_one(wslot + 1, ya.y); wide_one(wslot + 2, yb.x); wide_one(wslot + 3, yb.y);
            }
            for (; wslot < nsteps; wslot++) wide_one(wslot, readlane_f64(ychunk, wslot));
            if (__builtin_amdgcn_ballot_w64(bad) != 0) {
                x = x0;
                for (int slot = 0; slot < nsteps; slot++) {
                    double S, innov;
                    ekf4_mfma_step_checked<E1 ? 2 : 0>(K, readlane_f64(ychunk, slot), x, S, innov);
                    park[slot * kParkStride] = make_double2(S, innov);
                    const unsigned t = (unsigned)(t0 + slot);
                    Pw.store_s(x.P, p_off, t * 128u);
                    mw.store_s(x.uq, m_off, t * 32u);
                }
                if (!redo) n_checked++;
            } else if (!redo) n_wide++;
            if (checked_left > 0) checked_left--;
        }
        if (want_nll && !burn) {
            wave_lds_fence();
            const double2 si = park[(lane < nsteps ? lane : 0) * kParkStride];
            cum = nll_flush_wave(si.x, si.y, lane, nsteps, cum, nll ? nll + t0 : nullptr);
        }
    }
    if (span.state) {                                                       // the segment's last state and its NLL total, for the fix-up pass
        if (lane < 4) span.state[20 + lane] = x.uq;
        if (((lane >> 2) & 3) == 0) span.state[24 + 4 * r + q] = x.P;
        if (lane == 0) span.state[40] = cum;
    } else if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
    if (io.counters && lane == 0) {
        atomicAdd(io.counters + 0, (unsigned long long)n_high); atomicAdd(io.counters + 1, (unsigned long long)n_common);
        atomicAdd(io.counters + 2, (unsigned long long)n_redo); atomicAdd(io.counters + 3, (unsigned long long)n_checked);
        atomicAdd(io.counters + 4, (unsigned long long)n_high_left); atomicAdd(io.counters + 5, (unsigned long long)n_wide);
        atomicAdd(io.counters + 6, (unsigned long long)n_low); atomicAdd(io.counters + 7, (unsigned long long)n_mid);
    }
}

// ---------------------------------------------------------------------------------------------- linear model, d = 4: kf
// The same layout and update for a LINEAR cond_m_cov (kf, filters_smoothers.py:145-184; BASELINE config C1): the Jacobian is
// the constant F -- no softplus, no rotation, nothing to speculate on -- so a step is the eight matrix instructions and the
// rank-one update alone.  E1: the H = e_1 form of the update (see ekf4_mfma_finish).  What bounds such a step was measured
// with tools/ubench/kf_chain.hip: the matrix pipe takes one v_mfma_f64_4x4x4 per 20 cycles from a wavefront, in order (eight
// independent ones: 161 cycles), a dependent one 32; the step as the compiler orders it takes 216 cycles there, and an order
// pinned by hand with scheduling barriers (the chain's instructions first, the others in their shadows) 252 -- slower.
template <bool E1>
CGP_DEV void kf4_mfma_trial(const FilterIO& io, const ModelArgs& ma) {
    const int lane = threadIdx.x;
    const int r = lane >> 4, q = lane & 3;
    const int64_t trial = blockIdx.x;
    const double* __restrict__ prm = ma.params + trial * ma.param_stride;      // F (4 x 4, row-major) | Sigma (4 x 4)
    const double JT = prm[q * 4 + r];                                            // F[q][r]: A operand "F", B operand "F^T"
    const double Sig = prm[16 + r * 4 + q];
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const double Hr = Hp[r];
    const double Xi = io.Xi[trial * io.Xi_stride];

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    double ur = m0p[r], uq = m0p[q];
    double P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, r, q);

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    OobWindow mfs, Pfs;
    mfs.init(io.mfs ? io.mfs + trial * T * 4 : nullptr, T * 32);
    Pfs.init(io.Pfs ? io.Pfs + trial * T * 16 : nullptr, T * 128);
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool want_nll = io.nll != nullptr;
    const unsigned p_off = (((lane >> 2) & 3) == 0) ? 8u * (4 * r + q) : kOobOffset;
    const unsigned m_off = (lane < 4) ? 8u * lane : kOobOffset;

    __shared__ double2 park[64];
    __shared__ double ybuf[64 + 8];
    double cum = 0.0;
    double ynext = (lane < T) ? ys[lane] : 0.0;                                  // one chunk ahead (see ekf4_mfma_trial)
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        double ychunk = ynext;
        asm volatile("" : "+v"(ychunk));
        ynext = (t0 + 64 + lane < T) ? ys[t0 + 64 + lane] : 0.0;
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        wave_lds_fence();
        auto one = [&](int slot, double y) {
            const double f_r = mfma4(JT, ur, 0.0), f_q = mfma4(ur, JT, 0.0);   // F u by row and by column
            const double Q = mfma4(P, JT, 0.0);                                  // P F^T
            double Pp, PHr, PHq, S, innov;
            if constexpr (E1) {                                                  // see ekf4_mfma_finish: five matrix instructions
                Pp = mfma4(JT, Q, Sig);
                PHq = mfma4(Hr, Pp, 0.0);
                PHr = dpp_f64<kQuadBcast1>(Pp);                                  // (64-bit row broadcasts, which serve the EKF: 1.26 against 1.23 ms here)
                S = dpp_f64<kQuadBcast1>(PHq) + Xi;
                innov = y - dpp_f64<kQuadBcast1>(f_q);
            } else {
                Pp = mfma4(JT, Q, Sig);                                          // F P F^T + Sigma
                PHr = mfma4(Pp, Hr, 0.0);
                PHq = mfma4(Hr, Pp, 0.0);
                S = mfma4(Hr, PHr, Xi);
                innov = y - mfma4(Hr, f_r, 0.0);
            }
            const double rS = rcp_nr1(S);
            P = fma(-(PHr * rS), PHq, Pp);                                       // Pf = Pp - K K^T S (filters_smoothers.py:66)
            const double g = rS * innov;
            ur = fma(PHr, g, f_r);
            uq = fma(PHq, g, f_q);
            park[slot] = make_double2(S, innov);
            const unsigned t = (unsigned)(t0 + slot);
            Pfs.store_s(P, p_off, t * 128u);
            mfs.store_s(uq, m_off, t * 32u);
        };
        ybuf[lane] = ychunk;                                                     // measurements through LDS, as in ekf4_mfma_trial
        wave_lds_fence();
        int slot = 0;
        double2 ya = *reinterpret_cast<const double2*>(ybuf), yb = *reinterpret_cast<const double2*>(ybuf + 2);
        for (; slot + 4 <= nsteps; slot += 4) {
            const double2 na = *reinterpret_cast<const double2*>(ybuf + slot + 4), nb = *reinterpret_cast<const double2*>(ybuf + slot + 6);
            one(slot, ya.x); one(slot + 1, ya.y); one(slot + 2, yb.x); one(slot + 3, yb.y);
            ya = na; yb = nb;
        }
        for (; slot < nsteps; slot++) one(slot, readlane_f64(ychunk, slot));
        if (want_nll) {
            wave_lds_fence();
            const double2 si = park[lane < nsteps ? lane : 0];
            cum = nll_flush_wave(si.x, si.y, lane, nsteps, cum, nll ? nll + t0 : nullptr);
        }
    }
    if (lane == 0 && io.nll && nll_final) io.nll[trial] = cum;
}
__global__ void __launch_bounds__(64) kf4_mfma_kernel(FilterIO io, ModelArgs ma) {
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const bool e1 = Hp[0] == 0.0 && Hp[1] == 1.0 && Hp[2] == 0.0 && Hp[3] == 0.0;
    if (e1) kf4_mfma_trial<true>(io, ma);
    else kf4_mfma_trial<false>(io, ma);
}
inline int launch_kf4_mfma(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    if (io.T * 128 > kOobMaxBytes) return CGP_E_UNSUPPORTED;
    hipLaunchKernelGGL(kf4_mfma_kernel, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}

__global__ void __launch_bounds__(64) ekf4_mfma_kernel(FilterIO io, ModelArgs ma) {
    const int64_t trial = filter_span(io, blockIdx.x).trial;
    if (trial >= io.B) return;
    // the measurement vector of every chirp / La Scala builder is e_1 (models.py:118): the short-chain form of the update;
    // any other H (the API takes one per trial) runs the general form -- a wave-uniform choice
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    const bool e1 = Hp[0] == 0.0 && Hp[1] == 1.0 && Hp[2] == 0.0 && Hp[3] == 0.0;
    if (e1) ekf4_mfma_trial<true>(io, ma);
    else ekf4_mfma_trial<false>(io, ma);
}


#endif  // CGP_EKF4_KERNELS

// (round 5) the four-trials-per-wavefront kernel lives in a translation unit of its own (cgp_inst_ekf4_x4.hip): the one-trial kernel's
// unit is compiled without machine-level loop-invariant code motion (Makefile), which costs this one 14 %
inline bool ekf4_mfma_x4_fits(const FilterIO& io) { return io.T * 512 <= kOobMaxBytes; }
int launch_ekf4_mfma_x4(const FilterIO& io, const ModelArgs& ma, hipStream_t stream);

#ifdef CGP_EKF4_X4_KERNELS
// ---------------------------------------------------------------------------------------------- four trials per wave
// The four blocks of the MFMA are independent, so for batches beyond one wave per SIMD each block carries its own
// trial: lane 16 r + 4 b + q works on trial 4 * blockIdx.x + b.  The step functions above are used as they are -- they
// are plain per-lane arithmetic, the quad broadcasts stay inside a block -- with per-lane constants; what changes is
// the plumbing: measurements are staged in LDS (one 512-B load per trial and chunk, a broadcast ds_read per step), the
// verdicts of the four trials are pooled with a ballot (a chunk is repeated for the whole wave), the checked step and
// the anchor use the per-lane softplus / sincos, and the output windows span the wave's four consecutive trials.
// One lane per trial needs T x 1.3 us whatever the batch (its step is a 500-instruction dependent chain); this
// kernel needs T x 0.4 us per 4096 trials.
template <int E1 = 0>
CGP_DEV void ekf4_mfma_step_checked_lane(const Ekf4MfmaConst& K, double y, Ekf4State& x, double& S, double& innov) {
    double sp, dsp, s1, c1;
    // (round 5: the branch-free softplus for any |u2| < 700 first -- softplus_pair_wide's fast form holds for u2 >= 1.5 only, and a record
    // that is here has usually left that regime; beyond 700, inf and NaN take the reference's naive form)
    bool ok;
    softplus_pair_any(x.u2(), sp, dsp, ok);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
        double sp_n, dsp_n;
        softplus_pair(x.u2(), sp_n, dsp_n);
        sp = ok ? sp : sp_n;
        dsp = ok ? dsp : dsp_n;
    }
    fast_sincos(K.ang * sp, s1, c1);
    ekf4_mfma_finish<E1>(K, y, c1, s1, dsp, x, S, innov);
}
CGP_DEV void ekf4_anchor_lane(const Ekf4MfmaConst& K, double u2, Ekf4Anchor& a) {
    double sp, dsp, s1, c1;
    softplus_pair_wide(u2, sp, dsp);
    fast_sincos(K.ang * sp, s1, c1);
    a.set(K, sp, c1, s1);
}

// DENSE = false: constants pinned, 297 registers, one wave per SIMD (the dispatcher then has to spread the waves over all
// SIMDs: best up to 4096 trials).  DENSE = true: constants left to the compiler, 234 registers, two waves per SIMD
// (beyond 4096 trials, where waves have to share SIMDs anyway: 8 - 14 % faster there, 50 % slower below).
template <bool DENSE, bool E1>
CGP_DEV void ekf4_mfma_x4_trials(const FilterIO& io, const ModelArgs& ma) {
    const int lane = threadIdx.x;
    const int r = lane >> 4, q = lane & 3, b = (lane >> 2) & 3;
    const int64_t first = (int64_t)blockIdx.x * 4;
    const int ntr = (io.B - first < 4) ? (int)(io.B - first) : 4;          // trials of this wave
    const int64_t trial = first + (b < ntr ? b : ntr - 1);                 // blocks past the batch redo the last trial

    HarmonicLCD<1> model;
    model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
    Ekf4MfmaConst K;
    K.M0 = model.M[0]; K.M1 = model.M[1]; K.M2 = model.M[2]; K.M3 = model.M[3];
    K.rho = model.rho;
    K.ang = (model.dt * kTwoPi) * model.fs;
    const double* __restrict__ Hp = io.H + trial * io.H_stride;
    K.Hr = Hp[r];
    K.Xi = io.Xi[trial * io.Xi_stride];
    K.Sig = 0.0;
    if (r == q) K.Sig = (r < 2) ? model.q : (r == 2 ? model.MS[0] : model.MS[2]);
    else if (r + q == 5) K.Sig = model.MS[1];
    K.SigHq = (q == 1) ? model.q : 0.0;                                 // Sigma[1][q]: the chirp block of Sigma is q I
    K.c0 = model.q + K.Xi;
    K.kc = ((q == 0 && r == 0) || (q == 1 && r == 1)) ? 1.0 : 0.0;
    K.ks = (q == 0 && r == 1) ? -1.0 : ((q == 1 && r == 0) ? 1.0 : 0.0);
    K.kj = (r == 2 && q == 0) ? -1.0 : ((r == 2 && q == 1) ? 1.0 : 0.0);
    K.kk = (q == 2) ? (r == 2 ? K.M0 : (r == 3 ? K.M1 : 0.0)) : ((q == 3) ? (r == 2 ? K.M2 : (r == 3 ? K.M3 : 0.0)) : 0.0);
    K.fold();

    const double* __restrict__ m0p = io.m0 + trial * io.m0_stride;
    Ekf4State x;
    x.ur = m0p[r]; x.uq = m0p[q];
    x.P = coop4_load_sym_entry(io.P0 + trial * io.P0_stride, r, q);

    const int64_t T = io.T;
    // output windows over the wave's consecutive trials: a block past the batch lies beyond the window and is dropped
    OobWindow mfs, Pfs;
    mfs.init(io.mfs ? io.mfs + first * T * 4 : nullptr, (int64_t)ntr * T * 32);
    Pfs.init(io.Pfs ? io.Pfs + first * T * 16 : nullptr, (int64_t)ntr * T * 128);
    const unsigned p_base = (unsigned)b * (unsigned)T * 128u + 8u * (4 * r + q);
    const unsigned m_base = (r == 0) ? (unsigned)b * (unsigned)T * 32u + 8u * q : kOobOffset;
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    const bool want_nll = io.nll != nullptr;

    SpecRegs R;
    R.init<!DENSE>(K.angm);
    const SpecRegsHigh RH{};                                               // (the HIGH regime is tried by the one-trial-per-wavefront kernel only)
    __shared__ double ych[4][64];
    __shared__ double2 park[4][64];
    double cum[4] = {0.0, 0.0, 0.0, 0.0};
    int checked_left = 0;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        wave_lds_fence();
        CGP_UNROLL for (int bb = 0; bb < 4; bb++) {
            const int64_t tr = first + (bb < ntr ? bb : ntr - 1);
            ych[bb][lane] = (t0 + lane < T) ? io.record(tr)[t0 + lane] : 0.0;
        }
        wave_lds_fence();
        const Ekf4State x0 = x;
        unsigned long long uncommon = 0;                         // the verdicts of the four trials, pooled: a wave mask
        if (checked_left == 0) {
            Ekf4Anchor anchor;
            ekf4_anchor_lane(K, x.u2(), anchor);
            Ekf4Verdict verdict;
            auto one = [&](int slot) {
                double S, innov;
                ekf4_mfma_step_spec1<E1 ? 1 : 0>(K, R, RH, ych[b][slot], x, anchor, S, innov, verdict);
                park[b][slot] = make_double2(S, innov);
                const unsigned t = (unsigned)(t0 + slot);
                Pfs.store_s(x.P, p_base, t * 128u);              // the step's row offset rides in the scalar offset
                mfs.store_s(x.uq, m_base, t * 32u);
            };
            int slot = 0;
            for (; slot + kEkf4Unroll <= nsteps; slot += kEkf4Unroll) {
                CGP_UNROLL for (int k = 0; k < kEkf4Unroll; k++) one(slot + k);
            }
            for (; slot < nsteps; slot++) one(slot);
            uncommon = verdict.uncommon();
        }
        const bool redo = uncommon != 0;
        if (checked_left > 0 || redo) {
            if (redo) { x = x0; checked_left = kCheckedChunks; }
            for (int slot = 0; slot < nsteps; slot++) {
                double S, innov;
                ekf4_mfma_step_checked_lane<E1 ? 1 : 0>(K, ych[b][slot], x, S, innov);
                park[b][slot] = make_double2(S, innov);
                const unsigned t = (unsigned)(t0 + slot);
                Pfs.store_s(x.P, p_base, t * 128u);
                mfs.store_s(x.uq, m_base, t * 32u);
            }
            checked_left--;
        }
        if (want_nll) {
            wave_lds_fence();
            CGP_UNROLL for (int bb = 0; bb < 4; bb++) {
                const double2 si = park[bb][lane < nsteps ? lane : 0];
                double* out = (!nll_final && bb < ntr) ? io.nll + (first + bb) * T + t0 : nullptr;
                cum[bb] = nll_flush_wave(si.x, si.y, lane, nsteps, cum[bb], out);
            }
        }
    }
    if (lane == 0 && io.nll && nll_final) {
        CGP_UNROLL for (int bb = 0; bb < 4; bb++) if (bb < ntr) io.nll[first + bb] = cum[bb];
    }
}

template <bool DENSE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, DENSE ? 2 : 1))) ekf4_mfma_x4_kernel(FilterIO io, ModelArgs ma) {
    const int64_t first = (int64_t)blockIdx.x * 4;
    if (first >= io.B) return;
    // the H = e_1 form of the update when all four trials of the wave have it (a wave-uniform choice)
    const int bq = (threadIdx.x >> 2) & 3;
    const int ntr0 = (io.B - first < 4) ? (int)(io.B - first) : 4;
    const double* __restrict__ Hp = io.H + (first + (bq < ntr0 ? bq : ntr0 - 1)) * io.H_stride;
    const bool e1 = Hp[0] == 0.0 && Hp[1] == 1.0 && Hp[2] == 0.0 && Hp[3] == 0.0;
    if (__builtin_amdgcn_ballot_w64(!e1) == 0) ekf4_mfma_x4_trials<DENSE, true>(io, ma);
    else ekf4_mfma_x4_trials<DENSE, false>(io, ma);
}

// 4 T x 128 bytes of covariance rows must fit the 2 GiB window of a wave
inline int launch_ekf4_mfma_x4_impl(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B > 4096) hipLaunchKernelGGL(ekf4_mfma_x4_kernel<true>, dim3((unsigned)((io.B + 3) / 4)), dim3(64), 0, stream, io, ma);
    else hipLaunchKernelGGL(ekf4_mfma_x4_kernel<false>, dim3((unsigned)((io.B + 3) / 4)), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}
#endif  // CGP_EKF4_X4_KERNELS

#ifdef CGP_EKF4_KERNELS
inline int launch_ekf4_mfma(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return CGP_OK;
    // the kernels address a trial's covariance rows through a 32-bit byte offset into a 2 GiB buffer window: a record too
    // long for it is refused HERE, next to the kernels that need it (the dispatcher routes such records to the DPP kernel)
    if (io.T * 128 > kOobMaxBytes) return CGP_E_UNSUPPORTED;
    // beyond one wave per SIMD the four MFMA blocks carry four trials (CGP_ONE_TRIAL_PER_WAVE keeps one, for tests)
    if (io.segs > 1)           // time-split with burn-in: one wavefront per (trial, segment)
        hipLaunchKernelGGL(ekf4_mfma_kernel, dim3((unsigned)(io.B * io.segs)), dim3(64), 0, stream, io, ma);
    else if ((io.B > 1024 || (io.flags & CGP_FOUR_TRIALS_PER_WAVE)) && ekf4_mfma_x4_fits(io) && !(io.flags & CGP_ONE_TRIAL_PER_WAVE))
        return launch_ekf4_mfma_x4(io, ma, stream);
    else
        hipLaunchKernelGGL(ekf4_mfma_kernel, dim3((unsigned)io.B), dim3(64), 0, stream, io, ma);
    return hip_rc(hipGetLastError());
}

#endif  // CGP_EKF4_KERNELS

}  // namespace cgp
