// cgp_steps.hpp -- one time step of every filter / smoother of chirpgp/filters_smoothers.py, as device code.
//
// Two execution shapes share this code (template parameter WAVE):
//   WAVE = true   one 64-lane wavefront per trial.  Sigma-point fans are spread over the lanes (lane p evaluates
//                 points p, p + 64, ...), the weighted sums are reduced through LDS, everything else is computed
//                 redundantly (identically) by all lanes so no broadcast is ever needed.
//   WAVE = false  one lane per trial (large batches): the fan is a serial loop, no LDS.
#pragma once
#include <type_traits>
#include "cgp_math.hpp"
#include "cgp_models.hpp"

namespace cgp {

// ---------------------------------------------------------------------------------------------- sigma-point fan
constexpr int kRedChunk = 32;          // values reduced per LDS pass
constexpr int kRedLd = 66;             // row pitch in doubles: 64 lanes + 16 B pad (conflict-free ds_read_b128 across rows)
constexpr int kFanLdsDoubles = kRedChunk * kRedLd + kRedChunk;

// Sigma-point set.  The wave-per-trial kernels stage it in (dynamic) LDS once per workgroup: the fan re-reads the points
// every time step, and a global (L2) load on the T-serial chain costs ~500 cycles against ~100 for LDS.  Whether a kernel
// reads the staged copy is a compile-time property (template parameter ST of the accessors): the wave-per-trial shapes
// always do -- the host routes sets that do not fit to the one-lane-per-trial kernels -- so no per-access branch exists.
constexpr int kSigLdsMaxBytes = 44 * 1024;   // + 17 KB static reduction buffer < the 64 KB default LDS limit of a launch
// The large-batch lane kernel of the sigma-point filter (cgp_lane4.hpp: lane4_filter_kernel<SgpPredictLane>) has 28.7 KB of static LDS
// (measurement buffer 16 KB + covariance tile 3 KB + mean tile 9 KB) in front of the staged set: 34 KB more keeps a launch under 64 KB.
// A larger set (about 880 points at d = 4) takes the generic lane kernel, which reads its set from global memory.
constexpr int kLane4SigLdsMaxBytes = 34 * 1024;
// LDS pointers carry their address space explicitly: the reads become ds_read_* (not flat_load with an aperture test).
using LdsConstDoublePtr = const __attribute__((address_space(3))) double*;
using LdsConstIntPtr = const __attribute__((address_space(3))) int*;
typedef double double2_t __attribute__((ext_vector_type(2)));
using LdsConstDouble2Ptr = const __attribute__((address_space(3))) double2_t*;
struct SigmaSet {
    const double* __restrict__ xi;          // [s][d]   (global)
    const double* __restrict__ w;           // [s]
    const int* __restrict__ group_start;    // [n_groups + 1] or nullptr (every point its own group)
    int s, n_groups;
    unsigned flags;                          // CGP_SIGMA_* with CGP_SIGMA_STANDARD cleared when the launch wants the literal sums
    // LDS copies as 32-bit LDS addresses (this struct travels in the kernel arguments, where an address_space(3)
    // pointer member would have different sizes in the host and device layouts).
    unsigned lds_xi, lds_w, lds_gs, lds_tab;
    CGP_DEV int groups() const { return group_start ? n_groups : s; }
    template <bool ST> CGP_DEV int begin(int g) const {
        if (!group_start) return g;
        if constexpr (ST) return ((LdsConstIntPtr)lds_gs)[g]; else return group_start[g];
    }
    template <bool ST> CGP_DEV int end(int g) const {
        if (!group_start) return g + 1;
        if constexpr (ST) return ((LdsConstIntPtr)lds_gs)[g + 1]; else return group_start[g + 1];
    }
    template <bool ST> CGP_DEV double weight(int p) const {
        if constexpr (ST) return ((LdsConstDoublePtr)lds_w)[p]; else return w[p];
    }
    template <bool ST> CGP_DEV double coord(int idx) const {
        if constexpr (ST) return ((LdsConstDoublePtr)lds_xi)[idx]; else return xi[idx];
    }
    // The representative of group g for the collapsed quadratures: xi_0..D-2 of its first member and the group's total
    // weight.  Staged sets keep them as a table of D doubles per group (built once by stage()): D / 2 16-byte LDS reads
    // at an address that depends on g alone, instead of group bounds -> member weights -> coordinates one after the other.
    template <bool ST, int D> CGP_DEV void group(int g, double (&x)[D - 1], double& W) const {
        static_assert(D % 2 == 0, "collapsed quadratures: even state dimensions");
        if constexpr (ST) {
            const LdsConstDouble2Ptr t = (LdsConstDouble2Ptr)lds_tab + g * (D / 2);
            CGP_UNROLL for (int k = 0; k < D / 2; k++) {
                const double2_t v = t[k];
                x[2 * k] = v.x;
                if (2 * k + 1 < D - 1) x[2 * k + 1] = v.y; else W = v.y;
            }
        } else {
            const int p0 = begin<false>(g), p1 = end<false>(g);
            W = 0.0;
            for (int p = p0; p < p1; p++) W += w[p];
            CGP_UNROLL for (int c = 0; c < D - 1; c++) x[c] = xi[p0 * D + c];
        }
    }
    // doubles in front of the group table: points, weights, group bounds (ints), rounded up to a 16-byte boundary
    static inline size_t table_offset(int s, int d, int n_groups) { return ((size_t)s * d + s + (n_groups + 2) / 2 + 1) & ~(size_t)1; }
    static inline size_t stage_bytes(int s, int d, int n_groups, bool grouped) {
        return (grouped ? table_offset(s, d, n_groups) + (size_t)n_groups * d : (size_t)s * d + s) * sizeof(double);
    }
    // Cooperative copy into `buf` (stage_bytes of LDS); all `nthreads` threads of the block must call it.
    CGP_DEV void stage(double* buf, int tid, int nthreads, int d) {
        const int nxi = s * d, ngs = group_start ? n_groups + 1 : 0;
        for (int i = tid; i < nxi; i += nthreads) buf[i] = xi[i];
        for (int i = tid; i < s; i += nthreads) buf[nxi + i] = w[i];
        int* gs = reinterpret_cast<int*>(buf + nxi + s);
        for (int i = tid; i < ngs; i += nthreads) gs[i] = group_start[i];
        double* tab = buf + (((size_t)nxi + s + (n_groups + 2) / 2 + 1) & ~(size_t)1);
        if (group_start) {
            for (int g = tid; g < n_groups; g += nthreads) {
                const int p0 = group_start[g], p1 = group_start[g + 1];
                for (int c = 0; c < d - 1; c++) tab[g * d + c] = xi[p0 * d + c];
                double W = 0.0;
                for (int p = p0; p < p1; p++) W += w[p];
                tab[g * d + d - 1] = W;
            }
        }
        __syncthreads();
        lds_xi = (unsigned)(uintptr_t)(LdsConstDoublePtr)buf;
        lds_w = (unsigned)(uintptr_t)(LdsConstDoublePtr)(buf + nxi);
        lds_gs = (unsigned)(uintptr_t)(LdsConstIntPtr)gs;
        lds_tab = (unsigned)(uintptr_t)(LdsConstDoublePtr)tab;
    }
};

// All-lane sum of R per-lane partials through LDS: every lane ends with bitwise-identical totals.
// The workgroup IS one wavefront, whose DS instructions execute in order, so no s_barrier is needed between the
// phases -- only a wavefront-scope fence that keeps the compiler from reordering the LDS accesses.
// Pass of up to 16 values: lane (r, q) = (lane >> 2, lane & 3) sums one quarter (16 lanes' worth, read as eight
// 16-byte pairs) of row r, the four quarters meet with two cross-lane steps, lane q == 0 publishes the total and every
// lane reads the totals back (broadcast reads).  Idle lanes contribute zeros; the summation order is fixed.
constexpr int kRedPass = 16;
CGP_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int R>
CGP_DEV void wave_allreduce(double (&acc)[R], double* lds, int lane, int nl) {
    if (nl <= 16) {
        // At most 16 lanes carry partials (e.g. the 15 groups of the cubature rule in d = 8): one lane sums a whole
        // row, so up to 64 values go through in ONE pass -- 45 for the d = 8 prediction instead of three passes of 16.
        // Rows of 16 entries at a pitch of 18 doubles, totals behind them (64 * 18 + 64 doubles < the buffer).
        constexpr int kLd = 18;
        double* tot = lds + 64 * kLd;
        CGP_UNROLL for (int base = 0; base < R; base += 64) {
            if (lane < 16) {
                CGP_UNROLL for (int k = 0; k < 64; k++)
                    if (base + k < R) lds[k * kLd + lane] = acc[base + k];
            }
            wave_lds_fence();
            if (base + lane < R) {
                const double2* row = reinterpret_cast<const double2*>(lds + lane * kLd);
                double2 a = row[0], b = row[1], c = row[2], d = row[3];
                const double2 e = row[4], f = row[5], g = row[6], h = row[7];
                a.x += e.x; a.y += e.y; b.x += f.x; b.y += f.y; c.x += g.x; c.y += g.y; d.x += h.x; d.y += h.y;
                a.x += c.x; a.y += c.y; b.x += d.x; b.y += d.y;
                a.x += b.x; a.y += b.y;
                tot[lane] = a.x + a.y;
            }
            wave_lds_fence();
            CGP_UNROLL for (int k = 0; k < 64; k++)
                if (base + k < R) acc[base + k] = tot[k];
            wave_lds_fence();
        }
        return;
    }
    double* tot = lds + kRedChunk * kRedLd;
    const int r = lane >> 2, q = lane & 3;
    CGP_UNROLL for (int base = 0; base < R; base += kRedPass) {
        CGP_UNROLL for (int k = 0; k < kRedPass; k++)
            if (base + k < R) lds[k * kRedLd + lane] = acc[base + k];
        wave_lds_fence();
        const double2* row = reinterpret_cast<const double2*>(lds + r * kRedLd + q * 16);
        double s = 0.0;
        if (base + r < R) {
            CGP_UNROLL for (int j = 0; j < 8; j++) { const double2 v = row[j]; s += v.x; s += v.y; }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0) tot[r] = s;
        wave_lds_fence();
        CGP_UNROLL for (int k = 0; k < kRedPass; k++)
            if (base + k < R) acc[base + k] = tot[k];
        wave_lds_fence();
    }
}

// chi = m + L xi_p   (quadratures.py:198-201), L lower-triangular packed.
template <int D, bool ST> CGP_DEV void sigma_point(const Vec<D>& m, const Sym<D>& L, const SigmaSet& sg, int p, Vec<D>& chi) {
    double x[D];
    CGP_UNROLL for (int j = 0; j < D; j++) x[j] = sg.template coord<ST>(p * D + j);
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double t = L(i, 0) * x[0];
        CGP_UNROLL for (int j = 1; j <= i; j++) t = fma(L(i, j), x[j], t);
        chi.v[i] = m.v[i] + t;
    }
}

template <int NH> struct HarmonicLCD;
template <class DM> CGP_DEV bool sgp_collapsible(const SigmaSet& sg);
template <int NH, bool CROSS, bool ST>
CGP_DEV void sgpn_prediction_collapsed(const HarmonicLCD<NH>& model, const SigmaSet& sg, const Vec<2 * NH + 2>& mf, const Sym<2 * NH + 2>& Pf,
                                       Vec<2 * NH + 2>& mp, Sym<2 * NH + 2>& Pp, Mat<2 * NH + 2>& DT);
template <bool CROSS, bool ST>
CGP_DEV void sgp4_prediction_collapsed(const HarmonicLCD<1>& model, const SigmaSet& sg, const Vec<4>& mf, const Sym<4>& Pf,
                                       Vec<4>& mp, Sym<4>& Pp, Mat<4>& DT);

// Sigma-point prediction of a discrete model, filters_smoothers.py:88-121, plus (CROSS) the smoother's
// D^T = (sum_i w_i chi_i f_i^T - mf mp^T)^T, filters_smoothers.py:525.
// COLL (d >= 6 harmonic models, one lane doing the whole fan): the launch has checked on the host that the set may take the
// collapsed quadrature, so the kernel contains that path ONLY (both in one kernel cost registers: spills at d = 8).
template <class DM, bool WAVE, bool CROSS, bool ST = WAVE, bool COLL = false>
CGP_DEV void sgp_prediction(const DM& model, const SigmaSet& sg, int lane, double* lds,
                            const Vec<DM::D>& mf, const Sym<DM::D>& Pf, Vec<DM::D>& mp, Sym<DM::D>& Pp, Mat<DM::D>& DT) {
    constexpr int D = DM::D;
    constexpr int NS = Sym<D>::N;
    constexpr int R = 1 + D + NS + (CROSS ? D * D : 0);
    if constexpr (!WAVE && COLL && std::is_same<DM, HarmonicLCD<1>>::value) {
        // the launch has checked the set on the host: the kernel contains the collapsed path only (fewer registers, less code)
        sgp4_prediction_collapsed<CROSS, ST>(model, sg, mf, Pf, mp, Pp, DT);
        return;
    } else if constexpr (!WAVE && std::is_same<DM, HarmonicLCD<1>>::value) {
        // one lane does the whole fan: the collapsed quadrature where the set allows it (wave-uniform decision)
        if (sgp_collapsible<DM>(sg)) {
            sgp4_prediction_collapsed<CROSS, ST>(model, sg, mf, Pf, mp, Pp, DT);
            return;
        }
    } else if constexpr (!WAVE && COLL && (std::is_same<DM, HarmonicLCD<2>>::value || std::is_same<DM, HarmonicLCD<3>>::value)) {
        sgpn_prediction_collapsed<(DM::D - 2) / 2, CROSS, ST>(model, sg, mf, Pf, mp, Pp, DT);
        return;
    }
    Sym<D> L; Vec<D> inv;
    cholesky<D>(Pf, L, inv);
    double acc[R];
    CGP_UNROLL for (int r = 0; r < R; r++) acc[r] = 0.0;
    // Fan over groups of points that share chi_v (the coordinate the model is nonlinear in): one lane per group in the
    // wave shape, a serial loop otherwise; the model's transcendental part runs once per group.
    const int step = WAVE ? 64 : 1, ng = sg.groups();
    // one lane walking all groups anchors the rotation at the mean (cgp_models.hpp); with one group per lane the anchor
    // would only lengthen the chain
    typename DM::Anchor anchor;
    if (!WAVE) model.anchor(mf.v[DM::IVC], anchor);
    for (int g = WAVE ? lane : 0; g < ng; g += step) {
        int p = sg.template begin<ST>(g);
        const int pe = sg.template end<ST>(g);
        Vec<D> chi, f;
        sigma_point<D, ST>(mf, L, sg, p, chi);
        typename DM::Pre pre;
        if (WAVE) model.precompute(chi.v[DM::IVC], pre);
        else model.precompute(chi.v[DM::IVC], anchor, pre);
        for (;;) {
            model.mean_pre(chi, pre, f);
            const double w = sg.template weight<ST>(p);
            acc[0] += w;
            double wf[D];
            CGP_UNROLL for (int i = 0; i < D; i++) { wf[i] = w * f.v[i]; acc[1 + i] += wf[i]; }
            CGP_UNROLL for (int i = 0; i < D; i++)
                CGP_UNROLL for (int j = 0; j <= i; j++)
                    acc[1 + D + Sym<D>::idx(i, j)] = fma(wf[i], f.v[j], acc[1 + D + Sym<D>::idx(i, j)]);
            if (CROSS) {
                CGP_UNROLL for (int i = 0; i < D; i++)
                    CGP_UNROLL for (int j = 0; j < D; j++)
                        acc[1 + D + NS + i * D + j] = fma(chi.v[i], wf[j], acc[1 + D + NS + i * D + j]);
            }
            if (++p >= pe) break;
            sigma_point<D, ST>(mf, L, sg, p, chi);
        }
    }
    if (WAVE) wave_allreduce<R>(acc, lds, lane, ng < 64 ? ng : 64);
    CGP_UNROLL for (int i = 0; i < D; i++) mp.v[i] = acc[1 + i];
    // Pp = E[f f^T + Sigma] - mp mp^T ; E[Sigma] = (sum_i w_i) Sigma since Sigma does not depend on the point (N3)
    CGP_UNROLL for (int i = 0; i < NS; i++) Pp.a[i] = acc[1 + D + i];
    model.add_sigma(Pp, acc[0]);
    CGP_UNROLL for (int i = 0; i < D; i++)
        CGP_UNROLL for (int j = 0; j <= i; j++) Pp(i, j) -= mp.v[i] * mp.v[j];
    if (CROSS) {
        CGP_UNROLL for (int i = 0; i < D; i++)
            CGP_UNROLL for (int j = 0; j < D; j++) DT.a[j][i] = acc[1 + D + NS + i * D + j] - mf.v[i] * mp.v[j];
    }
}

// The same prediction for the d = 4 chirp / La Scala LCD model and a CGP_SIGMA_STANDARD set, with the linear structure
// taken out of the quadrature (include/chirpgp_hip.h).  f_0, f_1 = rho Rot(theta(chi_2)) (chi_0, chi_1) depend on
// xi_0..2 only (L is lower-triangular), f_2, f_3 = M (chi_2, chi_3) are linear, and sum w = 1, sum w xi = 0,
// sum w xi xi^T = I, so with W_g the total weight of a group and d = L xi restricted to xi_0..2 (d_3 without L_33):
//     mp_a = sum_g W_g f_a                                        a < 2        mp_{2,3} = M m_{2,3}
//     Pp_ab = sum_g W_g f_a f_b - mp_a mp_b + q delta_ab          a, b < 2     Pp_{2..3,2..3} = M P_{2..3,2..3} M^T + Sigma
//     Pp_ab = sum_c M_bc X_ac,  X_ac = sum_g W_g f_a d_{2+c}      a < 2 <= b
//     D[i][a] = sum_g W_g d_i f_a   (= X for i >= 2)              a < 2        D[i][b] = sum_c M_bc P[i][2+c]    b >= 2
// an exact regrouping of filters_smoothers.py:88-121 / :525 -- one evaluation per GROUP (27 for Gauss-Hermite order 3
// instead of 81 points), 9 (13 with the cross term) partial sums instead of 15 (31).  One lane does all groups; a
// failed Cholesky poisons every output with NaN like the literal sums do.
// SPEC: the groups' softplus and rotation without regime branches (cgp_models.hpp:precompute_spec), so the loop body is ONE
// basic block; returns false if some group of this lane was outside the regime -- the caller then repeats with SPEC = false.
template <bool CROSS, bool ST, int MODE>          // MODE: cgp_models.hpp kFanChecked / kFanSpec / kFanAny
CGP_DEV bool sgp4_prediction_collapsed_impl(const HarmonicLCD<1>& model, const SigmaSet& sg, const Vec<4>& mf, const Sym<4>& Pf,
                                            Vec<4>& mp, Sym<4>& Pp, Mat<4>& DT) {
    Sym<4> L; Vec<4> inv;
    bool all_ok = true;
    cholesky<4>(Pf, L, inv);
    double sf0 = 0.0, sf1 = 0.0, s00 = 0.0, s10 = 0.0, s11 = 0.0;
    double x02 = 0.0, x12 = 0.0, x03 = 0.0, x13 = 0.0, c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
    const int ng = sg.groups();
    HarmonicLCD<1>::Anchor anchor;
    model.anchor(mf.v[2], anchor);
    auto group = [&](int g) __attribute__((always_inline)) {
        double xg[3], W;
        sg.template group<ST, 4>(g, xg, W);
        const double xi0 = xg[0], xi1 = xg[1], xi2 = xg[2];
        const double d0 = L(0, 0) * xi0;
        const double d1 = fma(L(1, 1), xi1, L(1, 0) * xi0);
        const double d2 = fma(L(2, 2), xi2, fma(L(2, 1), xi1, L(2, 0) * xi0));
        const double d3 = fma(L(3, 2), xi2, fma(L(3, 1), xi1, L(3, 0) * xi0));
        const double h0 = mf.v[0] + d0, h1 = mf.v[1] + d1;
        HarmonicLCD<1>::Pre pre;
        if constexpr (MODE == kFanSpec) { bool ok; model.precompute_spec(mf.v[2] + d2, anchor, pre, ok); all_ok = all_ok && ok; }
        else if constexpr (MODE == kFanAny) { bool ok; model.precompute_any(mf.v[2] + d2, anchor, pre, ok); all_ok = all_ok && ok; }
        else model.precompute(mf.v[2] + d2, anchor, pre);
        const double f0 = pre.c[0] * h0 - pre.s[0] * h1, f1 = pre.s[0] * h0 + pre.c[0] * h1;
        const double w0 = W * f0, w1 = W * f1;
        sf0 += w0; sf1 += w1;
        s00 = fma(w0, f0, s00); s10 = fma(w1, f0, s10); s11 = fma(w1, f1, s11);
        x02 = fma(w0, d2, x02); x12 = fma(w1, d2, x12); x03 = fma(w0, d3, x03); x13 = fma(w1, d3, x13);
        if (CROSS) { c00 = fma(w0, d0, c00); c01 = fma(w1, d0, c01); c10 = fma(w0, d1, c10); c11 = fma(w1, d1, c11); }
    };
    if constexpr (MODE != kFanChecked) {
        // no branch in a group: three groups per iteration, their softplus -> sin / cos chains interleaved by the scheduler (round 5:
        // GH-3 at 262 144 x 500 on the CRLB records 17.2 -> 14.3 (two) -> 14.0 ms (three))
        int g = 0;
        for (; g + 3 <= ng; g += 3) { group(g); group(g + 1); group(g + 2); }
        for (; g < ng; g++) group(g);
    } else {
        for (int g = 0; g < ng; g++) group(g);
    }
    const double poison = L(0, 0) - L(0, 0);          // 0, or NaN when the factorisation failed
    const double M0 = model.M[0], M1 = model.M[1], M2 = model.M[2], M3 = model.M[3];
    mp.v[0] = sf0; mp.v[1] = sf1;
    mp.v[2] = fma(M0, mf.v[2], M1 * mf.v[3]) + poison;
    mp.v[3] = fma(M2, mf.v[2], M3 * mf.v[3]) + poison;
    Pp(0, 0) = (s00 - sf0 * sf0) + model.q;
    Pp(1, 0) = s10 - sf1 * sf0;
    Pp(1, 1) = (s11 - sf1 * sf1) + model.q;
    Pp(2, 0) = fma(M0, x02, M1 * x03); Pp(2, 1) = fma(M0, x12, M1 * x13);
    Pp(3, 0) = fma(M2, x02, M3 * x03); Pp(3, 1) = fma(M2, x12, M3 * x13);
    const double t20 = fma(M0, Pf(2, 2), M1 * Pf(3, 2)), t21 = fma(M0, Pf(3, 2), M1 * Pf(3, 3));
    const double t30 = fma(M2, Pf(2, 2), M3 * Pf(3, 2)), t31 = fma(M2, Pf(3, 2), M3 * Pf(3, 3));
    Pp(2, 2) = (fma(t20, M0, t21 * M1) + model.MS[0]) + poison;
    Pp(3, 2) = (fma(t30, M0, t31 * M1) + model.MS[1]) + poison;
    Pp(3, 3) = (fma(t30, M2, t31 * M3) + model.MS[2]) + poison;
    if (CROSS) {
        // DT[j][i] = D[i][j]
        DT.a[0][0] = c00; DT.a[0][1] = c10; DT.a[0][2] = x02; DT.a[0][3] = x03;
        DT.a[1][0] = c01; DT.a[1][1] = c11; DT.a[1][2] = x12; DT.a[1][3] = x13;
        CGP_UNROLL for (int i = 0; i < 4; i++) {
            const double pi2 = i >= 2 ? Pf(i, 2) : Pf(2, i), pi3 = Pf(3, i);
            DT.a[2][i] = fma(M0, pi2, M1 * pi3) + poison;
            DT.a[3][i] = fma(M2, pi2, M3 * pi3) + poison;
        }
    }
    return all_ok;
}
template <bool CROSS, bool ST>
CGP_DEV void sgp4_prediction_collapsed(const HarmonicLCD<1>& model, const SigmaSet& sg, const Vec<4>& mf, const Sym<4>& Pf,
                                       Vec<4>& mp, Sym<4>& Pp, Mat<4>& DT) {
    const bool ok = sgp4_prediction_collapsed_impl<CROSS, ST, kFanSpec>(model, sg, mf, Pf, mp, Pp, DT);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
        const bool ok2 = sgp4_prediction_collapsed_impl<CROSS, ST, kFanAny>(model, sg, mf, Pf, mp, Pp, DT);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok2) != 0, 0)) sgp4_prediction_collapsed_impl<CROSS, ST, kFanChecked>(model, sg, mf, Pf, mp, Pp, DT);
    }
}
// The same for a caller that evaluates the fan step after step on the same 64 trials (one lane per trial, cgp_lane4.hpp): a wavefront
// whose speculative pass failed -- a lane's frequency state below 1.5: the CRLB jobs' records, a zero-mean GP, in every step -- pays the
// speculative AND the checked fan, 5852 vector instructions per step at GH-3; it goes straight to the ANY fan (full-accuracy softplus,
// no branch in a group: precompute_any; the checked fan behind it only beyond 700 or for a wide fan) for the next kSpecSkip steps instead
// (`skip`: wave-uniform, kept by the caller).
constexpr int kSpecSkip = 16;
template <bool CROSS, bool ST>
CGP_DEV void sgp4_prediction_collapsed_sticky(const HarmonicLCD<1>& model, const SigmaSet& sg, const Vec<4>& mf, const Sym<4>& Pf,
                                              Vec<4>& mp, Sym<4>& Pp, Mat<4>& DT, int& skip) {
    if (skip > 0) {
        skip--;
        const bool ok = sgp4_prediction_collapsed_impl<CROSS, ST, kFanAny>(model, sg, mf, Pf, mp, Pp, DT);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) sgp4_prediction_collapsed_impl<CROSS, ST, kFanChecked>(model, sg, mf, Pf, mp, Pp, DT);
        return;
    }
    const bool ok = sgp4_prediction_collapsed_impl<CROSS, ST, kFanSpec>(model, sg, mf, Pf, mp, Pp, DT);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {
        const bool ok2 = sgp4_prediction_collapsed_impl<CROSS, ST, kFanAny>(model, sg, mf, Pf, mp, Pp, DT);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok2) != 0, 0)) sgp4_prediction_collapsed_impl<CROSS, ST, kFanChecked>(model, sg, mf, Pf, mp, Pp, DT);
        skip = kSpecSkip;
    }
}
// The same regrouping for n harmonics (d = 2 n + 2; rotating components a < 2 n, linear pair v = 2 n, v + 1), one lane doing
// all groups: with d = L xi restricted to xi_0..d-2 and g_a = f_a(m + d) of a group's representative,
//     mp_a = sum W g_a,   Pp_ab = sum W g_a g_b - mp_a mp_b + q delta_ab,   Pp_{a, v+b} = sum_c M_bc (sum W g_a d_{v+c}),
//     mp_lin = M m_lin,   Pp_lin,lin = M P_lin,lin M^T + Sigma_lin,
//     D[i][a] = sum W d_i g_a   (cross covariance, smoother),   D[i][v+b] = sum_c M_bc P[i][v+c]
// -- 15 evaluations and 39 (75 with the cross term) partial sums for the cubature rule in d = 8, instead of 16 and 45 (109).
// SPEC as in sgp4_prediction_collapsed_impl.
template <int NH, bool CROSS, bool ST, bool SPEC>
CGP_DEV bool sgpn_prediction_collapsed_impl(const HarmonicLCD<NH>& model, const SigmaSet& sg, const Vec<2 * NH + 2>& mf, const Sym<2 * NH + 2>& Pf,
                                            Vec<2 * NH + 2>& mp, Sym<2 * NH + 2>& Pp, Mat<2 * NH + 2>& DT) {
    constexpr int D = 2 * NH + 2, NL = 2 * NH, V = NL;
    bool all_ok = true;
    Sym<D> L; Vec<D> inv;
    cholesky<D>(Pf, L, inv);                       // L[D-1][D-1] is never used: its square root is dead code
    double sf[NL], sab[NL * (NL + 1) / 2], x[NL][2], cr[CROSS ? NL : 1][NL];
    CGP_UNROLL for (int a = 0; a < NL; a++) { sf[a] = 0.0; x[a][0] = 0.0; x[a][1] = 0.0; }
    CGP_UNROLL for (int a = 0; a < NL * (NL + 1) / 2; a++) sab[a] = 0.0;
    if (CROSS) { CGP_UNROLL for (int a = 0; a < NL; a++) CGP_UNROLL for (int c = 0; c < NL; c++) cr[a][c] = 0.0; }
    const int ng = sg.groups();
    typename HarmonicLCD<NH>::Anchor anchor;
    model.anchor(mf.v[V], anchor);
    for (int g = 0; g < ng; g++) {
        double xi[D - 1], dd[D], W;
        sg.template group<ST, D>(g, xi, W);
        CGP_UNROLL for (int a = 0; a < D; a++) {
            double t = L(a, 0) * xi[0];
            CGP_UNROLL for (int c = 1; c <= (a < D - 1 ? a : D - 2); c++) t = fma(L(a, c), xi[c], t);
            dd[a] = t;
        }
        typename HarmonicLCD<NH>::Pre pre;
        if constexpr (SPEC) { bool ok; model.precompute_spec(mf.v[V] + dd[V], anchor, pre, ok); all_ok = all_ok && ok; }
        else model.precompute(mf.v[V] + dd[V], anchor, pre);
        double wg[NL], gg[NL];
        CGP_UNROLL for (int k = 0; k < NH; k++) {
            const double h0 = mf.v[2 * k] + dd[2 * k], h1 = mf.v[2 * k + 1] + dd[2 * k + 1];
            gg[2 * k] = pre.c[k] * h0 - pre.s[k] * h1;
            gg[2 * k + 1] = pre.s[k] * h0 + pre.c[k] * h1;
        }
        CGP_UNROLL for (int a = 0; a < NL; a++) {
            wg[a] = W * gg[a];
            sf[a] += wg[a];
            CGP_UNROLL for (int c = 0; c <= a; c++) sab[a * (a + 1) / 2 + c] = fma(wg[a], gg[c], sab[a * (a + 1) / 2 + c]);
            x[a][0] = fma(wg[a], dd[V], x[a][0]);
            x[a][1] = fma(wg[a], dd[V + 1], x[a][1]);
            if (CROSS) { CGP_UNROLL for (int c = 0; c < NL; c++) cr[c][a] = fma(wg[a], dd[c], cr[c][a]); }
        }
    }
    const double poison = L(0, 0) - L(0, 0);          // 0, or NaN when the factorisation failed
    const double M0 = model.M[0], M1 = model.M[1], M2 = model.M[2], M3 = model.M[3];
    CGP_UNROLL for (int a = 0; a < NL; a++) mp.v[a] = sf[a];
    mp.v[V] = fma(M0, mf.v[V], M1 * mf.v[V + 1]) + poison;
    mp.v[V + 1] = fma(M2, mf.v[V], M3 * mf.v[V + 1]) + poison;
    CGP_UNROLL for (int a = 0; a < NL; a++) {
        CGP_UNROLL for (int c = 0; c <= a; c++) {
            const double v = sab[a * (a + 1) / 2 + c] - sf[a] * sf[c];
            Pp(a, c) = (a == c) ? v + model.q : v;
        }
        Pp(V, a) = fma(M0, x[a][0], M1 * x[a][1]);
        Pp(V + 1, a) = fma(M2, x[a][0], M3 * x[a][1]);
    }
    const double t20 = fma(M0, Pf(V, V), M1 * Pf(V + 1, V)), t21 = fma(M0, Pf(V + 1, V), M1 * Pf(V + 1, V + 1));
    const double t30 = fma(M2, Pf(V, V), M3 * Pf(V + 1, V)), t31 = fma(M2, Pf(V + 1, V), M3 * Pf(V + 1, V + 1));
    Pp(V, V) = (fma(t20, M0, t21 * M1) + model.MS[0]) + poison;
    Pp(V + 1, V) = (fma(t30, M0, t31 * M1) + model.MS[1]) + poison;
    Pp(V + 1, V + 1) = (fma(t30, M2, t31 * M3) + model.MS[2]) + poison;
    if (CROSS) {
        // DT[j][i] = D[i][j]
        CGP_UNROLL for (int a = 0; a < NL; a++) {
            CGP_UNROLL for (int c = 0; c < NL; c++) DT.a[a][c] = cr[c][a];
            DT.a[a][V] = x[a][0]; DT.a[a][V + 1] = x[a][1];
        }
        CGP_UNROLL for (int c = 0; c < D; c++) {
            const double pv = Pf(c, V), pw = Pf(c, V + 1);          // Sym::operator() is symmetric in its arguments
            DT.a[V][c] = fma(M0, pv, M1 * pw) + poison;
            DT.a[V + 1][c] = fma(M2, pv, M3 * pw) + poison;
        }
    }
    return all_ok;
}
template <int NH, bool CROSS, bool ST>
CGP_DEV void sgpn_prediction_collapsed(const HarmonicLCD<NH>& model, const SigmaSet& sg, const Vec<2 * NH + 2>& mf, const Sym<2 * NH + 2>& Pf,
                                       Vec<2 * NH + 2>& mp, Sym<2 * NH + 2>& Pp, Mat<2 * NH + 2>& DT) {
    // the branch-free loop pays where a lane runs the whole filter (C5 parameter sweep 49 -> 45.5 ms per 262144 x 1000); in the
    // d = 6 / 8 smoother, whose lanes also carry the cooperative walk's state, it measured slower (C5 smoother 4.74 -> 5.17 ms)
    if constexpr (ST) sgpn_prediction_collapsed_impl<NH, CROSS, ST, false>(model, sg, mf, Pf, mp, Pp, DT);
    else {
        const bool ok = sgpn_prediction_collapsed_impl<NH, CROSS, ST, true>(model, sg, mf, Pf, mp, Pp, DT);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) sgpn_prediction_collapsed_impl<NH, CROSS, ST, false>(model, sg, mf, Pf, mp, Pp, DT);
    }
}
// Whether a launch may take the collapsed path: the caller's assertion, groups, and the d = 4 harmonic family (run-time
// check inside the kernel; the d >= 6 kernels are compiled per path, see COLL above and sgp_collapsible_host).
template <class DM> CGP_DEV bool sgp_collapsible(const SigmaSet& sg) {
    if constexpr (std::is_same<DM, HarmonicLCD<1>>::value) return (sg.flags & 1u /* CGP_SIGMA_STANDARD */) && sg.group_start;
    else return false;
}

// Sigma-point moment ODE of an SDE model, filters_smoothers.py:124-137: dm = E[a], dP = C + C^T + gamma,
// C = E[(chi - m) a^T].
template <class SM, bool WAVE, bool ST = WAVE>
CGP_DEV void cd_sgp_common(const SM& model, const SigmaSet& sg, int lane, double* lds, const Sym<SM::D>& gamma,
                           const Vec<SM::D>& m, const Sym<SM::D>& P, Vec<SM::D>& dm, Sym<SM::D>& dP) {
    constexpr int D = SM::D;
    constexpr int R = D + D * D;
    Sym<D> L; Vec<D> inv;
    cholesky<D>(P, L, inv);
    double acc[R];
    CGP_UNROLL for (int r = 0; r < R; r++) acc[r] = 0.0;
    const int step = WAVE ? 64 : 1, ng = sg.groups();
    for (int g = WAVE ? lane : 0; g < ng; g += step) {
        int p = sg.template begin<ST>(g);
        const int pe = sg.template end<ST>(g);
        Vec<D> chi, a;
        sigma_point<D, ST>(m, L, sg, p, chi);
        typename SM::Pre pre;
        model.precompute(chi.v[SM::IVC], pre);
        for (;;) {
            model.drift_pre(chi, pre, a);
            const double w = sg.template weight<ST>(p);
            double wa[D];
            CGP_UNROLL for (int i = 0; i < D; i++) { wa[i] = w * a.v[i]; acc[i] += wa[i]; }
            CGP_UNROLL for (int i = 0; i < D; i++) {
                const double ci = chi.v[i] - m.v[i];
                CGP_UNROLL for (int j = 0; j < D; j++) acc[D + i * D + j] = fma(ci, wa[j], acc[D + i * D + j]);
            }
            if (++p >= pe) break;
            sigma_point<D, ST>(m, L, sg, p, chi);
        }
    }
    if (WAVE) wave_allreduce<R>(acc, lds, lane, ng < 64 ? ng : 64);
    CGP_UNROLL for (int i = 0; i < D; i++) dm.v[i] = acc[i];
    CGP_UNROLL for (int i = 0; i < D; i++)
        CGP_UNROLL for (int j = 0; j <= i; j++) dP(i, j) = (acc[D + i * D + j] + acc[D + j * D + i]) + gamma(i, j);
}

// ---------------------------------------------------------------------------------------------- RK4 on the (m, P) pair
// quadratures.py:34-54 / 57-81: k1..k4, x + dt (k1 + 2 k2 + 2 k3 + k4) / 6; the stage loop is kept rolled (code size).
template <int D, class Rhs>
CGP_DEV void rk4_m_cov(Rhs&& rhs, Vec<D>& m, Sym<D>& P, double dt) {
    Vec<D> km, tm = m, am;
    Sym<D> kP, tP = P, aP;
    CGP_UNROLL for (int i = 0; i < D; i++) am.v[i] = 0.0;
    CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) aP.a[i] = 0.0;
#pragma unroll 1
    for (int stage = 0; stage < 4; stage++) {
        rhs(tm, tP, km, kP);
        const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
        const double half = (stage == 2) ? 1.0 : 0.5;      // (dt * k) / 2 == (dt * k) * 0.5 exactly
        CGP_UNROLL for (int i = 0; i < D; i++) { am.v[i] = fma(wgt, km.v[i], am.v[i]); tm.v[i] = m.v[i] + (dt * km.v[i]) * half; }
        CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) { aP.a[i] = fma(wgt, kP.a[i], aP.a[i]); tP.a[i] = P.a[i] + (dt * kP.a[i]) * half; }
    }
    CGP_UNROLL for (int i = 0; i < D; i++) m.v[i] = m.v[i] + (dt * am.v[i]) / 6.0;
    CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) P.a[i] = P.a[i] + (dt * aP.a[i]) / 6.0;
}

// ---------------------------------------------------------------------------------------------- per-trial arguments
struct ModelArgs {
    const double* __restrict__ params; int64_t param_stride;
    const double* __restrict__ gamma;  int64_t gamma_stride;
    int model_id;
    SigmaSet sg;
    double dt;
};

// ============================================================================================== FILTER PREDICTORS
// predict(lane, lds, mf, Pf) -> (mp, Pp)

// ekf (filters_smoothers.py:251-261) and, with a linear model, kf (:174-181)
template <class DM, bool WAVE_> struct EkfPredict {
    static constexpr bool USES_SIGMA = false;
    static constexpr bool LANE_TWO_WAVES = true;      // one lane per trial: fits 256 registers (cgp_kernels.hpp: filter_kernel)
    static constexpr int D = DM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = false;
    DM model;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); model.uniform = WAVE; model.wide = WAVE; }
    CGP_DEV void large_batch() {                                   // cgp_lane4.hpp
        model.large_batch = true;
        if constexpr (std::is_same<DM, HarmonicLCD<1>>::value) model.setup_blocks();
    }
    CGP_DEV void predict(int, double*, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& mp, Sym<D>& Pp) const {
        if constexpr (std::is_same<DM, HarmonicLCD<1>>::value) {
            if (model.large_batch) { model.propagate_blocks(mf, Pf, mp, Pp); return; }      // large-batch launch: the block form, no T
        }
        Mat<D> T;
        model.propagate(mf, Pf, mp, T, Pp);
    }
};

// sgp_filter (filters_smoothers.py:480-487)
template <class DM, bool WAVE_, bool COLL = false> struct SgpPredict {
    static constexpr bool USES_SIGMA = true;
    static constexpr bool LANE_TWO_WAVES = false;
    static constexpr int D = DM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = WAVE_;
    DM model; SigmaSet sg;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); sg = a.sg; model.wide = WAVE; }
    CGP_DEV void predict(int lane, double* lds, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& mp, Sym<D>& Pp) const {
        Mat<D> unused;
        sgp_prediction<DM, WAVE, false, WAVE, COLL>(model, sg, lane, lds, mf, Pf, mp, Pp, unused);
    }
};

// the same with one lane per trial and the sigma-point set STAGED in LDS (cgp_lane4.hpp): a lane walks the whole fan, and its group
// table must not come from global memory -- a load behind the step's output stores waits for all of them (one in-order counter)
template <class DM, bool COLL = false> struct SgpPredictLane {
    static constexpr bool USES_SIGMA = true;
    static constexpr bool LANE_TWO_WAVES = false;
    static constexpr int D = DM::D; static constexpr bool WAVE = false; static constexpr bool USES_LDS = false;
    DM model; SigmaSet sg;
    mutable int skip = 0;                      // steps left on the checked fan (sgp4_prediction_collapsed_sticky)
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); sg = a.sg; model.wide = false; }
    CGP_DEV void large_batch() {                                   // cgp_lane4.hpp: the checked fan's softplus without its overflow selects
        if constexpr (std::is_same<DM, HarmonicLCD<1>>::value) model.large_batch = true;
    }
    CGP_DEV void predict(int lane, double* lds, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& mp, Sym<D>& Pp) const {
        Mat<D> unused;
        if constexpr (std::is_same<DM, HarmonicLCD<1>>::value) {
            if (COLL || sgp_collapsible<DM>(sg)) { sgp4_prediction_collapsed_sticky<false, true>(model, sg, mf, Pf, mp, Pp, unused, skip); return; }
        }
        sgp_prediction<DM, false, false, true, COLL>(model, sg, lane, lds, mf, Pf, mp, Pp, unused);
    }
};

// cd_ekf (filters_smoothers.py:384-394): dm = a(m), dP = P J^T + J P + gamma
template <class SM, bool WAVE_> struct CdEkfPredict {
    static constexpr bool USES_SIGMA = false;
    static constexpr bool LANE_TWO_WAVES = false;
    static constexpr int D = SM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = false;
    SM model; Sym<D> gamma; double dt;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) {
        model.setup(a.params + trial * a.param_stride, a.model_id);
        model.wide = WAVE;
        model.uniform = WAVE;
        load_sym<D>(a.gamma + trial * a.gamma_stride, gamma);
        dt = a.dt;
    }
    CGP_DEV void predict(int, double*, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& mp, Sym<D>& Pp) const {
        mp = mf; Pp = Pf;
        rk4_m_cov<D>([&](const Vec<D>& m, const Sym<D>& P, Vec<D>& dm, Sym<D>& dP) {
            Mat<D> T;
            model.drift_jp(m, P, dm, T);
            sym_from_sum<D>(T, gamma, 1.0, dP);
        }, mp, Pp, dt);
    }
};

// cd_sgp_filter (filters_smoothers.py:569-579)
template <class SM, bool WAVE_> struct CdSgpPredict {
    static constexpr bool USES_SIGMA = true;
    static constexpr bool LANE_TWO_WAVES = false;
    static constexpr int D = SM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = WAVE_;
    SM model; Sym<D> gamma; SigmaSet sg; double dt;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) {
        model.setup(a.params + trial * a.param_stride, a.model_id);
        model.wide = WAVE;
        load_sym<D>(a.gamma + trial * a.gamma_stride, gamma);
        sg = a.sg; dt = a.dt;
    }
    CGP_DEV void predict(int lane, double* lds, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& mp, Sym<D>& Pp) const {
        mp = mf; Pp = Pf;
        rk4_m_cov<D>([&](const Vec<D>& m, const Sym<D>& P, Vec<D>& dm, Sym<D>& dP) {
            cd_sgp_common<SM, WAVE>(model, sg, lane, lds, gamma, m, P, dm, dP);
        }, mp, Pp, dt);
    }
};

// ============================================================================================== SMOOTHER STEPS
// step(lane, lds, mf, Pf, ms, Ps): (ms, Ps) at k+1 -> (ms, Ps) at k, given the filtering result (mf, Pf) at k.

// eks (filters_smoothers.py:338-346) and, with a linear model, rts (:208-216)
template <class DM, bool WAVE_> struct EksStep {
    static constexpr bool USES_SIGMA = false;
    static constexpr int D = DM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = false;
    DM model;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); model.uniform = WAVE; model.wide = WAVE; }
    CGP_DEV void step(int, double*, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& ms, Sym<D>& Ps) const {
        Vec<D> mp; Sym<D> Pp; Mat<D> DT, G;
        model.propagate(mf, Pf, mp, DT, Pp);          // DT = J Pf
        smoother_gain<D>(DT, Pp, G);
        smoother_apply<D>(G, mf, Pf, mp, Pp, ms, Ps);
    }
};

// sgp_smoother (filters_smoothers.py:520-528)
template <class DM, bool WAVE_, bool COLL = false> struct SgpsStep {
    static constexpr bool USES_SIGMA = true;
    static constexpr int D = DM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = WAVE_;
    DM model; SigmaSet sg;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); sg = a.sg; model.wide = WAVE; }
    CGP_DEV void step(int lane, double* lds, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& ms, Sym<D>& Ps) const {
        Vec<D> mp; Sym<D> Pp; Mat<D> DT, G;
        sgp_prediction<DM, WAVE, true, WAVE, COLL>(model, sg, lane, lds, mf, Pf, mp, Pp, DT);
        smoother_gain<D>(DT, Pp, G);
        smoother_apply<D>(G, mf, Pf, mp, Pp, ms, Ps);
    }
};

// Pf^{-1} gamma, constant over the four RK4 stages of a backward step (the reference recomputes it per stage:
// filters_smoothers.py:429, 617-618 -- same values, hoisted).
template <int D> CGP_DEV void pinv_gamma(const Sym<D>& Pf, const Sym<D>& gamma, Mat<D>& PG) {
    Sym<D> L; Vec<D> inv;
    cholesky<D>(Pf, L, inv);
    Mat<D> R;
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) R.a[i][j] = gamma(i, j);
    cho_solve_mat<D>(L, inv, R, PG);
}

// cd_eks (filters_smoothers.py:423-438), dt negated
template <class SM, bool WAVE_> struct CdEksStep {
    static constexpr bool USES_SIGMA = false;
    static constexpr int D = SM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = false;
    SM model; Sym<D> gamma; double dt;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) {
        model.setup(a.params + trial * a.param_stride, a.model_id);
        model.wide = WAVE;
        model.uniform = WAVE;
        load_sym<D>(a.gamma + trial * a.gamma_stride, gamma);
        dt = -a.dt;
    }
    CGP_DEV void step(int, double*, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& ms, Sym<D>& Ps) const {
        Mat<D> PG;
        pinv_gamma<D>(Pf, gamma, PG);
        rk4_m_cov<D>([&](const Vec<D>& m, const Sym<D>& P, Vec<D>& dm, Sym<D>& dP) {
            Mat<D> A, T;
            model.drift_jac(m, dm, A);
            // A = J_a + gamma Pf^{-1} = J_a + (Pf^{-1} gamma)^T ; dm = a + gamma Pf^{-1} (m - mf)
            CGP_UNROLL for (int i = 0; i < D; i++) {
                double s = dm.v[i];
                CGP_UNROLL for (int k = 0; k < D; k++) { A.a[i][k] += PG.a[k][i]; s = fma(PG.a[k][i], m.v[k] - mf.v[k], s); }
                dm.v[i] = s;
            }
            mul_dense_sym<D>(A, P, T);
            sym_from_sum<D>(T, gamma, -1.0, dP);          // A P + P A^T - gamma
        }, ms, Ps, dt);
    }
};

// cd_sgp_smoother (filters_smoothers.py:611-629)
template <class SM, bool WAVE_> struct CdSgpsStep {
    static constexpr bool USES_SIGMA = true;
    static constexpr int D = SM::D; static constexpr bool WAVE = WAVE_; static constexpr bool USES_LDS = WAVE_;
    SM model; Sym<D> gamma; SigmaSet sg; double dt;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) {
        model.setup(a.params + trial * a.param_stride, a.model_id);
        model.wide = WAVE;
        load_sym<D>(a.gamma + trial * a.gamma_stride, gamma);
        sg = a.sg; dt = -a.dt;
    }
    CGP_DEV void step(int lane, double* lds, const Vec<D>& mf, const Sym<D>& Pf, Vec<D>& ms, Sym<D>& Ps) const {
        Mat<D> PG;
        pinv_gamma<D>(Pf, gamma, PG);
        rk4_m_cov<D>([&](const Vec<D>& m, const Sym<D>& P, Vec<D>& dm, Sym<D>& dP) {
            cd_sgp_common<SM, WAVE>(model, sg, lane, lds, gamma, m, P, dm, dP);       // (_m, _P), _P includes + gamma
            Mat<D> GT, T;
            CGP_UNROLL for (int i = 0; i < D; i++) {
                double s = dm.v[i];
                CGP_UNROLL for (int k = 0; k < D; k++) { GT.a[i][k] = PG.a[k][i]; s = fma(PG.a[k][i], m.v[k] - mf.v[k], s); }
                dm.v[i] = s;                                                           // _m + G^T (m - mf)
            }
            mul_dense_sym<D>(GT, P, T);                                                // G^T P
            CGP_UNROLL for (int i = 0; i < D; i++)
                CGP_UNROLL for (int j = 0; j <= i; j++) dP(i, j) = (dP(i, j) + (T.a[i][j] + T.a[j][i])) - 2.0 * gamma(i, j);
        }, ms, Ps, dt);
    }
};

// ============================================================================================== TIME-PARALLEL SMOOTHER
// The discrete smoothers (rts / eks / sgp_smoother) are affine in the carry once the filtering results are known:
//     ms_k = G_k ms_{k+1} + c_k,            c_k = mf_k - G_k mp_k
//     Ps_k = G_k Ps_{k+1} G_k^T + C_k,      C_k = Pf_k - G_k Pp_k G_k^T = Pf_k - (G_k DT_k)^T     (G_k Pp_k = DT_k^T)
// with G_k, mp_k, Pp_k, DT_k functions of (mf_k, Pf_k) only (filters_smoothers.py:71-85, 212-215, 342-345, 524-527).
// So a wavefront takes 64 consecutive time steps of ONE trial, one per lane: every lane computes its own element
// (G, c, C) -- the expensive part: model, Jacobian / sigma fan, Cholesky, solves -- then a 6-round suffix scan over
// the lanes composes the affine maps, and each lane applies its composed map to the carry coming from later times.
template <int D> struct Affine {
    Mat<D> G; Vec<D> c; Sym<D> C;
};

template <int D> CGP_DEV void affine_identity(Affine<D>& e) {
    CGP_UNROLL for (int i = 0; i < D; i++) {
        e.c.v[i] = 0.0;
        CGP_UNROLL for (int j = 0; j < D; j++) e.G.a[i][j] = (i == j) ? 1.0 : 0.0;
    }
    CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) e.C.a[i] = 0.0;
}

// a <- a o b : first b (the later time steps), then a.  Written row by row so that only one D-vector of the
// intermediate product G_a C_b is live at a time and the new G overwrites the old one row by row: for d = 8 an affine
// map is 108 doubles and the register file holds 256 of them.
template <int D> CGP_DEV void affine_compose(Affine<D>& a, const Affine<D>& b) {
    // c <- G_a c_b + c_a
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = a.c.v[i];
        CGP_UNROLL for (int k = 0; k < D; k++) s = fma(a.G.a[i][k], b.c.v[k], s);
        a.c.v[i] = s;
    }
    // C <- G_a C_b G_a^T + C_a   (row i of T = G_a C_b, then its products with rows j <= i of G_a)
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double t[D];
        CGP_UNROLL for (int j = 0; j < D; j++) {
            double s = a.G.a[i][0] * b.C(0, j);
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(a.G.a[i][k], b.C(k, j), s);
            t[j] = s;
        }
        CGP_UNROLL for (int j = 0; j <= i; j++) {
            double s = t[0] * a.G.a[j][0];
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(t[k], a.G.a[j][k], s);
            a.C(i, j) = s + a.C(i, j);
        }
    }
    // G <- G_a G_b, row by row in place
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double r[D];
        CGP_UNROLL for (int j = 0; j < D; j++) {
            double s = a.G.a[i][0] * b.G.a[0][j];
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(a.G.a[i][k], b.G.a[k][j], s);
            r[j] = s;
        }
        CGP_UNROLL for (int j = 0; j < D; j++) a.G.a[i][j] = r[j];
    }
}

CGP_DEV double shfl_down_f64(double x, int delta) { return __shfl_down(x, delta, 64); }

template <int D> CGP_DEV void affine_shfl_down(const Affine<D>& e, int delta, Affine<D>& o) {
    CGP_UNROLL for (int i = 0; i < D; i++) {
        o.c.v[i] = shfl_down_f64(e.c.v[i], delta);
        CGP_UNROLL for (int j = 0; j < D; j++) o.G.a[i][j] = shfl_down_f64(e.G.a[i][j], delta);
    }
    CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) o.C.a[i] = shfl_down_f64(e.C.a[i], delta);
}

// a <- a o (the map held by lane + delta), fetching that map piecewise so that it is never resident as a whole:
// first its c and C (D + D(D+1)/2 doubles), then, once the new c and C are done, its G.  `active` lanes compose, the
// others only take part in the shuffles.  The scheduling barriers keep the compiler from hoisting the second batch of
// shuffles above the first phase (which would put both maps in registers at once: 216 doubles at d = 8).
template <int D> CGP_DEV void affine_compose_from_lane(Affine<D>& a, int delta, bool active) {
    {
        Vec<D> bc; Sym<D> bC;
        CGP_UNROLL for (int i = 0; i < D; i++) bc.v[i] = shfl_down_f64(a.c.v[i], delta);
        CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) bC.a[i] = shfl_down_f64(a.C.a[i], delta);
        if (active) {
            CGP_UNROLL for (int i = 0; i < D; i++) {
                double s = a.c.v[i];
                CGP_UNROLL for (int k = 0; k < D; k++) s = fma(a.G.a[i][k], bc.v[k], s);
                a.c.v[i] = s;
            }
            CGP_UNROLL for (int i = 0; i < D; i++) {
                double t[D];
                CGP_UNROLL for (int j = 0; j < D; j++) {
                    double s = a.G.a[i][0] * bC(0, j);
                    CGP_UNROLL for (int k = 1; k < D; k++) s = fma(a.G.a[i][k], bC(k, j), s);
                    t[j] = s;
                }
                CGP_UNROLL for (int j = 0; j <= i; j++) {
                    double s = t[0] * a.G.a[j][0];
                    CGP_UNROLL for (int k = 1; k < D; k++) s = fma(t[k], a.G.a[j][k], s);
                    a.C(i, j) = s + a.C(i, j);
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        Mat<D> bG;
        CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) bG.a[i][j] = shfl_down_f64(a.G.a[i][j], delta);
        if (active) {
            CGP_UNROLL for (int i = 0; i < D; i++) {
                double r[D];
                CGP_UNROLL for (int j = 0; j < D; j++) {
                    double s = a.G.a[i][0] * bG.a[0][j];
                    CGP_UNROLL for (int k = 1; k < D; k++) s = fma(a.G.a[i][k], bG.a[k][j], s);
                    r[j] = s;
                }
                CGP_UNROLL for (int j = 0; j < D; j++) a.G.a[i][j] = r[j];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The element of one time step from its (mp, Pp, DT).  Row c of G = (Pp^{-1} DT)^T is the solution for the right-hand
// side DT[:, c], so the gain is produced row by row with no transposed copy; C = Pf - G Pp G^T is then formed row by
// row from Pp (DT is dead after the solves).
template <int D>
CGP_DEV void affine_from_prediction(const Vec<D>& mf, const Sym<D>& Pf, const Vec<D>& mp, const Sym<D>& Pp, const Mat<D>& DT,
                                    Affine<D>& e) {
    {
        Sym<D> L; Vec<D> inv;
        cholesky<D>(Pp, L, inv);
        CGP_UNROLL for (int c = 0; c < D; c++) {
            Vec<D> col;
            CGP_UNROLL for (int i = 0; i < D; i++) col.v[i] = DT.a[i][c];
            cho_solve_vec<D>(L, inv, col);
            CGP_UNROLL for (int i = 0; i < D; i++) e.G.a[c][i] = col.v[i];
        }
    }
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = mf.v[i];
        CGP_UNROLL for (int k = 0; k < D; k++) s = fma(-e.G.a[i][k], mp.v[k], s);
        e.c.v[i] = s;
    }
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double t[D];
        CGP_UNROLL for (int j = 0; j < D; j++) {
            double s = e.G.a[i][0] * Pp(0, j);
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(e.G.a[i][k], Pp(k, j), s);
            t[j] = s;
        }
        CGP_UNROLL for (int j = 0; j <= i; j++) {
            double s = Pf(i, j);
            CGP_UNROLL for (int k = 0; k < D; k++) s = fma(-t[k], e.G.a[j][k], s);
            e.C(i, j) = s;
        }
    }
}

// The gain alone, G = (Pp^{-1} DT)^T row by row (row c = solution for the right-hand side DT[:, c]): what the cooperative
// d >= 5 smoother keeps per step, next to (mp, Pp) -- it applies the reference's own form ms = mf + G (ms' - mp),
// Ps = Pf + G (Ps' - Pp) G^T (filters_smoothers.py:83-84) on the matrix cores and never forms c and C.
template <int D>
CGP_DEV void gain_from_prediction(const Sym<D>& Pp, const Mat<D>& DT, Mat<D>& G) {
    Sym<D> L; Vec<D> inv;
    cholesky<D>(Pp, L, inv);
    CGP_UNROLL for (int c = 0; c < D; c++) {
        Vec<D> col;
        CGP_UNROLL for (int i = 0; i < D; i++) col.v[i] = DT.a[i][c];
        cho_solve_vec<D>(L, inv, col);
        CGP_UNROLL for (int i = 0; i < D; i++) G.a[c][i] = col.v[i];
    }
}

// (c, C) of a step's affine map from its gain: c = mf - G mp, C = Pf - G Pp G^T.  With G = (Pp^{-1} D^T)^T the product G Pp is
// the cross-covariance D itself, which the prediction has just produced (DT = D^T), so C = Pf - G D^T costs one triangular
// product (d^2 (d + 1) / 2 multiply-adds) instead of two full ones -- the same quantity up to the rounding of the solves.
// For d > 4 that would keep the d x d matrix D^T alive across the factorisation and the solves (measured at d = 8: spills,
// 3.9 -> 4.6 ms for the EKS of BASELINE C5's model), so there C is formed from Pp row by row as affine_from_prediction does.
template <int D>
CGP_DEV void map_from_gain(const Vec<D>& mf, const Sym<D>& Pf, const Vec<D>& mp, const Sym<D>& Pp, const Mat<D>& DT, const Mat<D>& G, Vec<D>& c, Sym<D>& C) {
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = mf.v[i];
        CGP_UNROLL for (int k = 0; k < D; k++) s = fma(-G.a[i][k], mp.v[k], s);
        c.v[i] = s;
    }
    if constexpr (D <= 4) {
        CGP_UNROLL for (int i = 0; i < D; i++)
            CGP_UNROLL for (int j = 0; j <= i; j++) {
                double s = Pf(i, j);
                CGP_UNROLL for (int k = 0; k < D; k++) s = fma(-G.a[i][k], DT.a[k][j], s);
                C(i, j) = s;
            }
    } else {
        CGP_UNROLL for (int i = 0; i < D; i++) {
            double t[D];
            CGP_UNROLL for (int j = 0; j < D; j++) {
                double s = G.a[i][0] * Pp(0, j);
                CGP_UNROLL for (int k = 1; k < D; k++) s = fma(G.a[i][k], Pp(k, j), s);
                t[j] = s;
            }
            CGP_UNROLL for (int j = 0; j <= i; j++) {
                double s = Pf(i, j);
                CGP_UNROLL for (int k = 0; k < D; k++) s = fma(-t[k], G.a[j][k], s);
                C(i, j) = s;
            }
        }
    }
}

// Applies a map to a state, row by row.
template <int D>
CGP_DEV void affine_apply(const Affine<D>& e, const Vec<D>& ms, const Sym<D>& Ps, Vec<D>& xm, Sym<D>& xP) {
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = e.c.v[i];
        CGP_UNROLL for (int k = 0; k < D; k++) s = fma(e.G.a[i][k], ms.v[k], s);
        xm.v[i] = s;
    }
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double t[D];
        CGP_UNROLL for (int j = 0; j < D; j++) {
            double s = e.G.a[i][0] * Ps(0, j);
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(e.G.a[i][k], Ps(k, j), s);
            t[j] = s;
        }
        CGP_UNROLL for (int j = 0; j <= i; j++) {
            double s = t[0] * e.G.a[j][0];
            CGP_UNROLL for (int k = 1; k < D; k++) s = fma(t[k], e.G.a[j][k], s);
            xP(i, j) = s + e.C(i, j);
        }
    }
}

// Per-lane element producers (one lane = one time step, so the fan of the sigma-point variant is a serial loop).
template <class DM> struct EksElement {
    static constexpr bool USES_SIGMA = false;
    static constexpr int D = DM::D;
    DM model;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); model.wide = true; }
    CGP_DEV void element(const Vec<D>& mf, const Sym<D>& Pf, Affine<D>& e) const {
        Vec<D> mp; Sym<D> Pp; Mat<D> DT;
        model.propagate(mf, Pf, mp, DT, Pp);
        affine_from_prediction<D>(mf, Pf, mp, Pp, DT, e);
    }
    CGP_DEV void gain(const Vec<D>& mf, const Sym<D>& Pf, Mat<D>& G, Vec<D>& mp, Sym<D>& Pp) const {
        Mat<D> DT;
        model.propagate(mf, Pf, mp, DT, Pp);
        gain_from_prediction<D>(Pp, DT, G);
    }
    // the step's affine map (G, c, C) for the cooperative walks (cgp_walk4.hpp, cgp_coop8.hpp)
    CGP_DEV void map(const Vec<D>& mf, const Sym<D>& Pf, Mat<D>& G, Vec<D>& c, Sym<D>& C) const {
        Mat<D> DT; Vec<D> mp; Sym<D> Pp;
        model.propagate(mf, Pf, mp, DT, Pp);
        gain_from_prediction<D>(Pp, DT, G);
        map_from_gain<D>(mf, Pf, mp, Pp, DT, G, c, C);
    }
    CGP_DEV void map_spec(const Vec<D>& mf, const Sym<D>& Pf, Mat<D>& G, Vec<D>& c, Sym<D>& C, bool& ok) const {
        Mat<D> DT; Vec<D> mp; Sym<D> Pp;
        if constexpr (HAS_SPEC) model.propagate_spec(mf, Pf, mp, DT, Pp, ok);
        else { model.propagate(mf, Pf, mp, DT, Pp); ok = true; }
        gain_from_prediction<D>(Pp, DT, G);
        map_from_gain<D>(mf, Pf, mp, Pp, DT, G, c, C);
    }
    // gain() as straight-line code where the model has a branch-free form (cgp_models.hpp:propagate_spec): ok = false where
    // gain() would have taken a regime fallback
    static constexpr bool HAS_SPEC = std::is_same<DM, HarmonicLCD<1>>::value;
    CGP_DEV void gain_spec(const Vec<D>& mf, const Sym<D>& Pf, Mat<D>& G, Vec<D>& mp, Sym<D>& Pp, bool& ok) const {
        Mat<D> DT;
        if constexpr (HAS_SPEC) model.propagate_spec(mf, Pf, mp, DT, Pp, ok);
        else { model.propagate(mf, Pf, mp, DT, Pp); ok = true; }
        gain_from_prediction<D>(Pp, DT, G);
    }
};
template <class DM, bool COLL = false> struct SgpsElement {
    static constexpr bool HAS_SPEC = false;
    static constexpr bool USES_SIGMA = true;
    static constexpr int D = DM::D;
    DM model; SigmaSet sg;
    CGP_DEV void setup(const ModelArgs& a, int64_t trial) { model.setup(a.params + trial * a.param_stride, a.dt, a.model_id); sg = a.sg; model.wide = true; }
    CGP_DEV void element(const Vec<D>& mf, const Sym<D>& Pf, Affine<D>& e) const {
        Vec<D> mp; Sym<D> Pp; Mat<D> DT;
        sgp_prediction<DM, false, true, true, COLL>(model, sg, 0, nullptr, mf, Pf, mp, Pp, DT);
        affine_from_prediction<D>(mf, Pf, mp, Pp, DT, e);
    }
    CGP_DEV void gain(const Vec<D>& mf, const Sym<D>& Pf, Mat<D>& G, Vec<D>& mp, Sym<D>& Pp) const {
        Mat<D> DT;
        sgp_prediction<DM, false, true, true, COLL>(model, sg, 0, nullptr, mf, Pf, mp, Pp, DT);
        gain_from_prediction<D>(Pp, DT, G);
    }
    CGP_DEV void map(const Vec<D>& mf, const Sym<D>& Pf, Mat<D>& G, Vec<D>& c, Sym<D>& C) const {
        Mat<D> DT; Vec<D> mp; Sym<D> Pp;
        sgp_prediction<DM, false, true, true, COLL>(model, sg, 0, nullptr, mf, Pf, mp, Pp, DT);
        gain_from_prediction<D>(Pp, DT, G);
        map_from_gain<D>(mf, Pf, mp, Pp, DT, G, c, C);
    }
};

}  // namespace cgp
