// cgp_kernels.hpp -- the scan kernels: a forward pass over T measurements (filters) and a reverse pass over the
// filtering results (smoothers), batched over independent trials.  Layouts are the reference's batch-major
// (B, T, ...) arrays (jax.vmap over ys: tetralith/jobs/crlb_ekf.py:68-72).
//
// Launch shapes (block = one 64-lane wavefront in both):
//   WAVE  = true  : grid = B.   One wavefront per trial; ys is fetched 64 steps at a time with one coalesced 512-B
//                   load and handed out by v_readlane; lane 0 streams the 8(d + d^2 + 1) output bytes of each step
//                   (consecutive steps of a trial are contiguous in HBM, so L2 write-combines them into full lines).
//   WAVE  = false : grid = ceil(B / 64).  One lane per trial (large batches).
#pragma once
#include "cgp_steps.hpp"
#ifndef __HIPCC_RTC__          // the context is host code
#include "cgp_ctx.hpp"
#else
struct cgp_ctx;
#endif
#include "../../include/chirpgp_hip.h"

namespace cgp {

struct FilterIO {
    const double* __restrict__ H;  int64_t H_stride;
    const double* __restrict__ Xi; int64_t Xi_stride;
    const double* __restrict__ m0; int64_t m0_stride;
    const double* __restrict__ P0; int64_t P0_stride;
    const double* __restrict__ ys;
    int64_t ys_stride;                       // doubles between records (T for the dense [B][T] layout; 0 = one shared record)
    int64_t ys_repeat;                       // >= 1: consecutive trials served by one record
    const int32_t* __restrict__ ys_index;    // optional record number per group of ys_repeat trials
    int64_t B, T;
    double* __restrict__ mfs;
    double* __restrict__ Pfs;
    double* __restrict__ nll;
    uint32_t flags;
    unsigned long long* __restrict__ counters = nullptr;      // cgp_debug_set(CGP_DBG_COUNT_REGIMES): regime counters of the context, or NULL
    // Time-split filters with burn-in (cgp_filter_time_split; round 4): `segs` wavefronts per trial.  Wavefront (b, s) filters the
    // steps [s seg_len - burn_in, (s + 1) seg_len) of trial b from (m0, P0), writes rows from s seg_len on, and leaves in
    // seg_state[(b segs + s) seg_stride ..] its state at the junction (after its last burn-in step), its state after its last step
    // and its NLL total: (m, P) | (m, P) | nll.  seg_len and burn_in are multiples of 64 (the kernels' chunk).  segs <= 1: off.
    int segs = 1;
    int64_t seg_len = 0, burn_in = 0;
    double* __restrict__ seg_state = nullptr;
    int seg_stride = 0;
    // The measurement record of a trial (include/chirpgp_hip.h, cgp_filter): trial b reads record ys_index[b / ys_repeat]
    // (b / ys_repeat without an index) -- a parameter sweep or the 2 P + 1 probes of a difference gradient read ONE copy.
    __device__ __forceinline__ const double* record(int64_t trial) const {
        int64_t g = trial;
        if (ys_repeat > 1) g = (int64_t)((uint64_t)trial / (uint64_t)ys_repeat);
        if (ys_index) g = ys_index[g];
        return ys + g * ys_stride;
    }
};

// The span of time steps of workgroup v of a (possibly time-split) filter launch.
struct FilterSpan {
    int64_t trial, t_begin, t_out, t_end;      // filters [t_begin, t_end), writes rows [t_out, t_end)
    int seg;
    double* state;                             // this segment's record in FilterIO::seg_state, or NULL
};
// SPLIT = false: a kernel instantiated for whole-record launches only -- the span is (0, 0, T) at compile time and everything that
// serves the segments folds away (the sigma-point kernels measured 2 % slower with the run-time form in their default launch).
template <bool SPLIT = true>
__device__ __forceinline__ FilterSpan filter_span(const FilterIO& io, int64_t v) {
    if (!SPLIT || io.segs <= 1) return {v, 0, 0, io.T, 0, nullptr};
    const int64_t b = (int64_t)((uint64_t)v / (uint64_t)io.segs);
    const int s = (int)(v - b * io.segs);
    const int64_t t_out = (int64_t)s * io.seg_len;
    int64_t t_begin = s > 0 ? t_out - io.burn_in : 0;
    if (t_begin < 0) t_begin = 0;
    int64_t t_end = t_out + io.seg_len;
    if (t_end > io.T) t_end = io.T;
    return {b, t_begin, t_out, t_end, s, io.seg_state ? io.seg_state + (b * io.segs + s) * io.seg_stride : nullptr};
}

// Selected outputs of a smoother launch (cgp_smoother_select; SURVEY 8f-2: the step right behind the smoother in every driver of the
// reference keeps mss[:, k], Pss[:, k, k] and E[g(V)] of that marginal -- demos/ekfs_mle.py:69-77, quadratures.py:234-274):
// per (trial, step) the smoothed mean and variance of ONE state component and / or E[f(V)], V ~ N(mean, variance), by 1-D
// Gauss-Hermite -- written as [B][T] arrays by the kernel that produced them, so that a pipeline that needs nothing else (mss / Pss
// NULL) writes 8 - 24 bytes a step instead of 8 (d + d^2).
struct SmoothSel {
    int comp = -1;                          // state component; < 0: nothing selected
    int func = 0, order = 0;                // CGP_FN_* integrand and number of nodes of `expect`
    double* __restrict__ mean = nullptr;    // [B][T] or NULL
    double* __restrict__ var = nullptr;     // [B][T] or NULL
    double* __restrict__ expect = nullptr;  // [B][T] or NULL
    const double* __restrict__ xi = nullptr;   // [order] nodes, the reference's scaling (sqrt(2) x the Hermite roots)
    const double* __restrict__ w = nullptr;    // [order] weights, normalised
};
// E[f(m + s Z)] over the rule's nodes: the loop of quadratures.py:218-231 for d = 1.  The softplus integrand (the reference's g,
// models.py:50: the one every driver uses) in three tiers, chosen per LANE so that an element's value does not depend on its neighbours
// in the wavefront (a tier is evaluated by the wavefront only if one of its lanes needs it):
//   1. every node x >= 6 (a frequency above 6 Hz with its whole Gauss-Hermite fan):  log(e^x + 1) = x + log1p(t), t = e^-x <= 2.5e-3, the
//      exponential by the degree-7 near-minimax polynomial of the speculative EKF step (5.5e-11 of t, i.e. 2e-14 of the softplus) and
//      log1p by six terms of its series (truncation t^7 / 7 < 1e-19): ~25 instructions a node;
//   2. any |x| < 700: the branch-free full-accuracy softplus of cgp_fastmath.hpp (softplus_pair_any), ~80 a node;
//   3. otherwise (overflow, inf, NaN) and for the other integrands: the library functions, as cgp_gaussian_expectation_fn evaluates them.
CGP_DEV double exp_neg_lean_lit(double x) {
    const double nx = -x;
    const double k = __builtin_rint(nx * kLog2e);
    double r = fma(-k, kLn2Hi, nx);
    r = fma(-k, kLn2Lo, r);
    const double r2 = r * r;
    const double a0 = fma(kExpLean[1], r, kExpLean[0]), a1 = fma(kExpLean[3], r, kExpLean[2]);
    const double a2 = fma(kExpLean[5], r, kExpLean[4]), a3 = fma(kExpLean[7], r, kExpLean[6]);
    const double r4 = r2 * r2;
    return __builtin_amdgcn_ldexp(fma(fma(a3, r2, a2), r4, fma(a1, r2, a0)), (int)k);
}
CGP_DEV double gh_expectation(int func, double m, double s, const double* __restrict__ xi, const double* __restrict__ w, int order) {
    double fast = 0.0;
    bool have = false;
    if (func == CGP_FN_SOFTPLUS) {
        // the fan's extreme nodes decide the lane's tier before anything is evaluated (NaN: neither)
        double reach = 0.0;
        for (int p = 0; p < order; p++) reach = fmax(reach, fabs(xi[p]));
        reach *= fabs(s);
        const bool tier1 = (m - reach >= 6.0) && (m + reach < 700.0);
        const bool tier2 = !tier1 && (m - reach > -700.0) && (m + reach < 700.0);
        if (__builtin_amdgcn_ballot_w64(tier1) != 0) {
            double acc = 0.0;
            for (int p = 0; p < order; p++) {
                const double x = fma(s, xi[p], m);
                const double t = exp_neg_lean_lit(x);
                double q = fma(t, -1.0 / 6.0, 1.0 / 5.0);                      // log1p(t) / t = 1 - t/2 + t^2/3 - t^3/4 + t^4/5 - t^5/6
                q = fma(q, t, -1.0 / 4.0); q = fma(q, t, 1.0 / 3.0); q = fma(q, t, -0.5); q = fma(q, t, 1.0);
                acc = fma(w[p], fma(q, t, x), acc);
            }
            fast = acc;
        }
        if (__builtin_amdgcn_ballot_w64(tier2) != 0) {
            double acc = 0.0;
            for (int p = 0; p < order; p++) {
                const double x = fma(s, xi[p], m);
                double sp, dsp; bool okp;
                softplus_pair_any(x, sp, dsp, okp);
                acc = fma(w[p], sp, acc);
            }
            fast = tier2 ? acc : fast;
        }
        have = tier1 || tier2;
        if (__builtin_amdgcn_ballot_w64(!have) == 0) return fast;
    }
    double acc = 0.0;
    for (int p = 0; p < order; p++) {
        const double x = fma(s, xi[p], m);
        double f;
        if (func == CGP_FN_SOFTPLUS) f = log(exp(x) + 1.0);                 // models.py:50, the naive form as is
        else if (func == CGP_FN_EXP) f = exp(x);
        else if (func == CGP_FN_SQUARE) f = x * x;
        else f = x;
        acc = fma(w[p], f, acc);
    }
    return have ? fast : acc;
}
// The same as a CALL: the epilogue of the wave-per-trial smoothers runs once per tile, outside their walks -- a call keeps its three
// loops out of kernels that are scheduled around matrix instructions (inlined, hipcc 7.2's "Rewrite AGPR-Copy-MFMA" pass crashed on
// coop8_split_kernel<SgpsElement<LinearDisc<8>>, kWalkApply, true> under -amdgpu-mfma-vgpr-form).
__device__ __attribute__((noinline)) double gh_expectation_call(int func, double m, double s, const double* __restrict__ xi, const double* __restrict__ w, int order) {
    return gh_expectation(func, m, s, xi, w, order);
}
// The rule's nodes and weights staged in LDS once per workgroup (rule[0 .. order) nodes, rule[32 .. 32 + order) weights; order <= 32,
// checked by the C-ABI): read from global memory inside the node loop, every iteration waited for two dependent loads -- the epilogue of
// the lane kernel cost 11 ms per 262 144 x 500 instead of 3.  The caller fences (wave_lds_fence / __syncthreads) before the first use.
constexpr int kGhMaxOrder = 32;
CGP_DEV void sel_stage_rule(const SmoothSel& s, double* rule, int tid) {
    if (s.expect && tid < s.order) { rule[tid] = s.xi[tid]; rule[kGhMaxOrder + tid] = s.w[tid]; }
}
// one (trial, step) of the selected outputs; idx = trial * T + step
CGP_DEV void sel_write(const SmoothSel& s, const double* rule, int64_t idx, double m, double v) {
    if (s.mean) s.mean[idx] = m;
    if (s.var) s.var[idx] = v;
    if (s.expect) s.expect[idx] = gh_expectation_call(s.func, m, sqrt(v), rule, rule + kGhMaxOrder, s.order);
}

struct SmootherIO {
    const double* __restrict__ mfs;
    const double* __restrict__ Pfs;
    int64_t B, T;
    double* __restrict__ mss;
    double* __restrict__ Pss;
    uint32_t flags;
    // time-split discrete smoothers (cgp_walk4.hpp, cgp_coop8.hpp): `segs` wavefronts per trial, each walking `tiles_per_seg`
    // 64-step tiles; the segments' composed maps (A, c, C) live in `ws` between the two passes.  segs <= 1: one wave per trial.
    int segs = 1;                 // as handed to a launcher: 0 = choose from (B, T, num_cus); 1 = off; > 1 = this many at most
    int min_tiles = 4;            // per segment (1 when the caller forces the time-split form)
    int num_cus = 256;
    int tiles_per_seg = 0;
    double* __restrict__ ws = nullptr;
    cgp_ctx* host_ctx = nullptr;  // host side only: the context whose per-stream workspace (cgp::ctx_workspace) serves `ws`
    // Time-split with burn-in of the continuous-discrete smoothers (cgp_smoother_time_split; round 6): their backward ODE is not affine in the
    // carry, so a segment cannot be composed exactly -- but the recursion FORGETS its terminal condition like the filters forget their initial
    // one: segment s (0 = the last in time) starts `burn_chunks` 64-step chunks LATER than its piece from the filtering row there, writes
    // nothing until its piece begins, and leaves its state at the junction in junction[(trial * bsegs + s) * 20] (m 4 | P 16) for the
    // fix-up pass to compare with the row the segment before it wrote.  bsegs <= 1: off.
    int bsegs = 1, chunks_per_bseg = 0, burn_chunks = 0;
    double* __restrict__ junction = nullptr;
    SmoothSel sel;                // cgp_smoother_select: selected outputs (mss / Pss may then be NULL); comp < 0: off
    int lane_buffers = 0;         // host side: cgp_debug_set(CGP_DBG_LANE_BUFFERS) -- 3 = the large-batch smoothers request their rows two steps ahead, else one
};

// Dynamic LDS (sized by the launch): the staged sigma-point set.  The static LDS in front of it (the 17 152-byte
// reduction buffer) is a multiple of 16 bytes, so the base stays 16-byte aligned (cdna_hip_programming.md G17).
CGP_DEV double* dyn_lds() {
    extern __shared__ double cgp_dyn_lds[];
    return cgp_dyn_lds;
}
inline size_t sigma_lds_bytes(const ModelArgs& ma, int d) {
    return ma.sg.xi ? SigmaSet::stage_bytes(ma.sg.s, d, ma.sg.n_groups, ma.sg.group_start != nullptr) : 0;
}

// Linear scalar measurement y = H x + noise (every filter but ekf_for_kpt).
template <int D> struct LinearMeasurement {
    static constexpr bool LINEAR = true;
    CGP_DEV static void update(const Vec<D>& mp, const Sym<D>& Pp, const Vec<D>& H, double Xi, double y, Vec<D>& mf, Sym<D>& Pf,
                               double& S, double& innov) {
        scalar_update<D>(mp, Pp, H, Xi, y, false, 0.0, mf, Pf, S, innov);
    }
};
// ekf_for_kpt (filters_smoothers.py:298-311): H = grad h(mp), pred = h(mp).
template <int NH, bool UNIFORM = false> struct KptUpdate {
    static constexpr int D = NH + 2;
    static constexpr bool LINEAR = false;
    CGP_DEV static void update(const Vec<D>& mp, const Sym<D>& Pp, const Vec<D>&, double Xi, double y, Vec<D>& mf, Sym<D>& Pf,
                               double& S, double& innov) {
        Vec<D> H;
        const double pred = KptMeasurement<NH, UNIFORM>::eval(mp, H);
        scalar_update<D>(mp, Pp, H, Xi, y, true, pred, mf, Pf, S, innov);
    }
};

// ---- one-lane-per-trial shape: coalesced rows through LDS -----------------------------------------------------------
// In this shape lane l owns trial (block * 64 + l), and what it reads or writes per step is one ROW of N doubles that is
// contiguous for the trial but T * N doubles away from the neighbouring lane's row.  Issued directly, every 16-byte
// store instruction touches 64 different 128-byte lines (measured: 2.5 TB/s filter, 3.7 TB/s smoother at B = 10^6).
// Instead the 64 rows are transposed through LDS so that consecutive lanes access consecutive 16-byte chunks of the same
// trial's row: each wave instruction then covers whole lines (N = 16: 8 trials x 128 B).  Rows are padded by 16 bytes
// in LDS (pitch N + 2 doubles), which keeps both the row writes and the chunk reads conflict-free.
template <int N> struct RowTile {
    static constexpr int C = N / 2;            // 16-byte chunks per row
    static constexpr int PITCH = N + 2;        // doubles
    static constexpr int DOUBLES = 64 * PITCH;
};

template <int N>
CGP_DEV void block_store_rows(double* tile, int lane, const double (&row)[N], double* __restrict__ out_block,
                              int64_t trial_stride, int nvalid) {
    static_assert(N % 2 == 0, "rows of an even number of doubles");
    using RT = RowTile<N>;
    CGP_UNROLL for (int c = 0; c < RT::C; c++)
        *reinterpret_cast<double2*>(tile + lane * RT::PITCH + 2 * c) = make_double2(row[2 * c], row[2 * c + 1]);
    wave_lds_fence();
    CGP_UNROLL for (int k = 0; k < RT::C; k++) {
        const int g = k * 64 + lane, tr = g / RT::C, ch = g % RT::C;
        const double2 v = *reinterpret_cast<const double2*>(tile + tr * RT::PITCH + 2 * ch);
        if (tr < nvalid) *reinterpret_cast<double2*>(out_block + tr * trial_stride + 2 * ch) = v;
    }
    wave_lds_fence();
}

template <int N>
CGP_DEV void block_load_rows(double* tile, int lane, const double* __restrict__ in_block, int64_t trial_stride, int nvalid,
                             double (&row)[N]) {
    static_assert(N % 2 == 0, "rows of an even number of doubles");
    using RT = RowTile<N>;
    CGP_UNROLL for (int k = 0; k < RT::C; k++) {
        const int g = k * 64 + lane, tr = g / RT::C, ch = g % RT::C;
        const int trc = tr < nvalid ? tr : nvalid - 1;
        const double2 v = *reinterpret_cast<const double2*>(in_block + trc * trial_stride + 2 * ch);
        *reinterpret_cast<double2*>(tile + tr * RT::PITCH + 2 * ch) = v;
    }
    wave_lds_fence();
    CGP_UNROLL for (int c = 0; c < RT::C; c++) {
        const double2 v = *reinterpret_cast<const double2*>(tile + lane * RT::PITCH + 2 * c);
        row[2 * c] = v.x; row[2 * c + 1] = v.y;
    }
    wave_lds_fence();
}

template <int D> CGP_DEV void sym_to_row(const Sym<D>& P, double (&row)[D * D]) {
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) row[i * D + j] = P(i, j);
}
template <int D> CGP_DEV void row_to_sym(const double (&row)[D * D], Sym<D>& P) {
    CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j <= i; j++) P(i, j) = row[i * D + j];
}

// ---- one-wave-per-trial shape: a replicated row leaves through LDS -------------------------------------------------------
// Every lane holds the same (mf, Pf); written by lane 0 alone, the d + d^2 doubles of a step are d (d + 1) / 2 16-byte
// store instructions with one active lane each -- 36 for d = 8, 0.9 us of a 4.7 us step (ablation on the C5 filter).
// Instead lane 0 parks the row in LDS (ds_write is cheap with one lane) and lanes 0 .. N/2 - 1 store one 16-byte piece
// each: a single coalesced store instruction per array.
template <int D> struct WaveRow { static constexpr int DOUBLES = D + D * D + ((D + D * D) & 1); };
template <int D>
CGP_DEV void wave_store_step(double* park, int lane, const Vec<D>& m, const Sym<D>& P, double* __restrict__ m_out, double* __restrict__ P_out) {
    if (lane == 0) {
        if constexpr (D % 2 == 0) {
            CGP_UNROLL for (int i = 0; i < D; i += 2) *reinterpret_cast<double2*>(park + i) = make_double2(m.v[i], m.v[i + 1]);
            CGP_UNROLL for (int i = 0; i < D; i++)
                CGP_UNROLL for (int j = 0; j < D; j += 2) *reinterpret_cast<double2*>(park + D + i * D + j) = make_double2(P(i, j), P(i, j + 1));
        } else {
            CGP_UNROLL for (int i = 0; i < D; i++) park[i] = m.v[i];
            CGP_UNROLL for (int i = 0; i < D; i++) CGP_UNROLL for (int j = 0; j < D; j++) park[D + i * D + j] = P(i, j);
        }
    }
    wave_lds_fence();
    if constexpr (D % 2 == 0) {
        CGP_UNROLL for (int base = 0; base < D * D / 2; base += 64) {
            const int c = base + lane;
            if (P_out && c < D * D / 2) *reinterpret_cast<double2*>(P_out + 2 * c) = *reinterpret_cast<const double2*>(park + D + 2 * c);
        }
        if (m_out && lane < D / 2) *reinterpret_cast<double2*>(m_out + 2 * lane) = *reinterpret_cast<const double2*>(park + 2 * lane);
    } else {
        CGP_UNROLL for (int base = 0; base < D * D; base += 64) {
            const int c = base + lane;
            if (P_out && c < D * D) P_out[c] = park[D + c];
        }
        if (m_out && lane < D) m_out[lane] = park[lane];
    }
    wave_lds_fence();
}

// (one lane per trial, EKF at d <= 4: at least two wavefronts per SIMD -- 256 registers; left alone the staged form below takes 371
// and runs one wavefront per SIMD, which leaves its LDS round trips and memory latencies uncovered: 262 144 x 500, full outputs,
// 5.81 -> 5.58 ms, means only 3.15 -> 2.25 ms.  The sigma-point predictions spill under that cap -- 8.6 -> 13.3 ms -- and keep all
// registers: Pred::LANE_TWO_WAVES)
template <class Pred, class Meas>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((!Pred::WAVE && Pred::D <= 4 && Pred::D % 2 == 0 && Pred::LANE_TWO_WAVES) ? 2 : 1)))
filter_kernel(FilterIO io, ModelArgs ma) {
    constexpr int D = Pred::D;
    constexpr bool WAVE = Pred::WAVE;
    constexpr bool TILED = !WAVE && D % 2 == 0;      // one lane per trial: rows go through an LDS transpose
    // ... and, for d = 2 and 4, everything that is NOT a whole 128-byte line per trial and step is STAGED until it is one
    // (round 4, below): measurements and cumulative NLL in groups of 16 steps, filtered means in groups of 16 / d steps
    constexpr bool STAGED = TILED && (D == 2 || D == 4);
    constexpr int kTileDoubles = !TILED ? 1 : (STAGED && RowTile<D * D>::DOUBLES < RowTile<16>::DOUBLES ? RowTile<16>::DOUBLES : RowTile<D * D>::DOUBLES);
    __shared__ double lds[Pred::USES_LDS ? kFanLdsDoubles : 1];
    __shared__ double tile[kTileDoubles];
    __shared__ __attribute__((aligned(16))) double rowpark[WAVE ? WaveRow<D>::DOUBLES : 1];
    constexpr int kYPitch = 17;                      // 16 steps + 1: an odd pitch keeps the per-lane and the per-line accesses conflict-free
    __shared__ double ytile[STAGED ? 64 * kYPitch : 1];
    __shared__ const double* recs[STAGED ? 64 : 1];
    const int lane = threadIdx.x;
    int64_t trial = WAVE ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 64 + lane;
    const int64_t block_first = (int64_t)blockIdx.x * 64;
    const int nvalid = (io.B - block_first < 64) ? (int)(io.B - block_first) : 64;      // lane-per-trial shape only
    if constexpr (TILED) {
        // every lane takes part in the cooperative stores: lanes past the batch redo the last trial and store nothing
        if (block_first >= io.B) return;
        if (trial >= io.B) trial = io.B - 1;
    } else {
        if (trial >= io.B) return;
    }

    Pred pred;
    pred.setup(ma, trial);
    if constexpr (Pred::USES_SIGMA && WAVE) pred.sg.stage(dyn_lds(), lane, 64, D);
    Vec<D> H, mf;
    Sym<D> Pf;
    if (io.H) load_vec<D>(io.H + trial * io.H_stride, H);
    else { CGP_UNROLL for (int i = 0; i < D; i++) H.v[i] = 0.0; }
    const double Xi = io.Xi[trial * io.Xi_stride];
    load_vec<D>(io.m0 + trial * io.m0_stride, mf);
    load_sym<D>(io.P0 + trial * io.P0_stride, Pf);

    const int64_t T = io.T;
    const double* __restrict__ ys = io.record(trial);
    double* __restrict__ mfs = io.mfs ? io.mfs + trial * T * D : nullptr;
    double* __restrict__ Pfs = io.Pfs ? io.Pfs + trial * T * D * D : nullptr;
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    double* __restrict__ nll = (io.nll && !nll_final) ? io.nll + trial * T : nullptr;
    const bool writer = !WAVE || lane == 0;

    // The negative log-likelihood needs sqrt, log and a divide per step but nothing downstream depends on it, so the
    // wave-per-trial shape keeps it off the serial chain: lane (t mod 64) latches (S, innovation) of step t, and every
    // 64 steps all lanes evaluate their increment at once, prefix-sum it across the wave and store 64 cumulative
    // values with one coalesced 512-B store.
    const bool want_nll = io.nll != nullptr;
    double cum = 0.0, S_l = 1.0, innov_l = 0.0;
    if constexpr (WAVE) {
        for (int64_t t0 = 0; t0 < T; t0 += 64) {
            // 64 measurements with one coalesced 512-B load.  The empty asm consumes the loaded register here, so the
            // compiler's s_waitcnt vmcnt(0) for it sits in this outer loop and not in front of every step's v_readlane
            // (where it would also wait for the previous step's ten output stores to be acknowledged).
            double ychunk = (t0 + lane < T) ? ys[t0 + lane] : 0.0;
            asm volatile("" : "+v"(ychunk));
            const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
            for (int slot = 0; slot < nsteps; slot++) {
                const int64_t t = t0 + slot;
                const double y = readlane_f64(ychunk, slot);
                Vec<D> mp; Sym<D> Pp;
                double S, innov;
                pred.predict(lane, lds, mf, Pf, mp, Pp);
                Meas::update(mp, Pp, H, Xi, y, mf, Pf, S, innov);
                if (lane == slot) { S_l = S; innov_l = innov; }
                if constexpr (D >= 6) {
                    if (mfs || Pfs) wave_store_step<D>(rowpark, lane, mf, Pf, mfs ? mfs + t * D : nullptr, Pfs ? Pfs + t * D * D : nullptr);
                } else if (writer) {      // few enough pieces that the LDS round trip costs more than it saves
                    if (mfs) store_vec<D>(mfs + t * D, mf);
                    if (Pfs) store_sym_full<D>(Pfs + t * D * D, Pf);
                }
            }
            if (want_nll) {
                double v = (lane < nsteps) ? nll_increment(S_l, innov_l) : 0.0;
                v = wave_inclusive_scan(v) + cum;
                if (nll && lane < nsteps) nll[t0 + lane] = v;
                cum = readlane_f64(v, nsteps - 1);
            }
        }
    } else if constexpr (STAGED) {
        // One lane per trial at d = 2 / 4, whole lines only.  Per trial and step the filter reads 8 bytes of measurement and
        // writes 8 bytes of NLL and 8 d bytes of mean -- pieces of 128-byte lines that 64 lanes x several waves per SIMD keep
        // half-filled in L2 for many steps: at 262 144 x 500 (the reference's CRLB job, tetralith/jobs/crlb_ekf.py:59-79) the
        // counters showed 40.9 GB of HBM traffic for 23.1 GB of algorithmic bytes (the measurements fetched seven times over, the
        // means and the NLL written in 32-byte pieces; profiles/r04_ekf_large_*).  So:
        //   * measurements: 64 trials x 16 steps per cooperative load (every load instruction covers four whole lines), parked
        //     in LDS, one ds_read per lane and step; the step's cumulative NLL takes the consumed measurement's slot, and the
        //     tile leaves as whole lines after the 16 steps;
        //   * filtered means: held for 16 / d steps in registers (a lane-uniform select per slot) and stored as ONE 128-byte
        //     row per trial through the same LDS transpose as the covariance rows.
        constexpr int MS = 16 / D;                                   // steps of means per 128-byte line
        recs[lane] = io.record(block_first + (lane < nvalid ? lane : nvalid - 1));
        const int64_t T_lines = T - T % MS;                          // steps whose means leave as whole lines; the rest row by row
        double mh[16];
        CGP_UNROLL for (int i = 0; i < 16; i++) mh[i] = 0.0;
        for (int64_t t0 = 0; t0 < T; t0 += 16) {
            wave_lds_fence();
            _Pragma("unroll 4") for (int k = 0; k < 16; k++) {           // (four loads in flight: more only costs registers)
                const int g = k * 64 + lane, tr = g >> 4, e = g & 15;
                ytile[tr * kYPitch + e] = (t0 + e < T) ? recs[tr][t0 + e] : 0.0;
            }
            wave_lds_fence();
            const int nst = (T - t0 < 16) ? (int)(T - t0) : 16;
            for (int k = 0; k < nst; k++) {
                const int64_t t = t0 + k;
                const double y = ytile[lane * kYPitch + k];
                Vec<D> mp; Sym<D> Pp;
                double S, innov;
                pred.predict(lane, lds, mf, Pf, mp, Pp);
                Meas::update(mp, Pp, H, Xi, y, mf, Pf, S, innov);
                if (want_nll) {
                    cum += nll_increment(S, innov);
                    ytile[lane * kYPitch + k] = cum;
                }
                if (io.mfs) {
                    if (t < T_lines) {
                        const int slot = (int)(t % MS);
                        CGP_UNROLL for (int s = 0; s < MS; s++)
                            CGP_UNROLL for (int i = 0; i < D; i++) mh[s * D + i] = (slot == s) ? mf.v[i] : mh[s * D + i];
                        if (slot == MS - 1) block_store_rows<16>(tile, lane, mh, io.mfs + (block_first * T + (t - (MS - 1))) * D, T * D, nvalid);
                    } else {
                        block_store_rows<D>(tile, lane, mf.v, io.mfs + (block_first * T + t) * D, T * D, nvalid);
                    }
                }
                if (io.Pfs) {
                    double row[D * D];
                    sym_to_row<D>(Pf, row);
                    block_store_rows<D * D>(tile, lane, row, io.Pfs + (block_first * T + t) * D * D, T * D * D, nvalid);
                }
            }
            if (nll) {
                wave_lds_fence();
                _Pragma("unroll 4") for (int k = 0; k < 16; k++) {
                    const int g = k * 64 + lane, tr = g >> 4, e = g & 15;
                    if (tr < nvalid && t0 + e < T) io.nll[(block_first + tr) * T + t0 + e] = ytile[tr * kYPitch + e];
                }
            }
        }
    } else {
        for (int64_t t = 0; t < T; t++) {
            const double y = ys[t];
            Vec<D> mp; Sym<D> Pp;
            double S, innov;
            pred.predict(lane, lds, mf, Pf, mp, Pp);
            Meas::update(mp, Pp, H, Xi, y, mf, Pf, S, innov);
            if (want_nll) {
                cum += nll_increment(S, innov);
                if (nll) nll[t] = cum;
            }
            if constexpr (TILED) {
                if (io.mfs) block_store_rows<D>(tile, lane, mf.v, io.mfs + (block_first * T + t) * D, T * D, nvalid);
                if (io.Pfs) {
                    double row[D * D];
                    sym_to_row<D>(Pf, row);
                    block_store_rows<D * D>(tile, lane, row, io.Pfs + (block_first * T + t) * D * D, T * D * D, nvalid);
                }
            } else {
                if (mfs) store_vec<D>(mfs + t * D, mf);
                if (Pfs) store_sym_full<D>(Pfs + t * D * D, Pf);
            }
        }
    }
    if (writer && io.nll && nll_final) io.nll[trial] = cum;
}

template <class Step>
__global__ void __launch_bounds__(64) smoother_kernel(SmootherIO io, ModelArgs ma) {
    constexpr int D = Step::D;
    constexpr bool WAVE = Step::WAVE;
    constexpr bool TILED = !WAVE && D % 2 == 0;      // one lane per trial: rows go through an LDS transpose
    __shared__ double lds[Step::USES_LDS ? kFanLdsDoubles : 1];
    __shared__ double tile[TILED ? RowTile<D * D>::DOUBLES : 1];
    __shared__ __attribute__((aligned(16))) double rowpark[WAVE ? WaveRow<D>::DOUBLES : 1];
    const int lane = threadIdx.x;
    int64_t trial = WAVE ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 64 + lane;
    const int64_t block_first = (int64_t)blockIdx.x * 64;
    const int nvalid = (io.B - block_first < 64) ? (int)(io.B - block_first) : 64;
    if constexpr (TILED) {
        if (block_first >= io.B) return;
        if (trial >= io.B) trial = io.B - 1;
    } else {
        if (trial >= io.B) return;
    }

    Step step;
    step.setup(ma, trial);
    if constexpr (Step::USES_SIGMA && WAVE) step.sg.stage(dyn_lds(), lane, 64, D);
    const int64_t T = io.T;
    Vec<D> ms, mf;
    Sym<D> Ps, Pf;
    if constexpr (TILED) {
        // filters_smoothers.py:140-142: the last smoothing row is the last filtering row (copied verbatim)
        double row[D * D];
        block_load_rows<D>(tile, lane, io.mfs + (block_first * T + T - 1) * D, T * D, nvalid, ms.v);
        block_store_rows<D>(tile, lane, ms.v, io.mss + (block_first * T + T - 1) * D, T * D, nvalid);
        block_load_rows<D * D>(tile, lane, io.Pfs + (block_first * T + T - 1) * D * D, T * D * D, nvalid, row);
        block_store_rows<D * D>(tile, lane, row, io.Pss + (block_first * T + T - 1) * D * D, T * D * D, nvalid);
        row_to_sym<D>(row, Ps);
        for (int64_t t = T - 2; t >= 0; t--) {
            block_load_rows<D>(tile, lane, io.mfs + (block_first * T + t) * D, T * D, nvalid, mf.v);
            block_load_rows<D * D>(tile, lane, io.Pfs + (block_first * T + t) * D * D, T * D * D, nvalid, row);
            row_to_sym<D>(row, Pf);
            step.step(lane, lds, mf, Pf, ms, Ps);
            block_store_rows<D>(tile, lane, ms.v, io.mss + (block_first * T + t) * D, T * D, nvalid);
            sym_to_row<D>(Ps, row);
            block_store_rows<D * D>(tile, lane, row, io.Pss + (block_first * T + t) * D * D, T * D * D, nvalid);
        }
    } else {
        const double* __restrict__ mfs = io.mfs + trial * T * D;
        const double* __restrict__ Pfs = io.Pfs + trial * T * D * D;
        double* __restrict__ mss = io.mss + trial * T * D;
        double* __restrict__ Pss = io.Pss + trial * T * D * D;
        const bool writer = !WAVE || lane == 0;
        load_vec<D>(mfs + (T - 1) * D, ms);
        load_sym<D>(Pfs + (T - 1) * D * D, Ps);
        if (writer) {   // filters_smoothers.py:140-142: the last smoothing row is the last filtering row (copied verbatim)
            CGP_UNROLL for (int i = 0; i < D; i++) mss[(T - 1) * D + i] = mfs[(T - 1) * D + i];
            CGP_UNROLL for (int i = 0; i < D * D; i++) Pss[(T - 1) * D * D + i] = Pfs[(T - 1) * D * D + i];
        }
        for (int64_t t = T - 2; t >= 0; t--) {
            load_vec<D>(mfs + t * D, mf);
            load_sym<D>(Pfs + t * D * D, Pf);
            step.step(lane, lds, mf, Pf, ms, Ps);
            if constexpr (WAVE && D >= 6) wave_store_step<D>(rowpark, lane, ms, Ps, mss + t * D, Pss + t * D * D);
            else if (writer) {
                store_vec<D>(mss + t * D, ms);
                store_sym_full<D>(Pss + t * D * D, Ps);
            }
        }
    }
}

// Time-parallel discrete smoother (rts / eks / sgp_smoother): one wavefront per trial, K consecutive time steps per
// lane, 64 K steps per tile, tiles walked from the end of the record to its start.  See "TIME-PARALLEL SMOOTHER" in
// cgp_steps.hpp.  Per tile a lane (1) builds the affine maps e_0..e_{K-1} of its steps (the expensive, fully parallel
// part), (2) composes them into one lane aggregate, (3) the 6-round suffix scan runs on the 64 aggregates, (4) the lane
// fetches the composed map of all later lanes, applies it to the carry and then walks its own K steps backwards.
// Cost per 64 steps in wave instructions ~ 734 + 1556 / K (d = 4): K = 4 halves the K = 1 cost.  Loads and stores are
// per-lane runs of K consecutive rows, i.e. one contiguous 64 K * 8 (d + d^2) byte block per tile.
template <class Elem, int K>
__global__ void __launch_bounds__(64) tp_smoother_kernel(SmootherIO io, ModelArgs ma) {
    constexpr int D = Elem::D;
    const int lane = threadIdx.x;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    Elem elem;
    elem.setup(ma, trial);
    if constexpr (Elem::USES_SIGMA) elem.sg.stage(dyn_lds(), lane, 64, D);
    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * D;
    const double* __restrict__ Pfs = io.Pfs + trial * T * D * D;
    double* __restrict__ mss = io.mss + trial * T * D;
    double* __restrict__ Pss = io.Pss + trial * T * D * D;

    Vec<D> ms;
    Sym<D> Ps;
    load_vec<D>(mfs + (T - 1) * D, ms);
    load_sym<D>(Pfs + (T - 1) * D * D, Ps);
    if (lane == 0) {   // filters_smoothers.py:140-142: last smoothing row = last filtering row, copied verbatim
        CGP_UNROLL for (int i = 0; i < D; i++) mss[(T - 1) * D + i] = mfs[(T - 1) * D + i];
        CGP_UNROLL for (int i = 0; i < D * D; i++) Pss[(T - 1) * D * D + i] = Pfs[(T - 1) * D * D + i];
    }
    for (int64_t hi = T - 2; hi >= 0; hi -= 64 * K) {
        const int64_t base = hi - (64 * K - 1) + (int64_t)lane * K;      // first (earliest) step of this lane
        // all K rows are requested before the first element is built (rows before the start of the record are clamped to
        // row 0 and not used): one memory round trip per tile instead of one per element
        Affine<D> e[K];
        Vec<D> mfk[K]; Sym<D> Pfk[K];
        CGP_UNROLL for (int j = 0; j < K; j++) {
            const int64_t t = base + j >= 0 ? base + j : 0;
            load_vec<D>(mfs + t * D, mfk[j]);
            load_sym<D>(Pfs + t * D * D, Pfk[j]);
        }
        CGP_UNROLL for (int j = 0; j < K; j++) {
            affine_identity<D>(e[j]);
            if (base + j >= 0) elem.element(mfk[j], Pfk[j], e[j]);
        }
        // lane aggregate a = e_0 o e_1 o ... o e_{K-1}
        Affine<D> a = e[K - 1];
        CGP_UNROLL for (int j = K - 2; j >= 0; j--) {
            Affine<D> t = e[j];
            affine_compose<D>(t, a);
            a = t;
        }
        // suffix scan over lanes: a_l <- a_l o a_{l+1} o ... o a_63
        CGP_UNROLL for (int delta = 1; delta < 64; delta *= 2) affine_compose_from_lane<D>(a, delta, lane + delta < 64);
        Vec<D> xm = ms; Sym<D> xP = Ps;
        if constexpr (K == 1) {
            // one step per lane: the scanned map already contains the lane's own step
            affine_apply<D>(a, ms, Ps, xm, xP);
            if (base >= 0) {
                store_vec<D>(mss + base * D, xm);
                store_sym_full<D>(Pss + base * D * D, xP);
            }
        } else {
            // state entering this lane's run from later times: (a_{l+1} o ... o a_63)(carry); lane 63 takes the carry itself
            {
                Affine<D> nxt;
                affine_shfl_down<D>(a, 1, nxt);
                if (lane < 63) affine_apply<D>(nxt, ms, Ps, xm, xP);
            }
            CGP_UNROLL for (int j = K - 1; j >= 0; j--) {
                Vec<D> ym; Sym<D> yP;
                affine_apply<D>(e[j], xm, xP, ym, yP);
                xm = ym; xP = yP;
                if (base + j >= 0) {
                    store_vec<D>(mss + (base + j) * D, xm);
                    store_sym_full<D>(Pss + (base + j) * D * D, xP);
                }
            }
        }
        // carry for the next (earlier) tile: the state at the tile's first step, held by lane 0
        CGP_UNROLL for (int i = 0; i < D; i++) ms.v[i] = readlane_f64(xm.v[i], 0);
        CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) Ps.a[i] = readlane_f64(xP.a[i], 0);
    }
}

// Steps per lane of the time-parallel smoother: bounded by the VGPR budget (K affine maps of D^2 + D + D(D+1)/2 doubles).
template <int D> struct TpStepsPerLane { static constexpr int value = D <= 4 ? 4 : 1; };

#ifndef __HIPCC_RTC__          // ---- host side: launchers and the dispatch entry points of the translation units
template <class Elem>
inline hipError_t launch_tp_smoother(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return hipSuccess;
    const size_t dyn = Elem::USES_SIGMA ? sigma_lds_bytes(ma, Elem::D) : 0;
    hipLaunchKernelGGL((tp_smoother_kernel<Elem, TpStepsPerLane<Elem::D>::value>), dim3((unsigned)io.B), dim3(64), dyn, stream, io, ma);
    return hipGetLastError();
}

template <class Pred, class Meas>
inline hipError_t launch_filter(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return hipSuccess;
    const unsigned grid = Pred::WAVE ? (unsigned)io.B : (unsigned)((io.B + 63) / 64);
    const size_t dyn = (Pred::USES_SIGMA && Pred::WAVE) ? sigma_lds_bytes(ma, Pred::D) : 0;
    hipLaunchKernelGGL((filter_kernel<Pred, Meas>), dim3(grid), dim3(64), dyn, stream, io, ma);
    return hipGetLastError();
}
template <class Step>
inline hipError_t launch_smoother(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return hipSuccess;
    const unsigned grid = Step::WAVE ? (unsigned)io.B : (unsigned)((io.B + 63) / 64);
    const size_t dyn = (Step::USES_SIGMA && Step::WAVE) ? sigma_lds_bytes(ma, Step::D) : 0;
    hipLaunchKernelGGL((smoother_kernel<Step>), dim3(grid), dim3(64), dyn, stream, io, ma);
    return hipGetLastError();
}

// Dispatch tables implemented one per translation unit (cgp_inst_*.hip) so that they compile in parallel.
// `key` is d for the linear models and n_harm for the harmonic / KPT ones.  Return CGP_E_UNSUPPORTED if the
// combination is not compiled in, CGP_E_HIP on a launch error.
int dispatch_filter_disc_linear(int method, int key, bool wave, const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_kf4_mfma(const FilterIO&, const ModelArgs&, hipStream_t);      // kf at d = 4 on the matrix cores (cgp_mfma4.hpp)
int dispatch_filter_disc_harm(int method, int key, bool wave, const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_sde_linear(int method, int key, bool wave, const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_sde_harm(int method, int key, bool wave, const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_kpt(int key, bool wave, const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_coop4(const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_mfma4(const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_mfma4_sgp(const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_mfma4_cdsgp(const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_mfma4_cdsgp(const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_split_fixup(const SmootherIO&, double* junction_err, hipStream_t);
int dispatch_filter_mfma4_cdekf(const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_mfma4_cdeks(const SmootherIO&, const ModelArgs&, hipStream_t);
// the matrix-core EKF addresses a trial's outputs through 2 GiB buffer windows (cgp_mfma4.hpp)
inline bool ekf4_mfma_fits(const FilterIO& io) { return io.T * 128 <= 0x7FFFFF00ll; }
int dispatch_filter_coop4_sgp(const FilterIO&, const ModelArgs&, hipStream_t);
// d = 4, one lane per trial, large batches (cgp_lane4.hpp).  The launches it takes: dense or shared records whose rows start on
// 16-byte boundaries (its LDS-DMA moves 16-byte pieces), an even number of steps, output windows of 64 trials within the 2 GiB a
// raw buffer addresses, no time-split segments.
inline bool lane4_filter_fits(const FilterIO& io) {
    return io.segs <= 1 && io.T >= 2 && io.T % 2 == 0 && io.ys_stride % 2 == 0 && ((uintptr_t)io.ys & 15) == 0 && io.T * 128 * 64 <= 0x7FFFFF00ll;
}
int dispatch_filter_lane4(int method, const FilterIO&, const ModelArgs&, hipStream_t);
// ... and its smoothers (eks on the chirp / La Scala LCD models, cd_eks on the chirp SDE): 16-byte aligned inputs (LDS-DMA), output windows
// of 64 trials within 2 GiB
inline bool lane4_smoother_fits(const SmootherIO& io) {
    return io.T >= 2 && io.T * 128 * 64 <= 0x7FFFFF00ll && ((uintptr_t)io.mfs & 15) == 0 && ((uintptr_t)io.Pfs & 15) == 0;
}
int dispatch_smoother_lane4(int method, int model_id, const SmootherIO&, const ModelArgs&, hipStream_t);
// d = 6 / 8 harmonic models in the 8 x 8 tile layout (cgp_coop8.hpp)
bool coop8_filter_sgp_ok(int n_harm, int64_t T, const ModelArgs&);
bool walk4_smoother_fits(int64_t T, const ModelArgs&);
int dispatch_smoother_walk4_linear(int method, const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_walk4_harm(int method, const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_coop8_sgp(int n_harm, const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_coop8_ekf(int n_harm, const FilterIO&, const ModelArgs&, hipStream_t);
bool coop8_smoother_ok(int d, int64_t T, const ModelArgs&);
bool coop8_smoother_harm_ok(int method, const ModelArgs&);
int dispatch_smoother_coop8_linear(int method, int d, const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_coop8_harm(int method, int n_harm, const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_coop4_cdsgp(const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_coop4_cdsgp(const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_filter_coop4_cdekf(const FilterIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_coop4_cdeks(const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_disc_linear(int method, int key, bool wave, const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_disc_harm(int method, int key, bool wave, const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_sde_linear(int method, int key, bool wave, const SmootherIO&, const ModelArgs&, hipStream_t);
int dispatch_smoother_sde_harm(int method, int key, bool wave, const SmootherIO&, const ModelArgs&, hipStream_t);

inline int hip_rc(hipError_t e) { return e == hipSuccess ? CGP_OK : CGP_E_HIP; }
#endif                         // __HIPCC_RTC__

}  // namespace cgp
