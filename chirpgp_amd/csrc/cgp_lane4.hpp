// cgp_lane4.hpp -- d = 4 filters for LARGE batches: one lane per trial, 64 trials per wavefront, built around the memory system.
//
// The reference's only batched use of the filters is its CRLB job (tetralith/jobs/crlb_ekf.py:59-79, crlb_ghf.py:71-75): 10^4 ...
// 10^6 trials x 500 steps.  There every SIMD holds many trials, a trial-step costs ~ 5 vector instructions per lane and the launch
// is bound by its 176 bytes per trial-step -- IF the wavefront never waits for its own stores.  On gfx950 loads,
// stores and LDS-DMA retire through ONE in-order counter (vmcnt): the data of a load is usable only when every OLDER store has
// been acknowledged by memory, i.e. a load whose result is needed soon after the step's output stores costs a full drain of the
// store queue (several microseconds when HBM is saturated).  Round 4's lane kernel (cgp_kernels.hpp: filter_kernel, STAGED) did that
// twice: the measurements of a 16-step block were loaded and consumed at once, and -- capped at 256 registers -- it reloaded 34
// spilled values per step from scratch (profiles/r04_ekf_large_*: 4.0 TB/s with 1.12 x the algorithmic traffic).  This kernel:
//   * measurements: LDS-DMA (global_load_lds_dwordx4: no registers) of the NEXT block's 64 x 16 doubles while the current block
//     runs; the wait at the block boundary is a counted vmcnt(N), N = the stores issued since, so no store is waited for;
//   * no spill: the means of four steps (one 128-byte line) wait in registers with compile-time slots, the covariance leaves
//     packed (10 doubles per lane) through a 6 KB LDS transpose, and nothing else is held across steps;
//   * every global store instruction writes whole 128-byte lines (8 trials x 128 B): covariance rows per step, means per four
//     steps, cumulative NLL per 16 steps (written into the consumed measurements' LDS slots);
//   * outputs are addressed as raw-buffer windows over the wavefront's 64 trials: lanes past the batch are dropped by the range
//     check, so the step has no exec-masked store.
// tools/ubench/store_pattern.hip issues exactly this access pattern without arithmetic: 5.0 - 5.6 TB/s on MI355X, the roof here.
#pragma once
#include "cgp_kernels.hpp"
#include "cgp_coop4.hpp"

namespace cgp {

using LdsDouble2Ptr = __attribute__((address_space(3))) double2_t*;
using LdsDoublePtr = __attribute__((address_space(3))) double*;

struct Lane4 {
    static constexpr int KB = 16;                  // steps per block: 16 x 8 B = one line of measurements / NLL per trial
    static constexpr int YDOUBLES = 64 * KB;       // one block, laid out [piece 0..7][trial 0..63][2] (what the LDS-DMA writes)
    static constexpr int PITCH_P = 12;             // packed covariance row (10 doubles) + 2: conflict-free 16-byte row writes
    static constexpr int PITCH_M = 18;             // a line of means (4 steps x 4) + 2
    // The LDS transposes run in passes so that a wavefront needs 19 KB in all and EIGHT fit a CU: covariance rows 32 lanes at a
    // time, lines of means 16 lanes at a time (a pass = masked row writes, then whole-wave reads + stores; DS instructions of one
    // wavefront execute in order, so passes only need the compiler kept from reordering them).
    static constexpr int TILE = 32 * PITCH_P;      // doubles (= 16 * PITCH_M + 96 = 64 rows of 4 + 2 for the head / tail rows)
};

// 64 trials x 16 steps of measurements into LDS: instruction pc moves, for every lane's trial, the 16 bytes of steps t0 + 2 pc,
// t0 + 2 pc + 1 to lds_base + 1024 pc + 16 lane.  Pieces past the end of the record re-read its last piece (never consumed).
// M0 is the compiler's: saved, set and restored inside the statement (cdna_hip_programming.md 5.7).
CGP_DEV void lane4_dma_y(unsigned lds_base, const double* __restrict__ rec, int64_t t0, int64_t T) {
    CGP_UNROLL for (int pc = 0; pc < 8; pc++) {
        int64_t e = t0 + 2 * pc;
        if (e + 2 > T) e = T - 2;
        const double* src = rec + e;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + pc * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
}
// Wait until all but the N youngest vector-memory operations of this wavefront are done (N a compile-time constant).
template <int N> CGP_DEV void lane4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }
// ... N a wave-uniform run-time multiple of four up to 40 (anything else: wait for everything)
CGP_DEV void lane4_wait_vm_n(int n) {
    switch (n) {
    case 4: lane4_wait_vm<4>(); break;    case 8: lane4_wait_vm<8>(); break;    case 12: lane4_wait_vm<12>(); break;
    case 16: lane4_wait_vm<16>(); break;  case 20: lane4_wait_vm<20>(); break;  case 24: lane4_wait_vm<24>(); break;
    case 28: lane4_wait_vm<28>(); break;  case 32: lane4_wait_vm<32>(); break;  case 36: lane4_wait_vm<36>(); break;
    case 40: lane4_wait_vm<40>(); break;
    default: lane4_wait_vm<0>();
    }
}

// WHOLE LINES ONLY.  tools/ubench/store_pattern.hip, same bytes: 4.7 ms when every store instruction fills 128-byte lines, 6.1 - 6.3 ms
// when the 8-byte NLL stream alone leaves in aligned 64-byte halves or in 128-byte chunks that straddle two lines (a record of T = 500
// steps starts 32, 64 or 96 bytes into a line for three trials out of four), 7.7 ms with the means in 32-byte pieces.  So the blocks of
// a trial are cut where ITS rows cross line boundaries: trial b's rows start (b T mod 16) steps into a line, and a wavefront takes 64
// trials of the SAME phase -- every `period`-th trial, period = 16 / gcd(T, 16) -- runs (16 - phase) mod 16 head steps row by row and
// from there on blocks of 16 steps whose measurement, NLL and mean chunks are whole lines (base pointers on line boundaries assumed:
// torch allocations are; any other base only costs the partial lines back).  Workgroup (g, r) = (blockIdx / period, blockIdx mod
// period) takes the trials 64 period g + r + period l, lane l.
// WANT_P: the covariances are an output (compile time: the launch that keeps the means alone -- the CRLB job -- carries none of the
// covariance path's registers, addresses and branches).
template <class Pred, class Meas, bool WANT_P>
__global__ void __launch_bounds__(64) lane4_filter_kernel(FilterIO io, ModelArgs ma, int period) {
    static_assert(Pred::D == 4 && !Pred::WAVE, "one lane per trial, d = 4");
    constexpr int D = 4;
    __shared__ __attribute__((aligned(16))) double ybuf[2 * Lane4::YDOUBLES];
    __shared__ __attribute__((aligned(16))) double tile[Lane4::TILE];
    // LIGHT steps (the EKF: ~ 260 vector instructions) are unrolled four times and keep a line of means in registers with compile-time
    // slots; HEAVY steps (a sigma-point fan per step: thousands) are not -- four copies of the fan cost the kernel its registers (256 +
    // spills, measured) -- and park each step's mean in a second LDS tile instead (such a kernel is bound by its arithmetic, not by LDS
    // capacity)
    constexpr bool LIGHT = Pred::LANE_TWO_WAVES;
    __shared__ __attribute__((aligned(16))) double mtile[LIGHT ? 2 : 64 * Lane4::PITCH_M];
    __shared__ double lds[Pred::USES_LDS ? kFanLdsDoubles : 1];
    const int lane = threadIdx.x;
    const int64_t group = (int64_t)(blockIdx.x / (unsigned)period);
    const int64_t block_first = group * 64 * period + (int64_t)(blockIdx.x % (unsigned)period);
    if (block_first >= io.B) return;
    const int64_t nv = (io.B - block_first + period - 1) / period;
    const int nvalid = nv < 64 ? (int)nv : 64;
    int64_t trial = block_first + (int64_t)period * (lane < nvalid ? lane : nvalid - 1);      // lanes past the batch redo the last trial; the windows drop their stores

    Pred pred;
    pred.setup(ma, trial);
    if constexpr (Pred::USES_SIGMA) pred.sg.stage(dyn_lds(), lane, 64, D);       // (SgpPredictLane: the fan reads the set from LDS)
    pred.large_batch();                                  // small-angle sin / cos where every lane's angle allows (cgp_fastmath.hpp)
    // (the per-lane softplus stays the naive form here: the wide common-regime form -- cgp_models.hpp: softplus_pair_wide -- pays only
    // while every lane's frequency state is >= 1.5, and the CRLB job's is a zero-mean GP: its fallback ran in every wavefront, +23 %
    // vector instructions, measured)
    Vec<D> H, mf;
    Sym<D> Pf;
    if (io.H) load_vec<D>(io.H + trial * io.H_stride, H);
    else { CGP_UNROLL for (int i = 0; i < D; i++) H.v[i] = 0.0; }
    // wave-uniform: every lane's measurement vector is e_1 and the measurement is linear in the state (not ekf_for_kpt)
    const bool he1 = Meas::LINEAR && __builtin_amdgcn_ballot_w64(!(H.v[0] == 1.0 && H.v[1] == 0.0 && H.v[2] == 0.0 && H.v[3] == 0.0)) == 0;
    const double Xi = io.Xi[trial * io.Xi_stride];
    load_vec<D>(io.m0 + trial * io.m0_stride, mf);
    load_sym<D>(io.P0 + trial * io.P0_stride, Pf);

    const int64_t T = io.T;
    const double* __restrict__ rec = io.record(trial);
    const bool nll_final = (io.flags & CGP_NLL_FINAL_ONLY) != 0;
    const bool want_nll = io.nll != nullptr, nll_rows = want_nll && !nll_final;
    const bool want_m = io.mfs != nullptr;
    constexpr bool want_P = WANT_P;

    // output windows over this wavefront's trials; per-lane byte offsets of (trial 8 i + sub, 16-byte piece pc) in them
    OobWindow wP, wM, wN;
    const int64_t span = (int64_t)(nvalid - 1) * period + 1;         // trials from this wavefront's first to its last valid one
    wP.init(want_P ? io.Pfs + block_first * T * 16 : nullptr, span * T * 128);
    wM.init(want_m ? io.mfs + block_first * T * 4 : nullptr, span * T * 32);
    wN.init(nll_rows ? io.nll + block_first * T : nullptr, span * T * 8);
    // (eight CONSECUTIVE lanes cover one trial's 128-byte line: the memory pipeline merges neighbouring lanes' pieces into line
    // requests -- with the trial in the low lane bits instead, every lane's 16 bytes travel alone: 6.4 against ... ms, measured)
    const int sub = lane >> 3, pc = lane & 7;
    const unsigned rowP = (unsigned)(T * period) * 128u, rowM = (unsigned)(T * period) * 32u, rowN = (unsigned)(T * period) * 8u;      // lane to lane
    unsigned voffP = (unsigned)sub * rowP + (unsigned)pc * 16u;      // + 128 per step
    unsigned voffM = (unsigned)sub * rowM + (unsigned)pc * 16u;      // + 128 per four steps
    unsigned voffN = (unsigned)sub * rowN + (unsigned)pc * 16u;      // + 128 per block
    // the two packed entries of piece pc of a full 4 x 4 row-major covariance row: (i, j), (i, j + 1) with i = pc / 2, j = 2 (pc % 2)
    const int ia = (pc == 0) ? 0 : (pc == 1) ? 3 : (pc == 2) ? 1 : (pc == 3) ? 4 : (pc == 4) ? 3 : (pc == 5) ? 5 : (pc == 6) ? 6 : 8;
    const int ib = (pc == 0) ? 1 : (pc == 1) ? 6 : (pc == 2) ? 2 : (pc == 3) ? 7 : (pc == 4) ? 4 : (pc == 5) ? 8 : (pc == 6) ? 7 : 9;
    const double* tPa = tile + sub * Lane4::PITCH_P + ia;
    const double* tPb = tile + sub * Lane4::PITCH_P + ib;
    const double* tM = tile + sub * Lane4::PITCH_M + 2 * pc;

    double cum = 0.0;
    // one filtering step of this lane's trial; leaves its cumulative NLL in the consumed measurement's slot and its covariance row in HBM
    auto step = [&](double y, double* nll_slot) __attribute__((always_inline)) {      // (not inlined, its captures live in scratch)
        Vec<D> mp; Sym<D> Pp;
        double S, innov;
        pred.predict(lane, lds, mf, Pf, mp, Pp);
        if (he1) {
            // H = e_1 (every chirp model of the reference: models.py:118): H Pp is column 0 as it stands, S = Pp_00 + Xi
            S = Pp(0, 0) + Xi;
            innov = y - mp.v[0];
            const double rS = rcp_nr(S);
            Vec<D> K;
            CGP_UNROLL for (int i = 0; i < D; i++) K.v[i] = Pp(i, 0) * rS;
            CGP_UNROLL for (int i = 0; i < D; i++) mf.v[i] = fma(K.v[i], innov, mp.v[i]);
            CGP_UNROLL for (int i = 0; i < D; i++)
                CGP_UNROLL for (int j = 0; j <= i; j++) Pf(i, j) = fma(-K.v[i], Pp(j, 0), Pp(i, j));
        } else {
            Meas::update(mp, Pp, H, Xi, y, mf, Pf, S, innov);
        }
        if (want_nll) {
            cum += nll_increment(S, innov);
            if (nll_rows) *nll_slot = cum;
        }
        if constexpr (want_P) {
            // two passes of 32 trials through the tile; the second is written right behind the reads of the first (DS instructions of a
            // wavefront execute in order), so the reads' latency is paid once a step
            double* row = tile + (lane & 31) * Lane4::PITCH_P;
            double pa[2][4], pb[2][4];
            CGP_UNROLL for (int h = 0; h < 2; h++) {
                if ((lane >> 5) == h) {
                    CGP_UNROLL for (int c = 0; c < 5; c++) *reinterpret_cast<double2*>(row + 2 * c) = make_double2(Pf.a[2 * c], Pf.a[2 * c + 1]);
                }
                wave_lds_fence();
                CGP_UNROLL for (int i = 0; i < 4; i++) { pa[h][i] = tPa[i * 8 * Lane4::PITCH_P]; pb[h][i] = tPb[i * 8 * Lane4::PITCH_P]; }
                wave_lds_fence();
            }
            CGP_UNROLL for (int h = 0; h < 2; h++)
                CGP_UNROLL for (int i = 0; i < 4; i++) wP.store2(pa[h][i], pb[h][i], voffP + (unsigned)(4 * h + i) * 8u * rowP);
        }
        if constexpr (want_P) voffP += 128u;
    };

    // up to 16 steps [ta, ta + n) with a run-time count -- the head in front of the first line boundary and the tail behind the last:
    // same step, rows of means and NLL values one by one (partial lines: a record's first and last line are shared with its neighbours)
    auto partial = [&](double* yb, int64_t ta, int n) __attribute__((always_inline)) {
        for (int k = 0; k < n; k++) {
            double* slot = yb + (k >> 1) * 128 + lane * 2 + (k & 1);
            step(*slot, slot);
            if (want_m) block_store_rows<D>(tile, lane, mf.v, io.mfs + (block_first * T + ta + k) * D, T * period * D, nvalid);
        }
        if (nll_rows) {
            wave_lds_fence();
            for (int i = 0; i < 16; i++) {
                const int g = i * 64 + lane, tr = g >> 4, e = g & 15;
                if (tr < nvalid && e < n) io.nll[(block_first + (int64_t)tr * period) * T + ta + e] = yb[(e >> 1) * 128 + tr * 2 + (e & 1)];
            }
            wave_lds_fence();
        }
        voffM += 32u * (unsigned)n;
        voffN += 8u * (unsigned)n;
    };

    const unsigned ybase = (unsigned)(uintptr_t)(LdsDoublePtr)ybuf;
    // vector-memory operations a full block issues after the DMA of the block that follows it (capped at the counter's 63)
    const int block_stores = (want_P ? 128 : 0) + (want_m ? 32 : 0) + (nll_rows ? 8 : 0);
    int64_t t = 0;
    int cur = 0;
    bool drained = true;                                             // the next wait is a full one (no count of the stores since the DMA)
    lane4_dma_y(ybase, rec, 0, T);
    {
        const int phase = (int)((block_first * T) & 15);             // wave-uniform: period * T is a multiple of 16
        const int64_t head = (16 - phase) & 15;
        if (head > 0) {
            const int n = head < T ? (int)head : (int)T;
            lane4_wait_vm<0>();
            if (n < T) lane4_dma_y(ybase + Lane4::YDOUBLES * 8u, rec, n, T);
            partial(ybuf, 0, n);
            t = n; cur = 1;
        }
    }
    for (; t + Lane4::KB <= T; t += Lane4::KB) {
        double* yb = ybuf + cur * Lane4::YDOUBLES;
        // this block's measurements were requested a whole block ago: wait for them, not for the stores issued since
        if (drained || block_stores < 8) lane4_wait_vm<0>();
        else if (block_stores >= 63) lane4_wait_vm<63>();
        else if (block_stores == 40) lane4_wait_vm<40>();
        else if (block_stores == 32) lane4_wait_vm<32>();
        else lane4_wait_vm<8>();
        drained = false;
        if (t + Lane4::KB < T) lane4_dma_y(ybase + (unsigned)(cur ^ 1) * (Lane4::YDOUBLES * 8u), rec, t + Lane4::KB, T);
        if constexpr (!LIGHT) {
            _Pragma("unroll 1") for (int q = 0; q < 4; q++) {
                _Pragma("unroll 1") for (int sidx = 0; sidx < 4; sidx++) {
                    const int k = 4 * q + sidx;
                    double* slot = yb + (k >> 1) * 128 + lane * 2 + (k & 1);
                    step(*slot, slot);
                    if (want_m) {
                        double* row = mtile + lane * Lane4::PITCH_M + 4 * sidx;
                        *reinterpret_cast<double2*>(row) = make_double2(mf.v[0], mf.v[1]);
                        *reinterpret_cast<double2*>(row + 2) = make_double2(mf.v[2], mf.v[3]);
                    }
                }
                if (want_m) {
                    wave_lds_fence();
                    CGP_UNROLL for (int i = 0; i < 8; i++) {
                        const double2 v = *reinterpret_cast<const double2*>(mtile + (8 * i + sub) * Lane4::PITCH_M + 2 * pc);
                        wM.store2(v.x, v.y, voffM + (unsigned)i * 8u * rowM);
                    }
                    wave_lds_fence();
                }
                voffM += 128u;
            }
        } else
        _Pragma("unroll 1") for (int q = 0; q < 4; q++) {
            double* yq = yb + q * 256 + lane * 2;                    // steps 4 q, 4 q + 1 | + 128: steps 4 q + 2, 4 q + 3
            const double2 y01 = *reinterpret_cast<const double2*>(yq);
            const double2 y23 = *reinterpret_cast<const double2*>(yq + 128);
            double mh[16];
            step(y01.x, yq);
            CGP_UNROLL for (int i = 0; i < D; i++) mh[i] = mf.v[i];
            step(y01.y, yq + 1);
            CGP_UNROLL for (int i = 0; i < D; i++) mh[4 + i] = mf.v[i];
            step(y23.x, yq + 128);
            CGP_UNROLL for (int i = 0; i < D; i++) mh[8 + i] = mf.v[i];
            step(y23.y, yq + 129);
            CGP_UNROLL for (int i = 0; i < D; i++) mh[12 + i] = mf.v[i];
            if (want_m) {
                // four passes of 16 trials through the tile.  A wavefront's DS instructions execute in order, so pass h + 1 may be WRITTEN
                // right behind the reads of pass h, before their data is back: the reads' latency is paid once a quad, not four times
                // (the stores of pass h follow the reads of pass h + 1; the compiler counts lgkmcnt for them)
                double* row = tile + (lane & 15) * Lane4::PITCH_M;
                double2 va[4], vb[4];
                CGP_UNROLL for (int h = 0; h < 4; h++) {
                    if ((lane >> 4) == h) {
                        CGP_UNROLL for (int c = 0; c < 8; c++) *reinterpret_cast<double2*>(row + 2 * c) = make_double2(mh[2 * c], mh[2 * c + 1]);
                    }
                    wave_lds_fence();
                    va[h] = *reinterpret_cast<const double2*>(tM);
                    vb[h] = *reinterpret_cast<const double2*>(tM + 8 * Lane4::PITCH_M);
                    wave_lds_fence();
                    if (h > 0) {
                        wM.store2(va[h - 1].x, va[h - 1].y, voffM + (unsigned)(2 * h - 2) * 8u * rowM);
                        wM.store2(vb[h - 1].x, vb[h - 1].y, voffM + (unsigned)(2 * h - 1) * 8u * rowM);
                    }
                }
                wM.store2(va[3].x, va[3].y, voffM + 6u * 8u * rowM);
                wM.store2(vb[3].x, vb[3].y, voffM + 7u * 8u * rowM);
            }
            voffM += 128u;
        }
        if (nll_rows) {
            wave_lds_fence();
            CGP_UNROLL for (int i = 0; i < 8; i++) {
                const double2 v = *reinterpret_cast<const double2*>(yb + pc * 128 + (8 * i + sub) * 2);
                wN.store2(v.x, v.y, voffN + (unsigned)i * 8u * rowN);
            }
            wave_lds_fence();
        }
        voffN += 128u;
        cur ^= 1;
    }
    if (t < T) {
        lane4_wait_vm<0>();
        partial(ybuf + cur * Lane4::YDOUBLES, t, (int)(T - t));
    }
    if (want_nll && nll_final && lane < nvalid) io.nll[trial] = cum;
}

// Trials of equal line phase are `period` apart: the smallest power of two with period * T a multiple of 16 (T is even: <= 8), or 1
// (no alignment, partial lines) where 64 such trials would not fit the 2 GiB a raw-buffer window addresses.
inline int lane4_period(int64_t T) {
    int p = 1;
    while ((p * T) % 16 != 0) p *= 2;
    return (T * p * 128 * 64 <= kOobMaxBytes) ? p : 1;
}
template <class Pred, class Meas>
inline hipError_t launch_lane4_filter(const FilterIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return hipSuccess;
    const int period = lane4_period(io.T);
    const int64_t groups = (io.B + 64 * period - 1) / (64 * period);
    const size_t dyn = Pred::USES_SIGMA ? sigma_lds_bytes(ma, Pred::D) : 0;
    if (io.Pfs) hipLaunchKernelGGL((lane4_filter_kernel<Pred, Meas, true>), dim3((unsigned)(groups * period)), dim3(64), dyn, stream, io, ma, period);
    else hipLaunchKernelGGL((lane4_filter_kernel<Pred, Meas, false>), dim3((unsigned)(groups * period)), dim3(64), dyn, stream, io, ma, period);
    return hipGetLastError();
}

// ====================================================================================================================== smoothers
// The same rules for the backward pass (eks, cd_eks at d = 4; per trial and step 160 bytes in -- mf, Pf -- and 160 out -- ms, Ps): the
// generic lane kernel (cgp_kernels.hpp: smoother_kernel) loads a step's filtering rows right behind the previous step's stores (a full
// drain of the store queue per step) and moves the means in 32-byte pieces -- 3.7 TB/s at 65 536 x 500, with or without the RK4 of cd_eks
// in between (profiles/r05_lane_smoothers.txt).  Here:
//   * the covariance rows of the NEXT step (128 B per trial) and the mean lines of the NEXT four steps (one 128-byte line per trial) come
//     in by LDS-DMA while the current step computes; the waits are counted (vmcnt(8) / (16): the stores issued since stay in flight);
//   * a DMA instruction lays its 64 lanes' 16-byte pieces down contiguously, so it fetches 8 trials x 128 B (whole lines) with the pieces
//     of trial s rotated by s slots (and the instructions' 1 KB pieces 1040 bytes apart, the odd ones shifted by one slot more): a lane then reads ITS trial's row back without bank
//     conflicts -- what a pitch of 18 doubles does for the register-staged tiles;
//   * smoothed means leave as whole lines per four steps, covariances packed through the 3 KB tile, as in the filter; trials are grouped
//     by the phase of their rows against the lines (period = 4 / gcd(T, 4)), the few rows outside whole quads go row by row.
struct Lane4S { static constexpr int PITCH = 1040, ROWS = (8 * 1040 + 16) / 8 + 2; };      // bytes between the instructions' 1 KB pieces (the odd ones shifted by 16 more); doubles of a set

// 64 rows of 128 bytes, one per trial of the wavefront, into LDS.  g0: this lane's source for instruction 0 -- row of trial (lane >> 3),
// piece ((lane & 7) + (lane >> 3)) & 7; instruction i serves the trials 8 i .. 8 i + 7: `step` doubles further (trials past the batch
// clamped by the caller through `last`: the row of the last valid trial).
CGP_DEV void lane4_dma_rows(unsigned lds_base, const double* __restrict__ g0, int64_t step, int sub, int nvalid, const double* __restrict__ last) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (LDS reads of the buffer being refilled have returned)
    CGP_UNROLL for (int i = 0; i < 8; i++) {
        const double* src = (8 * i + sub < nvalid) ? g0 + i * step : last;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + i * Lane4S::PITCH + (i & 1) * 16);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
}
// byte offset (from the set's base) of piece p of the row of trial `lane`
CGP_DEV unsigned lane4_row_piece(int lane, int p) {
    const int i = lane >> 3, s = lane & 7;
    return (unsigned)(i * Lane4S::PITCH + (i & 1) * 16 + s * 128 + ((p - s) & 7) * 16);
}

// SEL (cgp_smoother_select): per step every lane also parks the selected component's smoothed mean / variance / E[f(V)] -- whichever
// are wanted, `nsel` of them -- in LDS, [16 steps][64 trials + 1]; every 16 steps (one 128-byte line of a [B][T] array per trial: trials
// of equal line phase share a wavefront, period = 16 / gcd(T, 16)) the wavefront writes them as whole lines, eight 16-byte-per-lane
// store instructions per output.  The full outputs are optional then (mss / Pss NULL: 8 - 24 bytes a step leave instead of 160).
// SELN = steps per flush: 16 (whole lines) for ONE selected output; 8 (aligned half lines, written 8 steps apart by the same wavefront) for
// two or three -- their 16-step buffers would take the workgroup past 40 KB of LDS, three workgroups a CU instead of four.
template <int SELN> struct Lane4Sel { static constexpr int PITCH = 65, DOUBLES = SELN * 65, LPT = SELN / 2, TPI = 64 / LPT; };
// NBUF (round 6): covariance-row buffers -- rows are requested NBUF - 1 steps ahead.  Two (one step ahead) is what launches use; three
// (where the LDS allows a third buffer beside four workgroups a CU) is kept behind CGP_DBG_LANE_BUFFERS as a measured negative result.
// FULL: the kernel may write full rows (the 3 KB transpose tile exists); a launch with the selected outputs alone runs the variant without.
template <class Step, int SELN = 0, int NBUF = 2, bool FULL = true>
__global__ void __launch_bounds__(64) lane4_smoother_kernel(SmootherIO io, ModelArgs ma, int period) {
    constexpr bool SEL = SELN > 0;
    static_assert(FULL || SEL, "a launch without full rows selects something");
    using LS = Lane4Sel<SEL ? SELN : 16>;
    static_assert(Step::D == 4 && !Step::WAVE && !Step::USES_SIGMA, "one lane per trial, d = 4, no sigma-point set");
    constexpr int D = 4;
    __shared__ __attribute__((aligned(16))) double pin[NBUF * Lane4S::ROWS];
    __shared__ __attribute__((aligned(16))) double min_[Lane4S::ROWS];
    __shared__ __attribute__((aligned(16))) double tile[FULL ? Lane4::TILE : 2];
    __shared__ double lds[Step::USES_LDS ? kFanLdsDoubles : 1];
    __shared__ double ghrule[SEL ? 2 * kGhMaxOrder : 1];
    const int lane = threadIdx.x;
    const int64_t group = (int64_t)(blockIdx.x / (unsigned)period);
    const int64_t block_first = group * 64 * period + (int64_t)(blockIdx.x % (unsigned)period);
    if (block_first >= io.B) return;
    const int64_t nv = (io.B - block_first + period - 1) / period;
    const int nvalid = nv < 64 ? (int)nv : 64;
    const bool valid = lane < nvalid;
    const int64_t trial = block_first + (int64_t)period * (valid ? lane : nvalid - 1);
    if constexpr (SEL) { sel_stage_rule(io.sel, ghrule, lane); wave_lds_fence(); }

    Step step;
    step.setup(ma, trial);
    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * D;
    const double* __restrict__ Pfs = io.Pfs + trial * T * D * D;
    const bool want_m = FULL && (!SEL || io.mss != nullptr), want_P = FULL && (!SEL || io.Pss != nullptr);      // (wave-uniform)
    double* __restrict__ mss = want_m ? io.mss + trial * T * D : nullptr;
    double* __restrict__ Pss = want_P ? io.Pss + trial * T * D * D : nullptr;
    // the selected outputs: `nsel` arrays parked in the dynamic LDS in the order mean, variance, expectation
    const int comp = SEL ? io.sel.comp : 0;
    const bool sel_m = SEL && io.sel.mean, sel_v = SEL && io.sel.var, sel_e = SEL && io.sel.expect;
    double* const selM = SEL ? dyn_lds() : nullptr;
    double* const selV = selM + (sel_m ? LS::DOUBLES : 0);
    double* const selE = selV + (sel_v ? LS::DOUBLES : 0);
    auto pick_mean = [&](const Vec<D>& m) { return comp == 0 ? m.v[0] : comp == 1 ? m.v[1] : comp == 2 ? m.v[2] : m.v[3]; };
    auto pick_var = [&](const Sym<D>& P) { return comp == 0 ? P.a[0] : comp == 1 ? P.a[2] : comp == 2 ? P.a[5] : P.a[9]; };      // the diagonal of the packed lower triangle

    // filters_smoothers.py:140-142: the last smoothing row is the last filtering row (copied verbatim, all 16 covariance entries)
    Vec<D> ms; Sym<D> Ps;
    load_vec<D>(mfs + (T - 1) * D, ms);
    load_sym<D>(Pfs + (T - 1) * D * D, Ps);
    if (valid) {
        if (want_m) { CGP_UNROLL for (int i = 0; i < D; i++) mss[(T - 1) * D + i] = mfs[(T - 1) * D + i]; }
        if (want_P) { CGP_UNROLL for (int i = 0; i < D * D; i++) Pss[(T - 1) * D * D + i] = Pfs[(T - 1) * D * D + i]; }
        if constexpr (SEL) sel_write(io.sel, ghrule, trial * T + T - 1, pick_mean(ms), pick_var(Ps));
    }
    // one row outside the whole quads: inputs and outputs straight from / to this lane's rows (a handful per record)
    auto edge = [&](int64_t t) __attribute__((always_inline)) {
        Vec<D> mf; Sym<D> Pf;
        load_vec<D>(mfs + t * D, mf);
        load_sym<D>(Pfs + t * D * D, Pf);
        step.step(lane, lds, mf, Pf, ms, Ps);
        if (valid) {
            if (want_m) store_vec<D>(mss + t * D, ms);
            if (want_P) store_sym_full<D>(Pss + t * D * D, Ps);
            if constexpr (SEL) sel_write(io.sel, ghrule, trial * T + t, pick_mean(ms), pick_var(Ps));
        }
    };
    // rows [a_lo, a_lo + 4 nq) are whole lines of four means for every trial of this wavefront (wave-uniform: period * T is a multiple of 4);
    // SEL: whole lines of SIXTEEN doubles of a [B][T] array (period * T a multiple of 16), and a whole number of them
    constexpr int kLine = SEL ? SELN : 4;                                // steps per aligned block: a line of means, or SELN selected values
    const int a_lo = (int)((kLine - ((block_first * T) & (kLine - 1))) & (kLine - 1));
    const int64_t nq = (T - 1 > a_lo) ? ((T - 1 - a_lo) / 4) & ~(int64_t)(kLine / 4 - 1) : 0;
    for (int64_t t = T - 2; t >= (nq > 0 ? a_lo + 4 * nq : 0); t--) edge(t);     // (no whole quad: every row goes this way)
    if (nq > 0) {
        // output windows over this wavefront's trials, per-lane offsets of (trial 8 i + sub, piece pc), as in the filter
        OobWindow wP, wM;
        const int64_t span = (int64_t)(nvalid - 1) * period + 1;
        wP.init(want_P ? io.Pss + block_first * T * 16 : nullptr, span * T * 128);
        wM.init(want_m ? io.mss + block_first * T * 4 : nullptr, span * T * 32);
        OobWindow wSm, wSv, wSe;                                         // the selected outputs: [B][T] doubles
        if constexpr (SEL) {
            wSm.init(sel_m ? io.sel.mean + block_first * T : nullptr, span * T * 8);
            wSv.init(sel_v ? io.sel.var + block_first * T : nullptr, span * T * 8);
            wSe.init(sel_e ? io.sel.expect + block_first * T : nullptr, span * T * 8);
        }
        const unsigned rowS = (unsigned)(T * period) * 8u;
        const int nflush = SEL ? LS::LPT * ((sel_m ? 1 : 0) + (sel_v ? 1 : 0) + (sel_e ? 1 : 0)) : 0;      // store instructions of one flush
        const int sub = lane >> 3, pc = lane & 7;
        const unsigned rowP = (unsigned)(T * period) * 128u, rowM = (unsigned)(T * period) * 32u;
        const int64_t t_hi = a_lo + 4 * nq - 1;                          // the first row of the quads (processed downwards)
        unsigned voffP = (unsigned)sub * rowP + (unsigned)pc * 16u + (unsigned)t_hi * 128u;
        unsigned voffM = (unsigned)sub * rowM + (unsigned)pc * 16u + (unsigned)(t_hi - 3) * 32u;
        const int ia = (pc == 0) ? 0 : (pc == 1) ? 3 : (pc == 2) ? 1 : (pc == 3) ? 4 : (pc == 4) ? 3 : (pc == 5) ? 5 : (pc == 6) ? 6 : 8;
        const int ib = (pc == 0) ? 1 : (pc == 1) ? 6 : (pc == 2) ? 2 : (pc == 3) ? 7 : (pc == 4) ? 4 : (pc == 5) ? 8 : (pc == 6) ? 7 : 9;
        const double* tPa = tile + sub * Lane4::PITCH_P + ia;
        const double* tPb = tile + sub * Lane4::PITCH_P + ib;
        const double* tM = tile + sub * Lane4::PITCH_M + 2 * pc;
        // DMA sources of this lane: trial (block_first + period * sub) for instruction 0, piece rotated by the trial's slot
        const int64_t tr_step = 8 * period * T;                          // trials from one instruction to the next, in rows
        const int rot = ((lane & 7) + sub) & 7;
        const double* gP = io.Pfs + (block_first + (int64_t)period * sub) * T * 16 + 2 * rot;
        const double* gM = io.mfs + (block_first + (int64_t)period * sub) * T * 4 + 2 * rot;
        const double* lastP = io.Pfs + (block_first + (int64_t)period * (nvalid - 1)) * T * 16 + 2 * rot;
        const double* lastM = io.mfs + (block_first + (int64_t)period * (nvalid - 1)) * T * 4 + 2 * rot;
        const unsigned pbase = (unsigned)(uintptr_t)(LdsDoublePtr)pin, mbase = (unsigned)(uintptr_t)(LdsDoublePtr)min_;
        const char* pin_b = reinterpret_cast<const char*>(pin);
        const char* min_b = reinterpret_cast<const char*>(min_);
        unsigned rd[8];                                                  // this lane's row pieces in a DMA-ed set
        CGP_UNROLL for (int p = 0; p < 8; p++) rd[p] = lane4_row_piece(lane, p);

        constexpr int AHEAD = NBUF - 1;                                  // steps between a row's request and its use
        lane4_dma_rows(mbase, gM + (t_hi - 3) * 4, tr_step * 4, sub, nvalid, lastM + (t_hi - 3) * 4);
        CGP_UNROLL for (int j = 0; j < AHEAD; j++)
            if (t_hi - j >= a_lo) lane4_dma_rows(pbase + (unsigned)j * (Lane4S::ROWS * 8u), gP + (t_hi - j) * 16, tr_step * 16, sub, nvalid, lastP + (t_hi - j) * 16);
        int cur = 0;
        // What was issued behind the request of the row about to be used, and may stay in flight across the wait (in-order completion):
        // the requests of the AHEAD - 1 rows after it, and the previous step's line of means and stores.
        int younger = 0;                                                 // ... of the previous step: its DMA of means + its stores
        for (int64_t k = nq - 1; k >= 0; k--) {
            const int64_t a = a_lo + 4 * k;
            double mq[16], mh[16];
            CGP_UNROLL for (int s4 = 3; s4 >= 0; s4--) {
                const int64_t t = a + s4;
                {
                    int rows_behind = 0;                                 // row requests younger than this row's: rows t - 1 .. t - AHEAD + 1
                    CGP_UNROLL for (int j = 1; j < AHEAD; j++) rows_behind += (t - j >= a_lo) ? 8 : 0;
                    lane4_wait_vm_n(rows_behind + younger);
                    younger = 0;
                }
                if (t - AHEAD >= a_lo)
                    lane4_dma_rows(pbase + (unsigned)((cur + AHEAD) % NBUF) * (Lane4S::ROWS * 8u), gP + (t - AHEAD) * 16, tr_step * 16, sub, nvalid, lastP + (t - AHEAD) * 16);
                if (s4 == 3) {
                    CGP_UNROLL for (int p = 0; p < 8; p++) {
                        const double2 v = *reinterpret_cast<const double2*>(min_b + rd[p]);
                        mq[2 * p] = v.x; mq[2 * p + 1] = v.y;
                    }
                    if (k > 0) { lane4_dma_rows(mbase, gM + (a - 4) * 4, tr_step * 4, sub, nvalid, lastM + (a - 4) * 4); younger += 8; }
                }
                Vec<D> mf; Sym<D> Pf;
                CGP_UNROLL for (int i = 0; i < D; i++) mf.v[i] = mq[4 * s4 + i];
                {
                    const char* row = pin_b + cur * (Lane4S::ROWS * 8);
                    const double2 p0 = *reinterpret_cast<const double2*>(row + rd[0]), p2 = *reinterpret_cast<const double2*>(row + rd[2]);
                    const double2 p4 = *reinterpret_cast<const double2*>(row + rd[4]), p5 = *reinterpret_cast<const double2*>(row + rd[5]);
                    const double2 p6 = *reinterpret_cast<const double2*>(row + rd[6]), p7 = *reinterpret_cast<const double2*>(row + rd[7]);
                    Pf.a[0] = p0.x; Pf.a[1] = p2.x; Pf.a[2] = p2.y; Pf.a[3] = p4.x; Pf.a[4] = p4.y; Pf.a[5] = p5.x;
                    Pf.a[6] = p6.x; Pf.a[7] = p6.y; Pf.a[8] = p7.x; Pf.a[9] = p7.y;                 // the lower triangle, like load_sym
                }
                step.step(lane, lds, mf, Pf, ms, Ps);
                if constexpr (SEL) {
                    const int slot = (int)((t - a_lo) & (SELN - 1));
                    const double m_k = pick_mean(ms), v_k = pick_var(Ps);
                    if (sel_m) selM[slot * LS::PITCH + lane] = m_k;
                    if (sel_v) selV[slot * LS::PITCH + lane] = v_k;
                    if (sel_e) selE[slot * LS::PITCH + lane] = gh_expectation(io.sel.func, m_k, sqrt(v_k), ghrule, ghrule + kGhMaxOrder, io.sel.order);
                    if (slot == 0) {
                        // SELN steps of 64 trials: lane (tsub, piece) writes the steps 2 piece, 2 piece + 1 of the trials TPI i + tsub
                        wave_lds_fence();
                        const int tsub = lane / LS::LPT, piece = lane % LS::LPT;
                        const unsigned off0 = (unsigned)tsub * rowS + (unsigned)t * 8u + (unsigned)piece * 16u;
                        const int at = (2 * piece) * LS::PITCH + tsub;
                        CGP_UNROLL for (int i = 0; i < LS::LPT; i++) {
                            if (sel_m) wSm.store2(selM[at + LS::TPI * i], selM[at + LS::PITCH + LS::TPI * i], off0 + (unsigned)(i * LS::TPI) * rowS);
                            if (sel_v) wSv.store2(selV[at + LS::TPI * i], selV[at + LS::PITCH + LS::TPI * i], off0 + (unsigned)(i * LS::TPI) * rowS);
                            if (sel_e) wSe.store2(selE[at + LS::TPI * i], selE[at + LS::PITCH + LS::TPI * i], off0 + (unsigned)(i * LS::TPI) * rowS);
                        }
                        wave_lds_fence();
                        younger += nflush;
                    }
                }
                if (want_P) {
                    double* row = tile + (lane & 31) * Lane4::PITCH_P;
                    CGP_UNROLL for (int h = 0; h < 2; h++) {
                        if ((lane >> 5) == h) {
                            CGP_UNROLL for (int c = 0; c < 5; c++) *reinterpret_cast<double2*>(row + 2 * c) = make_double2(Ps.a[2 * c], Ps.a[2 * c + 1]);
                        }
                        wave_lds_fence();
                        CGP_UNROLL for (int i = 0; i < 4; i++)
                            wP.store2(tPa[i * 8 * Lane4::PITCH_P], tPb[i * 8 * Lane4::PITCH_P], voffP + (unsigned)(4 * h + i) * 8u * rowP);
                        wave_lds_fence();
                    }
                    younger += 8;
                }
                voffP -= 128u;
                CGP_UNROLL for (int i = 0; i < D; i++) mh[4 * s4 + i] = ms.v[i];
                cur = (cur + 1) % NBUF;
            }
            if (want_m) {
                // (the passes are NOT pipelined as in the filter: at one wavefront per SIMD with all 256 registers taken, the reads held back
                // cost more than their latency -- eks 7.4 - 7.9 -> 8.15 ms per 262 144 x 500, measured)
                double* row = tile + (lane & 15) * Lane4::PITCH_M;
                CGP_UNROLL for (int h = 0; h < 4; h++) {
                    if ((lane >> 4) == h) {
                        CGP_UNROLL for (int c = 0; c < 8; c++) *reinterpret_cast<double2*>(row + 2 * c) = make_double2(mh[2 * c], mh[2 * c + 1]);
                    }
                    wave_lds_fence();
                    CGP_UNROLL for (int i = 0; i < 2; i++) {
                        const double2 v = *reinterpret_cast<const double2*>(tM + i * 8 * Lane4::PITCH_M);
                        wM.store2(v.x, v.y, voffM + (unsigned)(2 * h + i) * 8u * rowM);
                    }
                    wave_lds_fence();
                }
                younger += 8;
            }
            voffM -= 128u;
        }
        for (int64_t t = a_lo - 1; t >= 0; t--) edge(t);
    }
}

// period of the smoothers' trial groups: whole lines of four MEANS (p * T a multiple of 4); record lengths from two steps on
inline int lane4_smoother_period(int64_t T) {
    int p = 1;
    while ((p * T) % 4 != 0) p *= 2;
    return (T * p * 128 * 64 <= kOobMaxBytes) ? p : 1;
}
template <class Step>
inline hipError_t launch_lane4_smoother(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return hipSuccess;
    // covariance rows requested one step ahead (default) or, cgp_debug_set(CGP_DBG_LANE_BUFFERS, 3), two: measured at 262 144 x 500 on MI355X, no
    // gain -- full rows 8.2 -> 8.2 ms, the marginal alone 4.91 -> 4.86, E[g] alone 8.95 -> 8.85: the prefetch depth is not what limits the kernel
    const bool three = io.lane_buffers == 3;
    if (io.sel.comp >= 0) {
        // selected outputs: trials grouped by the phase of their [B][T] rows against the flushed blocks (16 or 8 doubles)
        const int nsel = (io.sel.mean ? 1 : 0) + (io.sel.var ? 1 : 0) + (io.sel.expect ? 1 : 0);
        const bool full = io.mss || io.Pss;
        // LDS per workgroup must stay within 40 KB (four workgroups a CU): a third row buffer fits beside ONE 8-step output buffer and no tile
        const bool nb3 = three && !full && nsel == 1;
        const int seln = (nsel == 1 && !nb3) ? 16 : 8;
        int period = 1;
        while ((period * io.T) % seln != 0) period *= 2;
        if (io.T * period * 128 * 64 > kOobMaxBytes) period = 1;
        const int64_t groups = (io.B + 64 * period - 1) / (64 * period);
        const unsigned grid = (unsigned)(groups * period);
        const size_t dyn = sizeof(double) * 65 * seln * nsel;
        if (full) {
            if (seln == 16) hipLaunchKernelGGL((lane4_smoother_kernel<Step, 16, 2, true>), dim3(grid), dim3(64), dyn, stream, io, ma, period);
            else hipLaunchKernelGGL((lane4_smoother_kernel<Step, 8, 2, true>), dim3(grid), dim3(64), dyn, stream, io, ma, period);
        } else if (nb3) hipLaunchKernelGGL((lane4_smoother_kernel<Step, 8, 3, false>), dim3(grid), dim3(64), dyn, stream, io, ma, period);
        else if (seln == 16) hipLaunchKernelGGL((lane4_smoother_kernel<Step, 16, 2, false>), dim3(grid), dim3(64), dyn, stream, io, ma, period);
        else hipLaunchKernelGGL((lane4_smoother_kernel<Step, 8, 2, false>), dim3(grid), dim3(64), dyn, stream, io, ma, period);
        return hipGetLastError();
    }
    const int period = lane4_smoother_period(io.T);
    const int64_t groups = (io.B + 64 * period - 1) / (64 * period);
    if (three) hipLaunchKernelGGL((lane4_smoother_kernel<Step, 0, 3, true>), dim3((unsigned)(groups * period)), dim3(64), 0, stream, io, ma, period);
    else hipLaunchKernelGGL((lane4_smoother_kernel<Step, 0, 2, true>), dim3((unsigned)(groups * period)), dim3(64), 0, stream, io, ma, period);
    return hipGetLastError();
}

}  // namespace cgp
