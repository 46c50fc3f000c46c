// cgp_walk4.hpp -- discrete smoothers (rts / eks / sgp_smoother, filters_smoothers.py:187-219, 317-349, 493-531) for d = 4:
// gains built lane-parallel, the recursion walked by the wavefront on the float64 matrix cores.
//
// The time-parallel smoother of cgp_kernels.hpp (tp_smoother_kernel) turns the backward recursion into a scan of affine maps:
// every lane composes 4 x 4 maps nine times per tile (three for its own four steps, six scan rounds) and applies five.  Here,
// as in the d = 6 / 8 kernel of cgp_coop8.hpp, the lanes still do the expensive, independent part of their own steps in
// parallel -- prediction (model / sigma fan) at (mf, Pf), Cholesky of Pp, the gain G = (Pp^{-1} D^T)^T: 64 chains, one per
// lane -- but nothing is composed: (G, -Pp G^T, Pf, mf - G mp) of the tile's 64 steps go to LDS (53 doubles a step, 27 KB: four
// workgroups a CU) and the wavefront walks the tile backwards COOPERATIVELY with the reference's own recursion
// (filters_smoothers.py:83-84),
//     X = Ps' - Pp,   W = X G^T,   Ps = G W + Pf,      ms = G (ms' - mp) + mf,
// the carry held in the output layout of v_mfma_f64_4x4x4_4b_f64 (lane (r, b, q): Ps[r][q]; ms in row form, lane (r, b, q):
// ms[r]).  With g = G[q][r] per lane -- which is both the B operand "G^T[k][q']" and the A operand "G[r'][k]" -- a step is
//     W  = mfma(Ps', g, N)        sum_k Ps'[k][r] G[q][k] + N,   N = -Pp G^T        (Ps' symmetric)
//     Ps = mfma(g, W, Pf)         sum_k G[r][k] W[k][q] + Pf
//     ms = mfma(g, ms', v)        sum_k G[r][k] ms'[k] + v,      v = mf - G mp
// three matrix instructions and nothing else: the two subtractions of the recursion are taken out of the serial chain by the
// lanes that build the gains (N and v are per-step constants).  The dependent chain from Ps' to Ps is two matrix instructions
// (tools/ubench/mfma_chain.hip).  What surrounds the chain is arranged not to stall it:
//   * the four LDS operands of a step are read three steps ahead (an LDS read is longer than a step);
//   * the results of FOUR steps leave in one store instruction each for Ps and ms -- the instruction's four blocks of 16
//     lanes carry the four steps' rows, 512 contiguous bytes -- because a wavefront may have only 63 memory instructions in
//     flight and two stores per step used that up (0.63 against 0.47 ms for the walk of the bench configuration);
//   * a lane's own filtering row for the NEXT tile is requested before the walk starts;
//   * for the EKS of the chirp model the gains of the next tile are built during the walk (see the kernel).
// HBM is read once and written once (320 B a step).
#pragma once
#include "cgp_coop8.hpp"

namespace cgp {

constexpr int kWalkRec = 53;                      // G 16 | N = -Pp G^T 16 | Pf 16 | v = mf - G mp 4 | pad: odd, conflict-free lane stride
constexpr int kWalkG = 0, kWalkN = 16, kWalkPf = 32, kWalkV = 48;
constexpr int kWalkAhead = 3;                     // steps between an operand's LDS read and its use

struct Walk4Operands { double g, N, Pf, v; };

template <class Elem>
__global__ void __launch_bounds__(64) walk4_smoother_kernel(SmootherIO io, ModelArgs ma) {
    static_assert(Elem::D == 4, "d = 4 kernel");
    __shared__ double recs[64 * kWalkRec];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;

    Elem elem;
    elem.setup(ma, trial);
    if constexpr (Elem::USES_SIGMA) elem.sg.stage(dyn_lds(), lane, 64, 4);
    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * 4;
    const double* __restrict__ Pfs = io.Pfs + trial * T * 16;
    double* __restrict__ mss = io.mss + trial * T * 4;
    double* __restrict__ Pss = io.Pss + trial * T * 16;
    OobWindow wPs, wms;                               // which lanes store is an offset, not a branch (cgp_coop4.hpp)
    wPs.init(Pss, T * 128); wms.init(mss, T * 32);
    // a store covers the four steps s .. s + 3: block b carries step s + b
    const unsigned offP = (unsigned)b * 128u + (unsigned)(4 * r + q) * 8u;
    const unsigned offm = (q == 0) ? (unsigned)b * 32u + (unsigned)r * 8u : kOobOffset;
    // the lane's operands inside a step's record
    const int oG = kWalkG + q * 4 + r, oN = kWalkN + r * 4 + q, oPf = kWalkPf + r * 4 + q, oV = kWalkV + r;

    // carry; the last row is copied verbatim (filters_smoothers.py:140-142)
    double Ps = Pfs[(T - 1) * 16 + ((r >= q) ? r * 4 + q : q * 4 + r)];              // the lower triangle, like the other kernels
    double ms = mfs[(T - 1) * 4 + r];
    if (lane < 16) Pss[(T - 1) * 16 + lane] = Pfs[(T - 1) * 16 + lane];
    if (lane < 4) mss[(T - 1) * 4 + lane] = mfs[(T - 1) * 4 + lane];

    // the lane's row of a tile (a row before the start of the record is clamped and not used)
    auto request = [&](int64_t base, Vec<4>& mf, Sym<4>& Pf) {
        const int64_t t = base + lane;
        load_vec<4>(mfs + (t >= 0 ? t : 0) * 4, mf);
        load_sym<4>(Pfs + (t >= 0 ? t : 0) * 16, Pf);
    };
    // the lane's record: G, N = -Pp G^T, Pf, v = mf - G mp; all zero for a step before the start of the record
    auto write_record = [&](bool valid, Mat<4>& G, Vec<4>& mp, Sym<4>& Pp, Vec<4>& mf, Sym<4>& Pf) {
        if (!valid) {
            CGP_UNROLL for (int a = 0; a < 4; a++) { mp.v[a] = 0.0; mf.v[a] = 0.0; CGP_UNROLL for (int c = 0; c < 4; c++) G.a[a][c] = 0.0; }
            CGP_UNROLL for (int a = 0; a < Sym<4>::N; a++) { Pp.a[a] = 0.0; Pf.a[a] = 0.0; }
        }
        double* mine = recs + lane * kWalkRec;
        CGP_UNROLL for (int a = 0; a < 4; a++) {
            double v = mf.v[a];
            CGP_UNROLL for (int c = 0; c < 4; c++) {
                double n = 0.0;                                             // -(Pp G^T)[a][c] = -sum_k Pp[a][k] G[c][k]
                CGP_UNROLL for (int k = 0; k < 4; k++) n = fma(-Pp(a, k), G.a[c][k], n);
                mine[kWalkG + a * 4 + c] = G.a[a][c];
                mine[kWalkN + a * 4 + c] = n;
                mine[kWalkPf + a * 4 + c] = Pf(a, c);                       // Sym::operator() is symmetric in its arguments
                v = fma(-G.a[a][c], mp.v[c], v);                            // mf - G mp
            }
            mine[kWalkV + a] = v;
        }
    };
    // the wavefront walks the tile in LDS from its last step to its first.  Steps before the start of the record (last tile)
    // are walked too: their records are zero and their stores fall outside the windows (the step index wraps).
    auto walk = [&](int64_t base) {
        auto fetch = [&](int s, Walk4Operands& o) {
            const double* p = recs + (s & 63) * kWalkRec;
            o.g = p[oG]; o.N = p[oN]; o.Pf = p[oPf]; o.v = p[oV];
        };
        Walk4Operands ring[kWalkAhead + 1];
        CGP_UNROLL for (int a = 0; a < kWalkAhead; a++) fetch(63 - a, ring[a]);
        double P4[4], m4[4];                                               // the results of the four steps of a store
        CGP_UNROLL for (int s = 63; s >= 0; s--) {                         // fully unrolled: the ring is register renaming
            fetch(s - kWalkAhead, ring[(63 - s + kWalkAhead) % (kWalkAhead + 1)]);       // (the records fetched past the tile's first step are not used)
            const Walk4Operands& cur = ring[(63 - s) % (kWalkAhead + 1)];
            const double W = mfma4x4(Ps, cur.g, cur.N);                    // (Ps' - Pp) G^T
            ms = mfma4x4(cur.g, ms, cur.v);                                // G (ms' - mp) + mf
            Ps = mfma4x4(cur.g, W, cur.Pf);                                // G W + Pf
            P4[s & 3] = Ps; m4[s & 3] = ms;
            if ((s & 3) == 0) {
                // block b takes step s + b: bank-masked moves (a DPP bank is a block), which -- unlike a select on b -- the
                // compiler cannot turn into divergent branches that would cut the walk into sixteen basic blocks
                constexpr int kSame = 0xE4;                                 // quad_perm:[0,1,2,3]
                const double Pv = dpp_banks_f64<kSame, 0x8>(dpp_banks_f64<kSame, 0x4>(dpp_banks_f64<kSame, 0x2>(P4[0], P4[1]), P4[2]), P4[3]);
                const double mv = dpp_banks_f64<kSame, 0x8>(dpp_banks_f64<kSame, 0x4>(dpp_banks_f64<kSame, 0x2>(m4[0], m4[1]), m4[2]), m4[3]);
                const unsigned step = (unsigned)(base + s);
                wPs.store(Pv, offP + step * 128u);
                wms.store(mv, offm + step * 32u);
            }
        }
    };

    Vec<4> mf; Sym<4> Pf;
    if constexpr (Elem::HAS_SPEC) {
        // The gains of tile n + 1 are built in the same basic block as the walk of tile n: as straight-line code without regime
        // branches (gain_spec) they are free to move around the walk's serial chain of matrix instructions (the compiler puts
        // the prediction and the factorisation in front of it and the solves behind it; a finer interleaving spelt out with
        // sched_group_barrier was not taken up).  A lane outside the regime (rare) has its gain rebuilt by the checked form
        // after the walk.  Rows are requested two tiles ahead.  0.79 -> 0.73 ms on the bench configuration.
        {
            Mat<4> G; Vec<4> mp; Sym<4> Pp;
            request(T - 2 - 63, mf, Pf);
            const bool valid = T - 2 - 63 + lane >= 0;
            if (valid) elem.gain(mf, Pf, G, mp, Pp);
            write_record(valid, G, mp, Pp, mf, Pf);
            wave_lds_fence();
            request(T - 2 - 127, mf, Pf);
        }
        for (int64_t hi = T - 2; hi >= 0; hi -= 64) {
            const int64_t base = hi - 63;                                  // step of lane 0 (may be negative in the last tile)
            Vec<4> mf2; Sym<4> Pf2;
            request(base - 128, mf2, Pf2);                                 // two tiles ahead: used at the end of the next iteration
            Mat<4> G; Vec<4> mp; Sym<4> Pp; bool ok;
            elem.gain_spec(mf, Pf, G, mp, Pp, ok);                         // tile n + 1 (on the clamped row where it does not exist)
            walk(base);                                                    // tile n
            wave_lds_fence();
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) elem.gain(mf, Pf, G, mp, Pp);
            write_record(base - 64 + lane >= 0, G, mp, Pp, mf, Pf);
            wave_lds_fence();
            mf = mf2; Pf = Pf2;
        }
    } else {
        request(T - 2 - 63, mf, Pf);
        for (int64_t hi = T - 2; hi >= 0; hi -= 64) {
            const int64_t base = hi - 63;                                  // step of lane 0 (may be negative in the last tile)
            // ---- every lane: prediction and gain of its own step, then its record
            {
                Mat<4> G; Vec<4> mp; Sym<4> Pp;
                const bool valid = base + lane >= 0;
                if (valid) elem.gain(mf, Pf, G, mp, Pp);
                write_record(valid, G, mp, Pp, mf, Pf);
            }
            wave_lds_fence();
            request(base - 64, mf, Pf);                                    // the NEXT (earlier) tile's row: requested now, used after the walk
            walk(base);
            wave_lds_fence();
        }
    }
}

// One workgroup holds 64 records (27 136 B) beside the staged sigma-point set; four of them have to share a CU's 160 KB.
inline bool walk4_smoother_ok(int64_t T, const ModelArgs& ma) {
    return T * 128 <= kOobMaxBytes && sigma_lds_bytes(ma, 4) + sizeof(double) * 64 * kWalkRec + 64 <= 40 * 1024;
}
template <class Elem>
inline hipError_t launch_walk4_smoother(const SmootherIO& io, const ModelArgs& ma, hipStream_t stream) {
    if (io.B <= 0 || io.T <= 0) return hipSuccess;
    if (!walk4_smoother_ok(io.T, ma)) return hipErrorInvalidValue;
    const size_t dyn = Elem::USES_SIGMA ? sigma_lds_bytes(ma, 4) : 0;
    hipLaunchKernelGGL((walk4_smoother_kernel<Elem>), dim3((unsigned)io.B), dim3(64), dyn, stream, io, ma);
    return hipGetLastError();
}

}  // namespace cgp
