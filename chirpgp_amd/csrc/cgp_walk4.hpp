// cgp_walk4.hpp -- discrete smoothers (rts / eks / sgp_smoother, filters_smoothers.py:187-219, 317-349, 493-531) for d = 4:
// per-step affine maps built lane-parallel, composed and applied by the wavefront on the float64 matrix cores, FOUR time steps
// per matrix instruction.
//
// The backward recursion (filters_smoothers.py:83-84) is an affine map of the carry whose coefficients depend on the filtering
// results only:   ms_t = G_t ms_{t+1} + c_t,   Ps_t = G_t Ps_{t+1} G_t^T + C_t,   c_t = mf_t - G_t mp_t,   C_t = Pf_t - G_t Pp_t G_t^T.
// A tile is 64 consecutive steps of one trial:
//   phase G  every lane builds the map of ITS step -- prediction (model / sigma fan) at (mf, Pf), Cholesky of Pp, the gain
//            G = (Pp^{-1} D^T)^T, then c and C: 64 independent chains on the vector ALU -- and parks (G, C, c) in LDS
//            (36 doubles a step, 18.9 KB a tile: eight workgroups a CU);
//   phase S  the wavefront replaces the maps of every quad of steps (4 Q .. 4 Q + 3) by their SUFFIX compositions -- step
//            4 Q + j gets f_j o f_{j+1} o .. o f_3, the map from the carry ENTERING the quad to that step -- three rounds of
//            four matrix instructions for sixteen quads at a time: v_mfma_f64_4x4x4_4b_f64 multiplies four independent
//            4 x 4 blocks, one quad per block, so this part is throughput-bound, not a dependent chain;
//   phase W  the wavefront walks the tile backwards one QUAD at a time: block b of the instruction applies the composed map of
//            step 4 Q + b to the same carry,
//                W  = mfma(Ps'', g, 0)     Ps'' M_b^T             (g = M_b[q][r]: serves as B operand "M^T" and as A operand "M")
//                Ps = mfma(g, W, C_b)      M_b W + C_b
//                ms = mfma(g, ms'', c_b)   M_b ms'' + c_b
//            -- three matrix instructions for FOUR steps (the step-by-step walk of round 2 needed twelve), whose four blocks
//            are the four consecutive output rows: one 512-byte store for Ps and one for ms per quad, no re-arrangement.
//            Block 0 (the earliest step) is the next quad's carry: two bank-masked DPP broadcasts (the instruction's
//            CBSZ / ABID block broadcast is ignored for the float64 4x4x4 shape on gfx950: tools/ubench/mfma_bcast.hip).
// The dependent chain is two matrix instructions and a broadcast per FOUR steps.  LDS operands are read ahead of their use,
// a lane's filtering rows are requested one (EKS of the chirp model: two) tiles ahead, and for that model the next tile's
// gains are straight-line code (gain_spec) in the walk's basic block.  HBM is read once and written once (320 B a step).
//
// Small batches: the time-split form.  With B << 1024 trials one wavefront per trial leaves most SIMDs idle while every
// trial walks its T steps alone.  The maps compose exactly, so a record can be cut into `segs` segments handled by different
// wavefronts (nothing is approximated):
//   pass 1 (MODE = kWalkCompose)  phases G and S as above, then the QUAD maps (step 4 Q's suffix composition = the whole quad)
//                                 are chained into the segment's map (A, c, C) -- 36 doubles to the workspace; no stores;
//   pass 2 (MODE = kWalkApply)    every wave applies the maps of the later segments to the record's last filtering row --
//                                 its own carry-in, a handful of matrix instructions -- and walks its segment as above.
// A trial then takes 2 T / segs steps instead of T; at B = 125 (BASELINE C3 / C5 sharded over 8 GPUs) segs = 16.
#pragma once
#include <atomic>
#include "cgp_coop8.hpp"

namespace cgp {

constexpr int kWalkRec = 37;                      // G 16 (row-major) | C 16 (full) | c 4 | pad: odd, conflict-free lane stride
constexpr int kWalkG = 0, kWalkC = 16, kWalkc = 32;
constexpr int kWalkAhead = 2;                     // quads between an operand's LDS read and its use
constexpr int kWalkMapDoubles = 40;               // a segment's composed map in the workspace: A 16 | C 16 | c 4 | pad
struct Walk4Operands { double g, C, c; };

// block 0's value in all four blocks (same r, q): blocks move inside a DPP row with row rotations and bank masks
CGP_DEV double blk_bcast0(double x) {
    return dpp_banks_f64<kRowRor12, 0x8>(dpp_banks_f64<kRowRor8, 0x4>(dpp_banks_f64<kRowRor4, 0x2>(x, x), x), x);
}

// SEL (cgp_smoother_select): the walk also leaves the selected component's smoothed mean and variance of every step of the tile in
// LDS (two ds_write per quad, by the lanes that hold them); behind the walk every lane finishes ITS step -- the marginal itself and / or
// E[f(V)] by 1-D Gauss-Hermite, 64 steps at a time, one coalesced 512-byte store per output and tile.  mss / Pss may be NULL then
// (a window of zero bytes drops their stores).
template <class Elem, int MODE = kWalkWhole, bool SEL = false>
__global__ void __launch_bounds__(64) walk4_smoother_kernel(SmootherIO io, ModelArgs ma) {
    static_assert(Elem::D == 4, "d = 4 kernel");
    static_assert(!(SEL && MODE == kWalkCompose), "pass 1 of the time-split form writes nothing");
    __shared__ double recs[64 * kWalkRec];
    __shared__ double selbuf[SEL ? 128 : 1];
    __shared__ double ghrule[SEL ? 2 * kGhMaxOrder : 1];
    const int lane = threadIdx.x;
    const int r = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const int64_t trial = (MODE == kWalkWhole) ? (int64_t)blockIdx.x : (int64_t)(blockIdx.x / (unsigned)io.segs);
    const int seg = (MODE == kWalkWhole) ? 0 : (int)(blockIdx.x % (unsigned)io.segs);
    if (trial >= io.B) return;

    Elem elem;
    elem.setup(ma, trial);
    if constexpr (SEL) { sel_stage_rule(io.sel, ghrule, lane); wave_lds_fence(); }
    if constexpr (Elem::USES_SIGMA) elem.sg.stage(dyn_lds(), lane, 64, 4);
    const int64_t T = io.T;
    const double* __restrict__ mfs = io.mfs + trial * T * 4;
    const double* __restrict__ Pfs = io.Pfs + trial * T * 16;
    double* __restrict__ mss = io.mss + trial * T * 4;
    double* __restrict__ Pss = io.Pss + trial * T * 16;
    OobWindow wPs, wms;                               // which lanes store is an offset, not a branch (cgp_coop4.hpp)
    wPs.init((MODE == kWalkCompose || !io.Pss) ? nullptr : Pss, T * 128); wms.init((MODE == kWalkCompose || !io.mss) ? nullptr : mss, T * 32);   // pass 1 stores nothing
    // the segment's tiles: tile j covers the steps T - 2 - 64 j - 63 .. T - 2 - 64 j (segment 0 is the LAST in time)
    const int64_t hi_first = (MODE == kWalkWhole) ? T - 2 : T - 2 - 64 * (int64_t)seg * io.tiles_per_seg;
    const int64_t hi_stop = (MODE == kWalkWhole) ? 0 : max((int64_t)0, hi_first - 64 * (int64_t)io.tiles_per_seg + 1);   // tiles with hi >= hi_stop
    if (hi_first < 0 && MODE != kWalkWhole) return;   // (more segments than tiles: nothing to do, nothing to compose)
    // a quad's store covers the four steps 4 Q .. 4 Q + 3: block b carries step 4 Q + b
    const unsigned offP = (unsigned)b * 128u + (unsigned)(4 * r + q) * 8u;
    const unsigned offm = (q == 0) ? (unsigned)b * 32u + (unsigned)r * 8u : kOobOffset;
    // the lane's operands inside a step's record: M as A operand / as B^T (g = M[q][r]), M, C as B operand ([r][q]), c in row form
    const int oGa = kWalkG + q * 4 + r, oGb = kWalkG + r * 4 + q, oC = kWalkC + r * 4 + q, oc = kWalkc + r;

    // carry (the same in the four blocks); the last row is copied verbatim (filters_smoothers.py:140-142)
    double Ps = Pfs[(T - 1) * 16 + ((r >= q) ? r * 4 + q : q * 4 + r)];              // the lower triangle, like the other kernels
    double ms = mfs[(T - 1) * 4 + r];
    double Acc = (r == q) ? 1.0 : 0.0;                // pass 1: the composed linear part, A <- M A
    if constexpr (MODE == kWalkCompose) { Ps = 0.0; ms = 0.0; }
    if (MODE != kWalkCompose && seg == 0) {
        if (lane < 16 && (!SEL || io.Pss)) Pss[(T - 1) * 16 + lane] = Pfs[(T - 1) * 16 + lane];
        if (lane < 4 && (!SEL || io.mss)) mss[(T - 1) * 4 + lane] = mfs[(T - 1) * 4 + lane];
        if constexpr (SEL) { if (lane == 0) sel_write(io.sel, ghrule, trial * T + T - 1, mfs[(T - 1) * 4 + io.sel.comp], Pfs[(T - 1) * 16 + io.sel.comp * 5]); }
    }
    const bool sel_var_lane = SEL && r == io.sel.comp && q == io.sel.comp, sel_mean_lane = SEL && r == io.sel.comp && q == 0;
    if constexpr (MODE == kWalkApply) {
        // carry-in of this segment: the maps of the segments later in time, applied in order to the last filtering row
        const double* __restrict__ maps = io.ws + trial * io.segs * kWalkMapDoubles;
        for (int s2 = 0; s2 < seg; s2++) {
            const double* __restrict__ mp_ = maps + s2 * kWalkMapDoubles;
            const double gA = mp_[q * 4 + r], Cs = mp_[16 + r * 4 + q], cs = mp_[32 + r];
            const double Wc = mfma4x4(Ps, gA, 0.0);
            ms = mfma4x4(gA, ms, cs);
            Ps = mfma4x4(gA, Wc, Cs);
        }
    }

    // the lane's row of a tile (a row before the start of the record is clamped and not used)
    auto request = [&](int64_t base, Vec<4>& mf, Sym<4>& Pf) {
        const int64_t t = base + lane;
        load_vec<4>(mfs + (t >= 0 ? t : 0) * 4, mf);
        load_sym<4>(Pfs + (t >= 0 ? t : 0) * 16, Pf);
    };
    // the lane's record: (G, C, c) of its step; all zero for a step before the start of the record
    auto write_record = [&](bool valid, const Mat<4>& G, const Vec<4>& c, const Sym<4>& C) {
        double* mine = recs + lane * kWalkRec;
        CGP_UNROLL for (int a = 0; a < 4; a++) {
            CGP_UNROLL for (int k = 0; k < 4; k++) {
                mine[kWalkG + a * 4 + k] = valid ? G.a[a][k] : 0.0;
                mine[kWalkC + a * 4 + k] = valid ? C(a, k) : 0.0;           // Sym::operator() is symmetric in its arguments
            }
            mine[kWalkc + a] = valid ? c.v[a] : 0.0;
        }
    };
    // phase S: in every quad, step 4 Q + j <- f_j o f_{j+1} o .. o f_3; block b of group i handles quad 4 i + b
    auto scan = [&]() {
        CGP_UNROLL for (int i = 3; i >= 0; i--) {                          // (the walk starts with the last quads)
            const double* p3 = recs + (16 * i + 4 * b + 3) * kWalkRec;
            double M = p3[oGb], C = p3[oC], c = p3[oc];
            CGP_UNROLL for (int j = 2; j >= 0; j--) {
                double* pj = recs + (16 * i + 4 * b + j) * kWalkRec;
                const double g = pj[oGa], Cj = pj[oC], cj = pj[oc];
                M = mfma4x4(g, M, 0.0);                                    // G_j M
                c = mfma4x4(g, c, cj);                                     // G_j c + c_j                 (row form)
                const double Tt = mfma4x4(C, g, 0.0);                      // C G_j^T = (G_j C)^T         (C symmetric)
                C = mfma4x4(Tt, g, Cj);                                    // (G_j C) G_j^T + C_j
                pj[oGb] = M; pj[oC] = C; pj[oc] = c;                       // (the four lanes q of a row write the same c)
            }
        }
    };
    // phase W: the tile in LDS from its last quad to its first.  Steps before the start of the record (last tile) are walked
    // too: their records are zero and their stores fall outside the windows (the step index wraps).
    auto walk = [&](int64_t base) {
        auto fetch = [&](int Q, Walk4Operands& o) {
            const double* p = recs + ((4 * Q + (MODE == kWalkCompose ? 0 : b)) & 63) * kWalkRec;
            o.g = p[oGa]; o.C = p[oC]; o.c = p[oc];
        };
        Walk4Operands ring[kWalkAhead + 1];
        CGP_UNROLL for (int a = 0; a < kWalkAhead; a++) fetch(15 - a, ring[a]);
        CGP_UNROLL for (int Q = 15; Q >= 0; Q--) {                         // fully unrolled: the ring is register renaming
            fetch(Q - kWalkAhead, ring[(15 - Q + kWalkAhead) % (kWalkAhead + 1)]);       // (the records fetched past the tile's first quad are not used)
            const Walk4Operands& cur = ring[(15 - Q) % (kWalkAhead + 1)];
            if constexpr (MODE == kWalkCompose) {
                // the quad's whole map (step 4 Q) onto the segment's: every block does the same
                Acc = mfma4x4(cur.g, Acc, 0.0);
                ms = mfma4x4(cur.g, ms, cur.c);
                const double Tt = mfma4x4(Ps, cur.g, 0.0);
                Ps = mfma4x4(cur.g, Tt, cur.C);
            } else {
                const double W = mfma4x4(Ps, cur.g, 0.0);                  // Ps'' M_b^T
                const double msn = mfma4x4(cur.g, ms, cur.c);              // M_b ms'' + c_b
                const double Psn = mfma4x4(cur.g, W, cur.C);               // M_b W + C_b
                const unsigned step = (unsigned)(base + 4 * Q);
                wPs.store(Psn, offP + step * 128u);
                wms.store(msn, offm + step * 32u);
                if constexpr (SEL) {
                    if (sel_var_lane) selbuf[64 + 4 * Q + b] = Psn;
                    if (sel_mean_lane) selbuf[4 * Q + b] = msn;
                }
                Ps = blk_bcast0(Psn);                                      // block 0 = step 4 Q: the next quad's carry
                ms = blk_bcast0(msn);
            }
        }
    };

    Vec<4> mf; Sym<4> Pf;
    if constexpr (Elem::HAS_SPEC) {
        // The maps of tile n + 1 are built in the same basic block as phases S and W of tile n: as straight-line code without
        // regime branches (gain_spec) they are free to move around the matrix instructions.  A lane outside the regime
        // (rare) has its gain rebuilt by the checked form afterwards.  Rows are requested two tiles ahead.
        {
            Mat<4> G; Vec<4> c; Sym<4> C;
            request(hi_first - 63, mf, Pf);
            const bool valid = hi_first - 63 + lane >= 0;
            if (valid) elem.map(mf, Pf, G, c, C);
            write_record(valid, G, c, C);
            wave_lds_fence();
            request(hi_first - 127, mf, Pf);
        }
        for (int64_t hi = hi_first; hi >= hi_stop; hi -= 64) {
            const int64_t base = hi - 63;                                  // step of lane 0 (may be negative in the last tile)
            Vec<4> mf2; Sym<4> Pf2;
            request(base - 128, mf2, Pf2);                                 // two tiles ahead: used at the end of the next iteration
            Mat<4> G; Vec<4> c; Sym<4> C; bool ok;
            elem.map_spec(mf, Pf, G, c, C, ok);                            // tile n + 1 (on the clamped row where it does not exist)
            scan();                                                        // tile n
            wave_lds_fence();
            walk(base);
            wave_lds_fence();
            if constexpr (SEL) { if (base + lane >= 0) sel_write(io.sel, ghrule, trial * T + base + lane, selbuf[lane], selbuf[64 + lane]); }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) elem.map(mf, Pf, G, c, C);
            write_record(base - 64 + lane >= 0, G, c, C);
            wave_lds_fence();
            mf = mf2; Pf = Pf2;
        }
    } else {
        request(hi_first - 63, mf, Pf);
        for (int64_t hi = hi_first; hi >= hi_stop; hi -= 64) {
            const int64_t base = hi - 63;                                  // step of lane 0 (may be negative in the last tile)
            // ---- every lane: the map of its own step, then its record
            {
                Mat<4> G; Vec<4> c; Sym<4> C;
                const bool valid = base + lane >= 0;
                if (valid) elem.map(mf, Pf, G, c, C);
                write_record(valid, G, c, C);
            }
            wave_lds_fence();
            request(base - 64, mf, Pf);                                    // the NEXT (earlier) tile's row: requested now, used after the walk
            scan();
            wave_lds_fence();
            walk(base);
            wave_lds_fence();
            if constexpr (SEL) { if (base + lane >= 0) sel_write(io.sel, ghrule, trial * T + base + lane, selbuf[lane], selbuf[64 + lane]); }
        }
    }
    if constexpr (MODE == kWalkCompose) {
        double* __restrict__ out = io.ws + (trial * io.segs + seg) * kWalkMapDoubles;
        if (b == 0) {
            out[r * 4 + q] = Acc;
            out[16 + r * 4 + q] = Ps;
            if (q == 0) out[32 + r] = ms;
        }
    }
}

// One workgroup holds 64 records (18 944 B) beside the staged sigma-point set; at least four of them have to share a CU's 160 KB.
inline bool walk4_smoother_ok(int64_t T, const ModelArgs& ma) {
    return T * 128 <= kOobMaxBytes && sigma_lds_bytes(ma, 4) + sizeof(double) * 64 * kWalkRec + 64 <= 40 * 1024;
}
template <class Elem>
inline hipError_t launch_walk4_smoother(const SmootherIO& io_in, const ModelArgs& ma, hipStream_t stream) {
    if (io_in.B <= 0 || io_in.T <= 0) return hipSuccess;
    if (!walk4_smoother_ok(io_in.T, ma)) return hipErrorInvalidValue;
    const size_t dyn = Elem::USES_SIGMA ? sigma_lds_bytes(ma, 4) : 0;
    // workgroups of the split kernel a CU holds (registers, LDS): asked once per kernel and dynamic-LDS size, then remembered
    static std::atomic<long long> occ_cache{-1};                           // (dyn << 8) | per_cu
    int per_cu = 0;
    if (io_in.segs != 1) {
        const long long seen = occ_cache.load(std::memory_order_relaxed);
        if (seen >= 0 && (size_t)(seen >> 8) == dyn) per_cu = (int)(seen & 0xFF);
        else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, walk4_smoother_kernel<Elem, kWalkApply>, 64, dyn) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
            occ_cache.store(((long long)dyn << 8) | (per_cu & 0xFF), std::memory_order_relaxed);
        }
    }
    const int segs = walk_segments(io_in, per_cu);
    const bool sel = io_in.sel.comp >= 0;
    auto whole = [&]() {
        if (sel) hipLaunchKernelGGL((walk4_smoother_kernel<Elem, kWalkWhole, true>), dim3((unsigned)io_in.B), dim3(64), dyn, stream, io_in, ma);
        else hipLaunchKernelGGL((walk4_smoother_kernel<Elem, kWalkWhole>), dim3((unsigned)io_in.B), dim3(64), dyn, stream, io_in, ma);
        return hipGetLastError();
    };
    if (segs <= 1) return whole();
    // time-split: two passes with the segments' maps in the context's per-stream workspace (nothing the caller sees)
    SmootherIO io = io_in;
    io.segs = segs;
    const int64_t tiles = (io.T - 1 + 63) / 64;
    io.tiles_per_seg = (int)((tiles + io.segs - 1) / io.segs);
    io.segs = (int)((tiles + io.tiles_per_seg - 1) / io.tiles_per_seg);           // no empty segments
    void* ws = ctx_workspace(io.host_ctx, stream, sizeof(double) * kWalkMapDoubles * (size_t)io.B * io.segs);
    hipError_t e;
    if (!ws) return whole();             // no workspace (allocation failed, pinned too small, growth inside a graph capture): the one-wave-per-trial form needs none
    io.ws = (double*)ws;
    const unsigned grid = (unsigned)(io.B * io.segs);
    hipLaunchKernelGGL((walk4_smoother_kernel<Elem, kWalkCompose>), dim3(grid), dim3(64), dyn, stream, io, ma);
    if (sel) hipLaunchKernelGGL((walk4_smoother_kernel<Elem, kWalkApply, true>), dim3(grid), dim3(64), dyn, stream, io, ma);
    else hipLaunchKernelGGL((walk4_smoother_kernel<Elem, kWalkApply>), dim3(grid), dim3(64), dyn, stream, io, ma);
    e = hipGetLastError();
    return e;
}

}  // namespace cgp
