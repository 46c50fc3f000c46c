// cgp_simulate.hpp -- Monte-Carlo trajectories and measurements generated in HBM (the step before the filters).
//
//   x_0 = m0 + chol(P0) z,   x_k = mean(x_{k-1}) + chol(Sigma) dw_k,   y_k = H . x_k + sqrt(Xi) e_k,   k = 1..T
//
// is chirpgp/tools.py:148-167 (simulate_sde) with the measurement line of tetralith/jobs/crlb_ekf.py:52-55.  Sigma of
// every discrete model of the path does not depend on the state, so its Cholesky factor is formed once
// per trial.  Random numbers: cgp_rng.hpp.
//
// Launch shapes, as for the filters:
//   one wavefront per trial  -- the 64 lanes draw and colour the noise of 64 consecutive steps at once (the expensive
//                               part: Philox + log + sincos per pair); the recursion itself runs replicated in all
//                               lanes and picks its increment by v_readlane; rows are parked in LDS and leave with
//                               one coalesced store per 64 steps.
//   one lane per trial       -- everything per lane; rows leave through LDS transposes so that stores cover whole lines.
#pragma once
#include "cgp_kernels.hpp"
#include "cgp_rng.hpp"

namespace cgp {

struct SimIO {
    const double* __restrict__ H;  int64_t H_stride;
    const double* __restrict__ Xi; int64_t Xi_stride;
    const double* __restrict__ m0; int64_t m0_stride;
    const double* __restrict__ P0; int64_t P0_stride;
    uint64_t seed;
    int64_t trial0;
    int64_t B, T;
    double* __restrict__ xs;
    double* __restrict__ ys;
    uint32_t vec_ok;       // bit 0: xs rows may be stored as 16-byte chunks, bit 1: ys rows
    uint32_t flags;
};

// y += L z, L lower-triangular packed
template <int D> CGP_DEV void add_lower_matvec(const Sym<D>& L, const double (&z)[D + (D & 1)], Vec<D>& y) {
    CGP_UNROLL for (int i = 0; i < D; i++) {
        double s = y.v[i];
        CGP_UNROLL for (int j = 0; j <= i; j++) s = fma(L(i, j), z[j], s);
        y.v[i] = s;
    }
}

template <int D> CGP_DEV void draw_state_noise(uint64_t seed, uint64_t gtrial, uint32_t first_index, uint32_t stream,
                                               double (&z)[D + (D & 1)]) {
    CGP_UNROLL for (int p = 0; p < (D + 1) / 2; p++) normal_pair(seed, gtrial, first_index + p, stream, z[2 * p], z[2 * p + 1]);
}

template <class DM> struct SimSetup {
    static constexpr int D = DM::D;
    DM model;
    Sym<D> LQ;
    Vec<D> H, x;
    double sqrt_xi;
    CGP_DEV void init(const SimIO& io, const ModelArgs& ma, int64_t trial) {
        model.setup(ma.params + trial * ma.param_stride, ma.dt, ma.model_id);
        Sym<D> Sigma, P0, L0;
        Vec<D> inv;
        CGP_UNROLL for (int i = 0; i < Sym<D>::N; i++) Sigma.a[i] = 0.0;
        model.add_sigma(Sigma, 1.0);
        cholesky<D>(Sigma, LQ, inv);
        if (io.H) load_vec<D>(io.H + trial * io.H_stride, H);
        else { CGP_UNROLL for (int i = 0; i < D; i++) H.v[i] = 0.0; }
        const double xi = io.Xi ? io.Xi[trial * io.Xi_stride] : 0.0;
        double r, ir;
        sqrt_rsqrt(xi, r, ir);
        sqrt_xi = xi == 0.0 ? 0.0 : r;
        load_vec<D>(io.m0 + trial * io.m0_stride, x);
        if (!(io.flags & CGP_SIM_FIXED_X0)) {
            load_sym<D>(io.P0 + trial * io.P0_stride, P0);
            cholesky<D>(P0, L0, inv);
            double z[D + (D & 1)];
            draw_state_noise<D>(io.seed, (uint64_t)(io.trial0 + trial), 0u, kStreamInit, z);
            add_lower_matvec<D>(L0, z, x);
        }
    }
};

constexpr int kSimPairs(int d) { return (d + 1) / 2; }

// ---------------------------------------------------------------------------------------------- one wavefront per trial
template <class DM>
__global__ void __launch_bounds__(64) simulate_wave_kernel(SimIO io, ModelArgs ma) {
    constexpr int D = DM::D, NP = kSimPairs(D), ROW = D + 1;
    __shared__ double park[64 * ROW];
    const int lane = threadIdx.x;
    const int64_t trial = blockIdx.x;
    if (trial >= io.B) return;
    SimSetup<DM> s;
    s.init(io, ma, trial);
    Vec<D> x = s.x;
    const uint64_t gtrial = (uint64_t)(io.trial0 + trial);
    const int64_t T = io.T;
    double* __restrict__ xs = io.xs ? io.xs + trial * T * D : nullptr;
    double* __restrict__ ys = io.ys ? io.ys + trial * T : nullptr;
    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        const int nsteps = (T - t0 < 64) ? (int)(T - t0) : 64;
        // lane l: the coloured increments of step t0 + l
        const int64_t k = t0 + lane;
        double z[D + (D & 1)];
        draw_state_noise<D>(io.seed, gtrial, (uint32_t)(k * NP), kStreamState, z);
        Vec<D> e;
        CGP_UNROLL for (int i = 0; i < D; i++) e.v[i] = 0.0;
        add_lower_matvec<D>(s.LQ, z, e);
        double za, zb;
        normal_pair(io.seed, gtrial, (uint32_t)(k >> 1), kStreamMeas, za, zb);
        const double ey = s.sqrt_xi * ((k & 1) ? zb : za);
        for (int slot = 0; slot < nsteps; slot++) {
            Vec<D> f;
            s.model.mean(x, f);
            double y = readlane_f64(ey, slot);
            CGP_UNROLL for (int i = 0; i < D; i++) {
                x.v[i] = f.v[i] + readlane_f64(e.v[i], slot);
                y = fma(s.H.v[i], x.v[i], y);
            }
            if (lane == 0) {
                CGP_UNROLL for (int i = 0; i < D; i++) park[slot * ROW + i] = x.v[i];
                park[slot * ROW + D] = y;
            }
        }
        wave_lds_fence();
        if (lane < nsteps) {
            if (xs) { CGP_UNROLL for (int i = 0; i < D; i++) xs[(t0 + lane) * D + i] = park[lane * ROW + i]; }
            if (ys) ys[t0 + lane] = park[lane * ROW + D];
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------- one lane per trial
// Rows of N doubles per trial, `ncols` of them filled, parked in LDS at pitch N + 2 by the owning lane; written out so
// that consecutive lanes cover consecutive addresses of a trial's row.
template <int N>
CGP_DEV void sim_flush(const double* tile, int lane, double* __restrict__ out_block, int64_t trial_stride, int nvalid,
                       int ncols, bool vec_ok) {
    constexpr int P = N + 2;
    wave_lds_fence();
    bool done = false;
    if constexpr (N % 2 == 0) {
        if (vec_ok && ncols == N) {
            CGP_UNROLL for (int k = 0; k < N / 2; k++) {
                const int g = k * 64 + lane, tr = g / (N / 2), ch = g % (N / 2);
                const double2 v = *reinterpret_cast<const double2*>(tile + tr * P + 2 * ch);
                if (tr < nvalid) *reinterpret_cast<double2*>(out_block + tr * trial_stride + 2 * ch) = v;
            }
            done = true;
        }
    }
    if (!done) {
        CGP_UNROLL for (int k = 0; k < N; k++) {
            const int g = k * 64 + lane, tr = g / N, c = g % N;
            if (tr < nvalid && c < ncols) out_block[tr * trial_stride + c] = tile[tr * P + c];
        }
    }
    wave_lds_fence();
}

// steps of x rows gathered per flush: about 16 doubles, an even number of doubles where possible
constexpr int kSimStepsPerFlush(int d) {
    int sx = 16 / d;
    if (sx < 1) sx = 1;
    if (((d * sx) & 1) && sx > 1) sx--;
    return sx;
}

template <class DM>
__global__ void __launch_bounds__(64) simulate_lane_kernel(SimIO io, ModelArgs ma) {
    constexpr int D = DM::D, NP = kSimPairs(D), SX = kSimStepsPerFlush(D), NX = SX * D, NY = 16;
    __shared__ __attribute__((aligned(16))) double xtile[64 * (NX + 2)];
    __shared__ __attribute__((aligned(16))) double ytile[64 * (NY + 2)];
    const int lane = threadIdx.x;
    const int64_t block_first = (int64_t)blockIdx.x * 64;
    if (block_first >= io.B) return;
    const int nvalid = (io.B - block_first < 64) ? (int)(io.B - block_first) : 64;
    int64_t trial = block_first + lane;
    if (trial >= io.B) trial = io.B - 1;      // idle lanes redo the last trial and store nothing
    SimSetup<DM> s;
    s.init(io, ma, trial);
    Vec<D> x = s.x;
    const uint64_t gtrial = (uint64_t)(io.trial0 + trial);
    const int64_t T = io.T;
    double zb = 0.0;
    for (int64_t t = 0; t < T; t++) {
        double z[D + (D & 1)];
        draw_state_noise<D>(io.seed, gtrial, (uint32_t)(t * NP), kStreamState, z);
        Vec<D> f;
        s.model.mean(x, f);
        x = f;
        add_lower_matvec<D>(s.LQ, z, x);
        double zy;
        if ((t & 1) == 0) normal_pair(io.seed, gtrial, (uint32_t)(t >> 1), kStreamMeas, zy, zb);
        else zy = zb;
        double y = s.sqrt_xi * zy;
        CGP_UNROLL for (int i = 0; i < D; i++) y = fma(s.H.v[i], x.v[i], y);
        const bool last = t == T - 1;
        if (io.xs) {
            const int sub = (int)(t % SX);
            CGP_UNROLL for (int i = 0; i < D; i++) xtile[lane * (NX + 2) + sub * D + i] = x.v[i];
            if (sub == SX - 1 || last)
                sim_flush<NX>(xtile, lane, io.xs + (block_first * T + (t - sub)) * D, T * D, nvalid, (sub + 1) * D, (io.vec_ok & 1) != 0);
        }
        if (io.ys) {
            const int sub = (int)(t % NY);
            ytile[lane * (NY + 2) + sub] = y;
            if (sub == NY - 1 || last)
                sim_flush<NY>(ytile, lane, io.ys + block_first * T + (t - sub), T, nvalid, sub + 1, (io.vec_ok & 2) != 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------- measurement noise only
// One thread per pair of consecutive steps of one trial; consecutive threads walk along T (coalesced).
__global__ void __launch_bounds__(256) add_noise_kernel(const double* __restrict__ clean, int64_t clean_stride,
                                                        const double* __restrict__ Xi, int64_t Xi_stride, uint64_t seed,
                                                        int64_t trial0, int64_t B, int64_t T, double* __restrict__ ys) {
    const int64_t pairs = (T + 1) / 2, total = B * pairs;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / pairs, j = g - b * pairs;
        double za, zb, r, ir;
        normal_pair(seed, (uint64_t)(trial0 + b), (uint32_t)j, kStreamMeas, za, zb);
        const double xi = Xi[b * Xi_stride];
        sqrt_rsqrt(xi, r, ir);
        const double sx = xi == 0.0 ? 0.0 : r;
        const double* c = clean + b * clean_stride;
        double* o = ys + b * T;
        o[2 * j] = fma(sx, za, c[2 * j]);
        if (2 * j + 1 < T) o[2 * j + 1] = fma(sx, zb, c[2 * j + 1]);
    }
}

__global__ void __launch_bounds__(256) debug_philox_kernel(const uint32_t* __restrict__ ctr, const uint32_t* __restrict__ key,
                                                           int64_t n, uint32_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const U32x4 w = philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[0], key[1]);
        CGP_UNROLL for (int q = 0; q < 4; q++) out[4 * i + q] = w.v[q];
    }
}

template <class DM>
inline int launch_simulate(bool wave, const SimIO& io, const ModelArgs& ma, hipStream_t st) {
    if (wave) hipLaunchKernelGGL((simulate_wave_kernel<DM>), dim3((unsigned)io.B), dim3(64), 0, st, io, ma);
    else hipLaunchKernelGGL((simulate_lane_kernel<DM>), dim3((unsigned)((io.B + 63) / 64)), dim3(64), 0, st, io, ma);
    return hip_rc(hipGetLastError());
}

int dispatch_simulate(int model_id, int key, bool wave, const SimIO&, const ModelArgs&, hipStream_t);

}  // namespace cgp
