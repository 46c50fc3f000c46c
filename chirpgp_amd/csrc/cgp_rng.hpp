// cgp_rng.hpp -- counter-based normal variates for the on-device simulators.
//
// The reference draws with jax.random (threefry keys; tools.py:105, 148-151, tetralith/jobs/crlb_ekf.py:41-45), whose
// streams cannot be reproduced without JAX.  What the simulators need is only "independent N(0, 1) per (trial, step,
// component), reproducible, no state carried between launches or ranks"; this file provides that with
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; known-answer vectors of the
// Random123 distribution are checked on the host restatement and on the device in tests/test_gpu_sim.py):
//
//     key     = (seed low word, seed high word)
//     counter = (trial low word, trial high word, index, stream)
//     4 x u32 -> two 52-bit uniforms in (0, 1) -> Box-Muller -> two normals
//
// `trial` is the GLOBAL trial number, so a batch sharded over ranks draws exactly what one rank would.
#pragma once
#include <cstdint>
#include "cgp_fastmath.hpp"

#define CGP_HD __host__ __device__ inline

namespace cgp {

constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;

// streams (counter word 3)
constexpr uint32_t kStreamInit = 0;      // x0 = m0 + chol(P0) z          index = pair number
constexpr uint32_t kStreamState = 1;     // process noise of step k       index = k * pairs_per_step + pair
constexpr uint32_t kStreamMeas = 2;      // measurement noise             index = k / 2 (the pair serves steps 2j, 2j + 1)

struct U32x4 { uint32_t v[4]; };

CGP_HD U32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)kPhiloxM0 * c0, p1 = (uint64_t)kPhiloxM1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += kPhiloxW0; k1 += kPhiloxW1;
    }
    return U32x4{{c0, c1, c2, c3}};
}

// 52 random bits -> (0, 1): ((26 high bits of a, 26 high bits of b) + 1/2) 2^-52, exact in float64.  Never 0 or 1.
CGP_HD double uniform52(uint32_t a, uint32_t b) {
    const double x = (double)(a >> 6) * 67108864.0 + (double)(b >> 6);
    return (x + 0.5) * (1.0 / 4503599627370496.0);
}

// Two independent N(0, 1): r = sqrt(-2 ln u1), (r cos 2 pi u2, r sin 2 pi u2).
CGP_DEV void normal_pair(uint64_t seed, uint64_t trial, uint32_t index, uint32_t stream, double& z0, double& z1) {
    const U32x4 w = philox4x32_10((uint32_t)trial, (uint32_t)(trial >> 32), index, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double u1 = uniform52(w.v[0], w.v[1]), u2 = uniform52(w.v[2], w.v[3]);
    double r, ir;
    sqrt_rsqrt(-2.0 * fast_log_ge1(u1), r, ir);     // the log series is valid for every positive normal argument
    double sn, cs;
    fast_sincos(kTwoPi * u2, sn, cs);
    z0 = r * cs;
    z1 = r * sn;
}

}  // namespace cgp
