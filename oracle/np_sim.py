"""CPU restatement of the on-device simulators (chirpgp_amd/csrc/cgp_rng.hpp, cgp_simulate.hpp).  TEST INFRASTRUCTURE.

What is simulated follows the reference:

* ``simulate``   -- chirpgp/tools.py:119-170 (``simulate_sde``: x0 = m0 + chol(P0) z, x <- m(x) + chol(cov(x)) dw) with
  the measurement line of tetralith/jobs/crlb_ekf.py:52-55 (y = H . x + sqrt(Xi) e);
* ``add_noise``  -- demos/ekfs_mle.py:33-35 (ys = chirp + sqrt(Xi) N(0, 1)).

The random numbers cannot follow the reference (jax.random streams need JAX): the engine defines its own counter-based
streams -- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), two 52-bit
uniforms per block, Box-Muller -- and this file restates them bit for bit (integers) / to rounding (log, sqrt, sincos).
The Philox restatement is pinned by the known-answer vectors of the Random123 distribution (tests/test_oracle_sim.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
STREAM_INIT, STREAM_STATE, STREAM_MEAS = 0, 1, 2


def philox4x32_10(ctr, key):
    """ctr (..., 4) uint32-valued, key (2,) -> (..., 4) uint32.  Ten rounds, key bumped by the Weyl constants."""
    ctr = np.asarray(ctr)
    c = [ctr[..., i].astype(np.uint64) & MASK for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    s32 = np.uint64(32)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> s32) ^ c[1] ^ np.uint64(k0), p1 & MASK, (p0 >> s32) ^ c[3] ^ np.uint64(k1), p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def uniform52(a, b):
    x = (np.asarray(a, dtype=np.uint64) >> np.uint64(6)).astype(np.float64) * 67108864.0 \
        + (np.asarray(b, dtype=np.uint64) >> np.uint64(6)).astype(np.float64)
    return (x + 0.5) * (1.0 / 4503599627370496.0)


def normal_pairs(seed, trial, index, stream):
    """Two N(0, 1) per (trial, index); trial, index broadcast against each other.  Returns (z0, z1)."""
    trial, index = np.broadcast_arrays(np.asarray(trial, dtype=np.uint64), np.asarray(index, dtype=np.uint64))
    ctr = np.stack([trial & MASK, trial >> np.uint64(32), index & MASK, np.full(trial.shape, stream, dtype=np.uint64)], axis=-1)
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u1, u2 = uniform52(w[..., 0], w[..., 1]), uniform52(w[..., 2], w[..., 3])
    r = np.sqrt(-2.0 * np.log(u1))
    ang = (2.0 * np.pi) * u2
    return r * np.cos(ang), r * np.sin(ang)


def _normals(seed, trial, first_index, n, stream):
    """n consecutive normals of one trial starting at pair `first_index` (pairs laid out (z0, z1), (z0, z1), ...)."""
    npairs = (n + 1) // 2
    z0, z1 = normal_pairs(seed, trial, first_index + np.arange(npairs), stream)
    return np.stack([z0, z1], axis=-1).reshape(-1)[:n]


def _chol(A):
    try:
        return np.linalg.cholesky(A)
    except np.linalg.LinAlgError:          # jnp.linalg.cholesky returns NaN instead of raising
        return np.full_like(A, np.nan)


def simulate(cond_m_cov, H, Xi, m0, P0, dt, T, seed, trials):
    """xs (B, T, d), ys (B, T) for the global trial numbers in `trials`."""
    m0, P0 = np.asarray(m0, dtype=np.float64), np.asarray(P0, dtype=np.float64)
    d = m0.size
    npairs = (d + 1) // 2
    trials = list(trials)
    xs, ys = np.empty((len(trials), T, d)), np.empty((len(trials), T))
    L0 = _chol(P0)
    for b, tr in enumerate(trials):
        x = m0 + L0 @ _normals(seed, tr, 0, d, STREAM_INIT)
        e0, e1 = normal_pairs(seed, tr, np.arange((T + 1) // 2), STREAM_MEAS)
        e = np.stack([e0, e1], axis=-1).reshape(-1)
        zs = _normals(seed, tr, 0, 2 * npairs * T, STREAM_STATE).reshape(T, 2 * npairs)
        for k in range(T):
            m, cov = cond_m_cov(x, dt)
            x = m + _chol(np.asarray(cov)) @ zs[k, :d]
            xs[b, k] = x
            ys[b, k] = np.dot(H, x) + np.sqrt(Xi) * e[k]
    return xs, ys


def add_noise(clean, Xi, seed, trials, T):
    """ys (B, T) = clean + sqrt(Xi) e with e the measurement-noise stream."""
    trials = np.asarray(list(trials), dtype=np.uint64)
    z0, z1 = normal_pairs(seed, trials[:, None], np.arange((T + 1) // 2)[None, :], STREAM_MEAS)
    e = np.stack([z0, z1], axis=-1).reshape(trials.size, -1)[:, :T]
    return np.asarray(clean)[..., :T] + np.sqrt(np.asarray(Xi, dtype=np.float64)).reshape(-1, 1) * e
