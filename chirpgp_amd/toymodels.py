"""Synthetic chirp generators either side of the hot path (chirpgp/toymodels.py), host side, NumPy.

Same function names and return conventions as the reference; random magnitudes take a NumPy Generator instead of a JAX
PRNG key (the jax.random streams cannot be reproduced without JAX).  ``tiled_meow`` extends the reference's benchmark
frequency law past its (0, pi) domain for the long records of BASELINE.json."""
import math
import numpy as np

__all__ = ['gen_chirp', 'gen_harmonic_chirp', 'gen_chirp_envelope', 'constant_mag', 'damped_exp_mag', 'random_ou_mag', 'affine_freq',
           'polynomial_freq', 'meow_freq', 'tiled_meow', 'noisy_copies']


def gen_chirp(ts, magnitude_func, phase_func, base_phase=0.):
    """alpha(t) sin(phi_0 + 2 pi phi(t))  (toymodels.py:37-70)."""
    return magnitude_func(ts) * np.sin(base_phase + 2 * math.pi * phase_func(ts))


def gen_harmonic_chirp(ts, magnitude_funcs, fundamental_phase_func, base_phase=0.):
    """sum_i alpha_i(t) sin(phi_0 + i 2 pi phi(t))  (toymodels.py:73-104)."""
    ph = fundamental_phase_func(ts)
    return sum(mag(ts) * np.sin(base_phase + (i + 1) * 2 * math.pi * ph) for i, mag in enumerate(magnitude_funcs))


def gen_chirp_envelope(ts, magnitude_func, phase_func, base_phase=0.):
    """Complex chirp alpha(t) exp(i (phi_0 + 2 pi phi(t)))  (toymodels.py:107-120)."""
    return magnitude_func(ts) * np.exp(1j * (base_phase + 2 * math.pi * phase_func(ts)))


def constant_mag(b):
    return lambda ts: np.full_like(np.asarray(ts, dtype=np.float64), b)


def damped_exp_mag(damp_rate):
    return lambda ts: np.exp(-damp_rate * np.asarray(ts))


def random_ou_mag(ell, sigma, rng):
    """One Ornstein-Uhlenbeck realisation as the magnitude (toymodels.py:144-167)."""
    def generate(ts):
        ts = np.asarray(ts, dtype=np.float64)
        a = math.exp(-(ts[1] - ts[0]) / ell)
        q = sigma * math.sqrt(1 - a * a)
        out = np.empty(ts.size)
        x = sigma * rng.standard_normal()
        for k in range(ts.size):
            x = a * x + q * rng.standard_normal()
            out[k] = x
        return out
    return generate


def affine_freq(a, b):
    return (lambda ts: a * ts + b), (lambda ts: 0.5 * a * ts ** 2 + b * ts)


def polynomial_freq(coeffs):
    """Frequency sum_k c_k t^k and its phase (toymodels.py:193-223)."""
    def freq(ts):
        return sum(c * np.asarray(ts) ** k for k, c in enumerate(coeffs))

    def phase(ts):
        return sum(c / (k + 1) * np.asarray(ts) ** (k + 1) for k, c in enumerate(coeffs))
    return freq, phase


def meow_freq(mag=500., scale=5., offset=5.5):
    """Frequency a b cot(t) csc(t) exp(-b csc t) + c and phase a exp(-b / sin t) + c t on (0, pi)  (toymodels.py:226-268)."""
    def freq(ts):
        ts = np.asarray(ts)
        return mag * scale * np.cos(ts) / np.sin(ts) ** 2 * np.exp(-scale / np.sin(ts)) + offset

    def phase(ts):
        ts = np.asarray(ts)
        return mag * np.exp(-scale / np.sin(ts)) + offset * ts
    return freq, phase


def tiled_meow(T, dt=1e-3, mag=500., scale=5., offset=8., window=3141):
    """(ts, freq, phase) of the meow law repeated in `window`-step tiles; continuous in phase and frequency at the seams."""
    k = np.arange(T)
    local = (k % window + 1) * dt
    f, p = meow_freq(mag, scale, offset)
    return (k + 1) * dt, f(local), (k // window) * (offset * window * dt) + p(local)


def noisy_copies(clean, Xi, key, batch, trial0=0):
    """(batch, T) CUDA tensor of clean + sqrt(Xi) N(0, 1) -- the Monte-Carlo measurements of demos/ekfs_mle.py:33-35 drawn
    on the device (include/chirpgp_hip.h: cgp_add_noise), so only the T clean samples cross PCIe.  ``clean`` is (T,)
    or (batch, T); trial ``trial0 + i`` draws the same noise in every call with the same integer ``key``."""
    from chirpgp_amd import _engine as E
    return E.add_noise(clean, Xi, key, int(batch), trial0=trial0)
