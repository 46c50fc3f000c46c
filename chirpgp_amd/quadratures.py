"""Sigma-point sets for the MI355X engine -- host side of chirpgp/quadratures.py.

The point sets are tiny host constants (81 x 4 doubles for Gauss-Hermite order 3 in d = 4) built once in NumPy
and uploaded; everything that uses them per step -- chi = m + chol(P) xi, the model evaluations, the weighted
sums, the RK4 stages -- runs inside the HIP kernels (csrc/cgp_steps.hpp).  ``gaussian_expectation`` is the
post-smoother step of every driver (demos/ekfs_mle.py:73-75) and runs on the device as well.
"""
import math
from typing import NamedTuple, Optional
import numpy as np

__all__ = ['rk4_m_cov', 'rk4_m_cov_backward', 'SigmaPoints', 'gaussian_expectation']


def _rk4_pair(rhs, m, P, dt):
    """Classic fourth-order Runge-Kutta step of size dt for the coupled pair (m, P); rhs(m, P) -> (dm, dP)."""
    am, aP = rhs(m, P)
    bm, bP = rhs(m + 0.5 * dt * am, P + 0.5 * dt * aP)
    cm, cP = rhs(m + 0.5 * dt * bm, P + 0.5 * dt * bP)
    em, eP = rhs(m + dt * cm, P + dt * cP)
    return m + dt * (am + 2 * bm + 2 * cm + em) / 6, P + dt * (aP + 2 * bP + 2 * cP + eP) / 6


def rk4_m_cov(m_cov_ode, m, P, dt):
    """One RK4 step of the moment ODEs (quadratures.py:34-54), host side, for user code that integrates its own
    ``m_cov_ode(m, P) -> (dm, dP)``.  The continuous-discrete filters run their RK4 inside the HIP kernels
    (csrc/cgp_steps.hpp) and do not call this."""
    return _rk4_pair(m_cov_ode, np.asarray(m, dtype=np.float64), np.asarray(P, dtype=np.float64), dt)


def rk4_m_cov_backward(m_cov_ode, m, P, mf, Pf, dt):
    """The smoother's variant (quadratures.py:57-81): the ODE also sees the filtering result (mf, Pf), held fixed over
    the four stages."""
    return _rk4_pair(lambda a, b: m_cov_ode(a, b, mf, Pf), np.asarray(m, dtype=np.float64), np.asarray(P, dtype=np.float64), dt)


def _hermite_physicists(order):
    """Coefficient arrays (highest degree first) of H_0 .. H_order."""
    hs = [np.array([1.]), np.array([2., 0.])]
    while len(hs) <= order:
        n = len(hs)
        hs.append(2. * np.append(hs[-1], 0.) - 2. * (n - 1) * np.pad(hs[-2], (2, 0)))
    return hs


class SigmaPoints(NamedTuple):
    """Same fields as the reference's SigmaPoints (quadratures.py:84-110): d, n_points, w (s,), wc, xi (s, d)."""
    d: int
    n_points: int
    w: np.ndarray
    wc: Optional[np.ndarray]
    xi: np.ndarray

    @classmethod
    def cubature(cls, d: int):
        """Spherical cubature rule: 2d points +-sqrt(d) e_i with equal weights (quadratures.py:138-150)."""
        eye = np.eye(d)
        return cls(d, 2 * d, np.full(2 * d, 1. / (2 * d)), None, math.sqrt(d) * np.vstack([eye, -eye]))

    @classmethod
    def unscented(cls, d: int, alpha: float, beta: float, lam: float):
        """Not implemented in the reference either (quadratures.py:152-154)."""
        raise NotImplementedError('Unscented transform is not implemented.')

    @classmethod
    def gauss_hermite(cls, d: int, order: int = 3):
        """Tensor-product Gauss-Hermite rule with order**d points (quadratures.py:156-196).

        Point n has 1-D node index (n // order**r) % order along dimension r, so dimension 0 varies fastest;
        the 1-D nodes are the Hermite roots in the order np.flip(np.roots(.)) gives, as in the reference.
        """
        hs = _hermite_physicists(order)
        nodes = np.real(np.flip(np.roots(hs[order])))
        w1 = (2. ** (order - 1) * math.factorial(order) * math.sqrt(math.pi)
              / (order ** 2 * np.polyval(hs[order - 1], nodes) ** 2))
        idx = (np.arange(order ** d)[None, :] // (order ** np.arange(d))[:, None]) % order      # (d, s)
        w = np.prod(w1[idx], axis=0) / math.sqrt(math.pi) ** d
        return cls(d, order ** d, np.ascontiguousarray(w), None, np.ascontiguousarray(math.sqrt(2.) * nodes[idx].T))

    def gen_sigma_points(self, m, chol_of_P):
        """chi_i = m + chol(P) xi_i (host helper; the kernels do this per step on the device)."""
        return np.asarray(m) + np.asarray(self.xi) @ np.asarray(chol_of_P).T

    def expectation(self, evals_of_integrand):
        return np.tensordot(self.w, np.asarray(evals_of_integrand), axes=(0, 0))


def identity(x):
    """The identity as an integrand of gaussian_expectation (a named function, so that the device path can recognise it)."""
    return x


def gaussian_expectation(ms, chol_Ps, func=None, d: int = 1, order: int = 10, force_shape: bool = False):
    """E[func(V_t)] for scalar Gaussian marginals by 1-D Gauss-Hermite, on the device (quadratures.py:234-274).

    d = 1 only (the reference's own use: "in this chirp application the dimension of V is 1", :257-259).  `func` is one of the
    integrands the kernel knows: ``models.g`` (the softplus every driver passes, demos/ekfs_mle.py:73-75; also the default),
    ``numpy.exp`` (the reference's test of this function, test/test_utils.py:84-95), ``numpy.square``, ``quadratures.identity`` --
    or their names 'g', 'exp', 'square', 'identity'.  Any other callable raises: it cannot run inside a HIP kernel and there is
    no CPU fallback.  Returns an array of shape (T, 1) like the reference.
    """
    from chirpgp_amd import models as _models
    from chirpgp_amd import _engine
    if d != 1:
        raise NotImplementedError('the device kernel implements d = 1 (the reference drivers\' only use)')
    known = {None: _engine.FN_SOFTPLUS, _models.g: _engine.FN_SOFTPLUS, 'g': _engine.FN_SOFTPLUS, np.exp: _engine.FN_EXP, 'exp': _engine.FN_EXP,
             np.square: _engine.FN_SQUARE, 'square': _engine.FN_SQUARE, identity: _engine.FN_IDENTITY, 'identity': _engine.FN_IDENTITY}
    try:
        fn = known[func]
    except (KeyError, TypeError):
        raise NotImplementedError('func must be one of models.g, numpy.exp, numpy.square, quadratures.identity (or their names): an '
                                  'arbitrary Python callable cannot run inside the HIP kernel') from None
    sg = SigmaPoints.gauss_hermite(1, order)
    return _engine.gaussian_expectation(ms, chol_Ps, sg.xi[:, 0], sg.w, fn)
