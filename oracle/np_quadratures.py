"""NumPy restatement of /root/reference/chirpgp/quadratures.py (TEST INFRASTRUCTURE).

SigmaPoints (cubature, Gauss-Hermite), the RK4 step on the (mean, cov) pair and
the 1-D Gauss-Hermite E[g(V)] post-processing step.
"""
import math
from typing import NamedTuple, Optional
import numpy as np
from oracle.np_models import g

__all__ = ['rk4_m_cov', 'rk4_m_cov_backward', 'SigmaPoints', 'gaussian_expectation']


def rk4_m_cov(m_cov_ode, m, P, dt):
    """Classic RK4 on (m, P); quadratures.py:34-54 (same operation order: dt * k / 2)."""
    k1m, k1P = m_cov_ode(m, P)
    k2m, k2P = m_cov_ode(m + dt * k1m / 2, P + dt * k1P / 2)
    k3m, k3P = m_cov_ode(m + dt * k2m / 2, P + dt * k2P / 2)
    k4m, k4P = m_cov_ode(m + dt * k3m, P + dt * k3P)
    return (m + dt * (k1m + 2 * k2m + 2 * k3m + k4m) / 6,
            P + dt * (k1P + 2 * k2P + 2 * k3P + k4P) / 6)


def rk4_m_cov_backward(m_cov_ode, m, P, mf, Pf, dt):
    """quadratures.py:57-81; (mf, Pf) threaded unchanged through the 4 stages."""
    k1m, k1P = m_cov_ode(m, P, mf, Pf)
    k2m, k2P = m_cov_ode(m + dt * k1m / 2, P + dt * k1P / 2, mf, Pf)
    k3m, k3P = m_cov_ode(m + dt * k2m / 2, P + dt * k2P / 2, mf, Pf)
    k4m, k4P = m_cov_ode(m + dt * k3m, P + dt * k3P, mf, Pf)
    return (m + dt * (k1m + 2 * k2m + 2 * k3m + k4m) / 6,
            P + dt * (k1P + 2 * k2P + 2 * k3P + k4P) / 6)


def _physicists_hermite(order):
    """Coefficients (highest power first) of H_0..H_order by H_n = 2x H_{n-1} - 2(n-1) H_{n-2};
    quadratures.py:112-136."""
    polys = [np.array([1.]), np.array([2., 0.])]
    for n in range(2, order + 1):
        polys.append(2 * np.append(polys[n - 1], 0.)
                     - 2 * (n - 1) * np.concatenate([np.zeros(2), polys[n - 2]]))
    return polys


class SigmaPoints(NamedTuple):
    """quadratures.py:84-231: (d, n_points, w (s,), wc, xi (s, d)); chi_i = m + chol(P) xi_i."""
    d: int
    n_points: int
    w: np.ndarray
    wc: Optional[np.ndarray]
    xi: np.ndarray

    @classmethod
    def cubature(cls, d):
        """quadratures.py:138-150: 2d points, equal weights, xi = sqrt(d) [I; -I]."""
        s = 2 * d
        return cls(d=d, n_points=s, w=np.ones(s) / s, wc=None,
                   xi=math.sqrt(d) * np.concatenate([np.eye(d), -np.eye(d)], axis=0))

    @classmethod
    def unscented(cls, d, alpha, beta, lam):
        """quadratures.py:152-154."""
        raise NotImplementedError('Unscented transform is not implemented.')

    @classmethod
    def gauss_hermite(cls, d, order=3):
        """quadratures.py:156-196.  Tensor grid of order**d points, dimension 0 varying fastest;
        1-D nodes = roots of the physicists' Hermite polynomial (np.roots, flipped), scaled by sqrt(2)."""
        polys = _physicists_hermite(order)
        roots = np.flip(np.roots(polys[order]))
        w1 = np.array([2 ** (order - 1) * math.factorial(order) * math.sqrt(math.pi)
                       / (order ** 2 * np.polyval(polys[order - 1], roots[i]) ** 2) for i in range(order)])
        s = order ** d
        n = np.arange(s)
        table = np.stack([(n // order ** r) % order for r in range(d)], axis=0)  # (d, s)
        w = (1 / (math.sqrt(math.pi) ** d)) * np.prod(w1[table], axis=0)
        xi = (math.sqrt(2) * roots[table]).T
        return cls(d=d, n_points=s, w=np.asarray(w, dtype=np.float64), wc=None,
                   xi=np.asarray(np.real(xi), dtype=np.float64))

    def gen_sigma_points(self, m, chol_of_P):
        """quadratures.py:198-201."""
        return m + np.einsum('ij,...j->...i', chol_of_P, self.xi)

    def expectation(self, evals):
        """quadratures.py:218-231: sum_i w_i z_i."""
        return np.einsum('i,i...->...', self.w, evals)

    def expectation_from_nodes(self, v_f, chi):
        """quadratures.py:203-216."""
        return np.einsum('i,i...->...', self.w, v_f(chi))


def gaussian_expectation(ms, chol_Ps, func=g, d=1, order=10, force_shape=False):
    """E[func(V_t)] for T Gaussian marginals by Gauss-Hermite; quadratures.py:234-274."""
    ms = np.asarray(ms, dtype=np.float64)
    chol_Ps = np.asarray(chol_Ps, dtype=np.float64)
    if force_shape:
        ms = ms.reshape(-1, 1)
        chol_Ps = chol_Ps.reshape(-1, 1, 1)
    sgps = SigmaPoints.gauss_hermite(d=d, order=order)
    out = []
    for m, chol in zip(ms, chol_Ps):
        chi = sgps.gen_sigma_points(m, chol)
        out.append(sgps.expectation_from_nodes(func, chi))
    return np.asarray(out)
