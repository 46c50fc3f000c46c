"""Model descriptors for the MI355X engine -- host side of the reference's chirpgp/models.py.

The reference hands its filters Python closures (``cond_m_cov(u, dt)``, ``a(u)``, ``b(u)``, ``h(x)``) that JAX traces
and differentiates (filters_smoothers.py:255, 382).  A HIP kernel cannot call a closure, so the same builder
functions here return *descriptor objects*: they are still callable (NumPy, for simulating data or inspecting a
model on the host) but what the filters consume is ``(model_id, d, n_harm, params)``, which is marshalled to the
C-ABI (include/chirpgp_hip.h) where the model and its analytic Jacobian are evaluated on the device.

Names, argument order and return tuples follow chirpgp/models.py.  Every ``params`` entry may carry a leading
batch axis (one parameter vector per trial: parameter sweeps / MLE grids).
"""
import math
import numpy as np

__all__ = ['g', 'g_inv', 'DiscreteModel', 'DriftModel', 'Dispersion', 'MeasurementKPT', 'CustomDiscrete', 'CustomDrift', 'CustomMeasurement',
           'custom_cond_m_cov', 'custom_sde', 'custom_measurement',
           'linear_cond_m_cov', 'linear_sde',
           'model_chirp', 'model_harmonic_chirp', 'model_lascala',
           'disc_chirp_lcd', 'disc_harmonic_chirp_lcd', 'disc_model_lascala_lcd', 'disc_m32',
           'build_chirp_model', 'build_harmonic_chirp_model', 'build_lascala_model', 'build_kpt_chirp_model']

# model ids of include/chirpgp_hip.h
M_LINEAR, M_HARMONIC_LCD, M_LASCALA_LCD, M_LINEAR_SDE, M_HARMONIC_SDE, M_KPT = range(6)


def g(x):
    """Positive bijection log(exp(x) + 1) (models.py:50), in the reference's naive form: +inf above 709.78 as there."""
    with np.errstate(over='ignore'):
        return np.log(np.exp(x) + 1.)


def g_inv(x):
    """models.py:53."""
    return np.log(np.exp(x) - 1.)


def _f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _m32(ell, sigma, dt):
    """Closed-form Matern-3/2 discretisation (models.py:61-73), host evaluation."""
    gamma = math.sqrt(3) / ell
    eta = dt * gamma
    beta = sigma ** 2 * math.exp(-2 * eta)
    F = np.array([[1 + eta, dt], [-dt * gamma ** 2, 1 - eta]]) * math.exp(-eta)
    off = 2 * dt ** 2 * gamma ** 3 * beta
    S = np.array([[sigma ** 2 - beta * (2 * eta + 2 * eta ** 2 + 1), off],
                  [off, gamma ** 2 * (sigma ** 2 + beta * (2 * eta - 2 * eta ** 2 - 1))]])
    return F, S


def _blkdiag(blocks):
    blocks = [np.atleast_2d(b) for b in blocks]
    n = sum(b.shape[0] for b in blocks)
    out = np.zeros((n, n))
    k = 0
    for b in blocks:
        out[k:k + b.shape[0], k:k + b.shape[0]] = b
        k += b.shape[0]
    return out


class _Spec:
    model_id = -1

    def __init__(self, d, n_harm, params):
        self.d = int(d)
        self.n_harm = int(n_harm)
        self.params = _f64(params)
        self.gamma = None

    @property
    def batched(self):
        return self.params.ndim == 2

    def _single(self):
        if self.batched:
            raise ValueError('host evaluation of a batched model descriptor is not defined; index its params first')
        return self.params

    def __repr__(self):
        return f'{type(self).__name__}(id={self.model_id}, d={self.d}, n_harm={self.n_harm}, params{self.params.shape})'


class DiscreteModel(_Spec):
    """Descriptor of a ``cond_m_cov(u, dt) -> (mean, cov)`` callable."""

    def __init__(self, model_id, d, n_harm, params):
        super().__init__(d, n_harm, params)
        self.model_id = model_id

    def __call__(self, u, dt):
        p, d = self._single(), self.d
        u = np.asarray(u)
        if self.model_id == M_LINEAR:
            return p[:d * d].reshape(d, d) @ u, p[d * d:].reshape(d, d)
        if self.model_id == M_LASCALA_LCD:
            lam, b, ell, sigma, fs = 0., 0., p[0], p[1], 1.
        else:
            lam, b, ell, sigma, fs = p
        w = 2 * math.pi * g(u[-2]) * fs
        rho = math.exp(-lam * dt)
        blocks = []
        for k in range(1, self.n_harm + 1):
            c, s = np.cos(dt * k * w), np.sin(dt * k * w)
            blocks.append(np.array([[c, -s], [s, c]]) * rho)
        Fm, Sm = _m32(ell, sigma, dt)
        q = b ** 2 * dt if lam == 0. else b ** 2 / (2 * lam) * (1 - math.exp(-2 * lam * dt))
        return _blkdiag(blocks + [Fm]) @ u, _blkdiag([q] * (2 * self.n_harm) + [Sm])


class DriftModel(_Spec):
    """Descriptor of an SDE drift ``a(u)``."""

    def __init__(self, model_id, d, n_harm, params):
        super().__init__(d, n_harm, params)
        self.model_id = model_id

    def __call__(self, u):
        p, d = self._single(), self.d
        u = np.asarray(u)
        if self.model_id == M_LINEAR_SDE:
            return p.reshape(d, d) @ u
        lam, ell, fs = p
        gam = math.sqrt(3) / ell
        w = 2 * math.pi * g(u[-2]) * fs
        blocks = [np.array([[-lam, -w * k], [w * k, -lam]]) for k in range(1, self.n_harm + 1)]
        return _blkdiag(blocks + [np.array([[0., 1.], [-(gam ** 2), -2 * gam]])]) @ u


class Dispersion:
    """Constant dispersion ``b(u) -> (d, dw)`` matrix (every model of the path has a state-independent one)."""

    def __init__(self, matrix):
        self.matrix = _f64(matrix)

    def __call__(self, _=None):
        return self.matrix

    def outer(self):
        """gamma = b b^T, with a leading batch axis if the matrix has one."""
        return self.matrix @ np.swapaxes(self.matrix, -1, -2)


class MeasurementKPT:
    """h(x) = sum_k x[k] sin(k g(x[0] + x[-1])) of the KPT model (models.py:575-578)."""

    def __init__(self, num_harmonics):
        self.n_harm = int(num_harmonics)

    def __call__(self, x):
        x = np.asarray(x)
        ks = np.arange(1, self.n_harm + 1)
        return np.dot(x[1:-1], np.sin(g(x[0] + x[-1]) * ks))


# --------------------------------------------------------------------------- models compiled at run time
M_CUSTOM = -1
CUSTOM_DISCRETE, CUSTOM_SDE, CUSTOM_MEASUREMENT = 0, 1, 2          # include/chirpgp_hip.h


class _CustomSpec(_Spec):
    """A model handed over as DEVICE SOURCE (include/chirpgp_hip.h: cgp_model_from_source; csrc/cgp_custom.hpp): compiled by ROCm's
    runtime compiler into the generic kernels on first use, Jacobian by forward-mode dual numbers in the kernel."""
    model_id = M_CUSTOM
    kind = None

    def __init__(self, source, d, params, host=None):
        super().__init__(d, 0, params)
        self.source = str(source)
        self.host = host

    def __repr__(self):
        return f'{type(self).__name__}(d={self.d}, params{self.params.shape}, {len(self.source)} characters of source)'


class CustomDiscrete(_CustomSpec):
    """cond_m_cov(u, dt) -> (mean, cov) as source: `cond_mean<T>` and `cond_cov` (see custom_cond_m_cov)."""
    kind = CUSTOM_DISCRETE

    def __call__(self, u, dt):
        if self.host is None:
            raise TypeError('this custom model has no host callable (pass host= to custom_cond_m_cov to evaluate it with NumPy)')
        return self.host(u, dt)


class CustomDrift(_CustomSpec):
    """SDE drift a(u) as source: `drift<T>` (see custom_sde)."""
    kind = CUSTOM_SDE

    def __call__(self, u):
        if self.host is None:
            raise TypeError('this custom drift has no host callable (pass host= to custom_sde to evaluate it with NumPy)')
        return self.host(u)


class CustomMeasurement(_CustomSpec):
    """ekf_for_kpt's measurement function h(u) as source: `measure<T>` (see custom_measurement).  `params` here is the vector q the source
    reads (d doubles at most; one row per trial if batched); the linear dynamics (F, Sigma) arrive with the call."""
    kind = CUSTOM_MEASUREMENT

    def __init__(self, source, d, q=None, host=None):
        q = np.zeros(d) if q is None else _f64(q)
        if q.shape[-1] > d:
            raise ValueError(f'the measurement function takes at most d = {d} parameters (they travel in the slot of H)')
        pad = np.zeros(q.shape[:-1] + (d,))
        pad[..., :q.shape[-1]] = q
        super().__init__(source, d, pad, host)

    def __call__(self, u):
        if self.host is None:
            raise TypeError('this custom measurement has no host callable (pass host= to custom_measurement to evaluate it with NumPy)')
        return self.host(u)

    def with_dynamics(self, F, Sigma):
        """The spec the engine launches: this measurement over lambda u, dt: (F @ u, Sigma) -- params [F | Sigma], q in the slot of H."""
        lin = linear_cond_m_cov(F, Sigma)
        if lin.d != self.d:
            raise ValueError(f'F must be {self.d} x {self.d}')
        spec = CustomMeasurement.__new__(CustomMeasurement)
        _CustomSpec.__init__(spec, self.source, self.d, lin.params, self.host)
        spec.q = self.params
        return spec


def custom_measurement(source, d, q=None, host=None):
    """Descriptor of a measurement function the library does not enumerate -- what the reference's ``ekf_for_kpt`` takes as any traceable
    scalar ``h(u)`` (filters_smoothers.py:267-314: ``H = jacfwd(h)(mp)``, ``pred = h(mp)``).  ``source`` is HIP device code defining

        template <class T> __device__ T measure(const T* u, const double* q);

    for a generic scalar type T (double or the kernel's dual numbers, as for custom_cond_m_cov); ``q`` (at most d doubles, (n,) or (B, n))
    is handed to it per trial; ``host`` an optional NumPy callable ``u -> h(u)``.  d <= 8."""
    return CustomMeasurement(source, d, q, host)


def custom_cond_m_cov(source, d, params, host=None):
    """Descriptor of a discrete model the library does not enumerate -- what the reference takes as any traceable ``cond_m_cov(u, dt)``
    (filters_smoothers.py:222-264, 317-349, 446-531): usable with ``ekf``, ``eks``, ``sgp_filter`` and ``sgp_smoother`` (the covariance is
    evaluated at every sigma point, as the reference's ``_sgp_prediction`` does).  ``source`` is HIP device code defining, for a generic scalar
    type T (double, or the dual numbers the kernel differentiates with: sin, cos, exp, log, sqrt, tanh, pow(x, const) and softplus are
    overloaded),

        template <class T> __device__ void cond_mean(const T* u, const double* p, double dt, T* mean);
        __device__ void cond_cov(const double* u, const double* p, double dt, double* cov);      // cov: [d][d] row-major, symmetric

    ``params`` (n,) or (B, n) is the vector ``p`` the source reads (one per trial if batched); ``host`` an optional NumPy callable
    ``(u, dt) -> (mean, cov)`` of the same model for host-side use (simulation, checking).  d <= 8."""
    return CustomDiscrete(source, d, params, host)


def custom_sde(source, d, params, dispersion, host=None):
    """(drift, dispersion) descriptors of an SDE model as source -- ``cd_ekf`` / ``cd_eks`` / ``cd_sgp_filter`` / ``cd_sgp_smoother``
    (filters_smoothers.py:352-443, 534-632):

        template <class T> __device__ void drift(const T* u, const double* p, T* a);

    ``dispersion`` is the constant (d, dw) matrix b (the kernels take b b^T)."""
    return CustomDrift(source, d, params, host), Dispersion(dispersion)


# --------------------------------------------------------------------------- generic linear descriptors
def linear_cond_m_cov(F, Sigma):
    """Descriptor of ``lambda u, dt: (F @ u, Sigma)`` (the linear test model of test_filters_smoothers.py:56-58)."""
    F, Sigma = _f64(F), _f64(Sigma)
    d = F.shape[-1]
    lead = np.broadcast_shapes(F.shape[:-2], Sigma.shape[:-2])          # either may carry the batch axis
    params = np.concatenate([np.broadcast_to(F, lead + (d, d)).reshape(lead + (d * d,)),
                             np.broadcast_to(Sigma, lead + (d, d)).reshape(lead + (d * d,))], axis=-1)
    return DiscreteModel(M_LINEAR, d, 0, params)


def linear_sde(A, B):
    """Descriptors of ``drift = lambda u: A @ u`` and ``dispersion = lambda _: B``."""
    A = _f64(A)
    d = A.shape[-1]
    return DriftModel(M_LINEAR_SDE, d, 0, A.reshape(A.shape[:-2] + (d * d,))), Dispersion(B)


# --------------------------------------------------------------------------- chirp models (models.py)
def _stack(*cols):
    cols = np.broadcast_arrays(*[np.asarray(c, dtype=np.float64) for c in cols])
    return np.stack(cols, axis=-1)


def _diag_rows(cols):
    """Batched np.diag from a list of (possibly batched) diagonal entries."""
    v = _stack(*cols)
    out = np.zeros(v.shape + (v.shape[-1],))
    idx = np.arange(v.shape[-1])
    out[..., idx, idx] = v
    return out


def model_harmonic_chirp(lam, b, ell, sigma, delta, num_harmonics=1, freq_scale=1.):
    """models.py:122-178 -> (drift, dispersion, m0, P0, H)."""
    n = int(num_harmonics)
    d = 2 * n + 2
    lam, b, ell, sigma, delta = (np.asarray(x, dtype=np.float64) for x in (lam, b, ell, sigma, delta))
    drift = DriftModel(M_HARMONIC_SDE, d, n, _stack(lam, ell, freq_scale))
    dispersion = Dispersion(_diag_rows([b, b] * n + [0., 2 * sigma * (math.sqrt(3) / ell) ** 1.5]))
    m0 = np.array([0., 1.] * n + [0., 0.])
    P0 = _diag_rows([delta, delta] * n + [sigma ** 2, (math.sqrt(3) / ell) ** 2 * sigma ** 2])
    H = np.array([0., 1.] * n + [0., 0.])
    return drift, dispersion, m0, P0, H


def model_chirp(lam, b, ell, sigma, delta):
    """models.py:76-119."""
    return model_harmonic_chirp(lam, b, ell, sigma, delta, 1, 1.)


def model_lascala(ell, sigma, delta):
    """models.py:181-261 (no damping, no chirp dispersion)."""
    return model_harmonic_chirp(0., 0., ell, sigma, delta, 1, 1.)


def disc_harmonic_chirp_lcd(lam, b, ell, sigma, num_harmonics=1, freq_scale=1.):
    """models.py:332-386."""
    n = int(num_harmonics)
    return DiscreteModel(M_HARMONIC_LCD, 2 * n + 2, n, _stack(lam, b, ell, sigma, freq_scale))


def disc_chirp_lcd(lam, b, ell, sigma):
    """models.py:264-311."""
    return disc_harmonic_chirp_lcd(lam, b, ell, sigma, 1, 1.)


def disc_chirp_lcd_cond_v(lam, b):
    """Chirp block conditioned on a given frequency-state value v (models.py:313-330): host-side NumPy closure
    ``m_and_cov(u (2,), v, dt) -> (rho Rot(dt 2 pi g(v)) u, q I)``.  Only the covariance-function utilities of the
    reference use it (cov_funcs.py:202); it is not a filter model and never reaches the kernels."""
    def m_and_cov(u, v, dt):
        th = dt * 2 * math.pi * g(v)
        rho = math.exp(-lam * dt)
        c, s_ = math.cos(th) * rho, math.sin(th) * rho
        q = b ** 2 * dt if lam == 0 else b ** 2 / (2 * lam) * (1 - math.exp(-2 * lam * dt))
        u = np.asarray(u, dtype=np.float64)
        return np.array([c * u[0] - s_ * u[1], s_ * u[0] + c * u[1]]), np.eye(2) * q
    return m_and_cov


def disc_chirp_euler_maruyama():
    """As in the reference (models.py:389-392): not provided, the chirp SDE is too stiff for Euler-Maruyama."""
    return NotImplemented


def disc_chirp_tme(*args, **kwargs):
    """The reference's Taylor-moment-expansion discretisation (models.py:395-416) needs the third-party ``tme`` package,
    which is neither vendored in the reference nor available here: out of scope."""
    raise NotImplementedError('disc_chirp_tme needs the un-vendored `tme` package of the reference; use disc_chirp_lcd')


def disc_model_lascala_lcd(ell, sigma):
    """models.py:419-434."""
    return DiscreteModel(M_LASCALA_LCD, 4, 1, _stack(ell, sigma))


def disc_m32(ell, sigma):
    """models.py:408-416, as a linear descriptor factory: call the result with dt via ``.at(dt)``."""
    class _M32:
        def at(self, dt):
            return linear_cond_m_cov(*_m32(ell, sigma, dt))

        def __call__(self, u, dt):
            F, S = _m32(ell, sigma, dt)
            return F @ np.asarray(u), S
    return _M32()


def _split(params, n):
    p = np.asarray(params, dtype=np.float64)
    if p.shape[-1] != n:
        raise ValueError(f'expected {n} parameters, got shape {p.shape}')
    return [p[..., i] for i in range(n)]


def build_harmonic_chirp_model(params, num_harmonics=1, freq_scale=1.):
    """models.py:462-494.  params = lam, b, delta, ell, sigma, m0_1 (optionally with a leading batch axis)."""
    lam, b, delta, ell, sigma, m0_v = _split(params, 6)
    drift, dispersion, _, P0, H = model_harmonic_chirp(lam, b, ell, sigma, delta, num_harmonics, freq_scale)
    m0 = _stack(*([0., 1.] * int(num_harmonics) + [m0_v, 0.]))
    m_and_cov = disc_harmonic_chirp_lcd(lam, b, ell, sigma, num_harmonics, freq_scale)
    return drift, dispersion, m_and_cov, m0, P0, H


def build_chirp_model(params):
    """models.py:437-459.  Note m0 = [0, 0, m0_v, 0] here (not the harmonic builder's [0, 1, ...])."""
    lam, b, delta, ell, sigma, m0_v = _split(params, 6)
    drift, dispersion, _, P0, H = model_chirp(lam, b, ell, sigma, delta)
    m0 = _stack(0., 0., m0_v, 0.)
    return drift, dispersion, disc_chirp_lcd(lam, b, ell, sigma), m0, P0, H


def build_lascala_model(params):
    """models.py:497-519.  params = delta, ell, sigma, m0_1."""
    delta, ell, sigma, m0_v = _split(params, 4)
    drift, dispersion, _, P0, H = model_lascala(ell, sigma, delta)
    # model_lascala's dispersion has zeros on the chirp rows (models.py:254-255)
    return drift, dispersion, disc_model_lascala_lcd(ell, sigma), _stack(0., 0., m0_v, 0.), P0, H


def build_kpt_chirp_model(params, fs, num_harmonics=1):
    """models.py:522-580 -> F, Sigma, m0, P0, h.  params = q1, q2, p0, f0, a0 (optionally with a leading batch axis: Sigma,
    m0 and P0 then carry it, F is shared)."""
    q1, q2, p0, f0, a0 = _split(params, 5)
    n = int(num_harmonics)
    dim_x = n + 2
    P0 = _diag_rows([p0] * dim_x)
    m0 = _stack(2 * math.pi * f0 / fs, *([a0] * n), 0.)
    F = np.eye(dim_x)
    F[-1, 0] = 1.
    Sigma = _diag_rows([(2 * math.pi * q1 / fs) ** 2] + [q2] * n + [0.])
    return F, Sigma, m0, P0, MeasurementKPT(n)
