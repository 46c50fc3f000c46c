"""ctypes front-end of the plain-C restatement oracle/c/port.c (TEST INFRASTRUCTURE; see oracle/__init__.py).

Takes duck-typed model descriptions (any object with ``model_id, d, n_harm, params, gamma``) so that tests can
hand it the very same spec objects they hand to the HIP library, without this package importing the product.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'c', 'libcgp_port.so')

# Enumerations of include/chirpgp_hip.h
F_EKF, F_SGP, F_CD_EKF, F_CD_SGP, F_EKF_KPT = range(5)
S_EKS, S_SGP, S_CD_EKS, S_CD_SGP = range(4)
M_LINEAR, M_HARMONIC_LCD, M_LASCALA_LCD, M_LINEAR_SDE, M_HARMONIC_SDE, M_KPT = range(6)
NLL_FINAL_ONLY = 0x1

_dp = C.POINTER(C.c_double)


class _Model(C.Structure):
    _fields_ = [('model_id', C.c_int32), ('d', C.c_int32), ('n_harm', C.c_int32), ('n_params', C.c_int32),
                ('params', _dp), ('param_stride', C.c_int64), ('gamma', _dp), ('gamma_stride', C.c_int64)]


class _Sigma(C.Structure):
    _fields_ = [('s', C.c_int32), ('d', C.c_int32), ('xi', _dp), ('w', _dp), ('group_start', C.c_void_p), ('n_groups', C.c_int32)]


class _Init(C.Structure):
    _fields_ = [('H', _dp), ('H_stride', C.c_int64), ('Xi', _dp), ('Xi_stride', C.c_int64),
                ('m0', _dp), ('m0_stride', C.c_int64), ('P0', _dp), ('P0_stride', C.c_int64)]


def build():
    """Compile oracle/c/port.c if the shared object is missing or stale."""
    src = os.path.join(_HERE, 'c', 'port.c')
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'c/libcgp_port.so'], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _declare(_lib):
    _lib.port_filter.restype = C.c_int
    _lib.port_filter.argtypes = [C.c_int, C.POINTER(_Model), C.POINTER(_Sigma), C.POINTER(_Init), C.c_double,
                                 _dp, C.c_int64, C.c_int64, _dp, _dp, _dp, C.c_uint32]
    _lib.port_smoother.restype = C.c_int
    _lib.port_smoother.argtypes = [C.c_int, C.POINTER(_Model), C.POINTER(_Sigma), C.c_double,
                                   _dp, _dp, C.c_int64, C.c_int64, _dp, _dp, C.c_uint32]
    _lib.port_gaussian_expectation.restype = C.c_int
    _lib.port_gaussian_expectation.argtypes = [_dp, _dp, C.c_int64, C.c_int64, _dp, _dp, C.c_int32, _dp]
    _lib.port_num_threads.restype = C.c_int
    _lib.port_set_num_threads.restype = None
    _lib.port_set_num_threads.argtypes = [C.c_int]
    _lib.port_fixed_d.restype = C.c_int
    return _lib


def lib():
    global _lib
    if _lib is None:
        _lib = _declare(C.CDLL(build()))
    return _lib


NATIVE_FLAGS = ['-O3', '-march=native', '-fopenmp', '-fPIC', '-std=gnu99', '-fno-fast-math', '-ffp-contract=fast']
_native = {}


def _host_tag():
    """Short hash of this machine's CPU model and ISA flags: a -march=native object must never run on another host
    (gpurun ships built .so files to the GPU box, whose CPU differs from the build container's)."""
    import hashlib
    try:
        txt = open('/proc/cpuinfo').read()
        keep = sorted({ln for ln in txt.splitlines() if ln.startswith(('model name', 'flags'))})
    except OSError:
        keep = []
    return hashlib.sha1('\n'.join(keep).encode()).hexdigest()[:10]


def build_native(d):
    """The TIMED build of the same source for bench.py's cpu_baseline: state dimension fixed at compile time, -march=native,
    contraction allowed (`gcc -O3 -march=native -DFIXED_D=d -ffp-contract=fast`), compiled on the machine that runs it.
    The checker build (build()) is unchanged.  -> path of the shared object."""
    so = os.path.join(_HERE, 'c', f'libcgp_port_native_d{int(d)}_{_host_tag()}.so')
    src = os.path.join(_HERE, 'c', 'port.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([os.environ.get('CC', 'gcc'), *NATIVE_FLAGS, f'-DFIXED_D={int(d)}', '-shared', '-o', so, src, '-lm'])
    return so


def native(d):
    """ctypes handle of the timed build for dimension d (pass it as ``use=`` to filter / smoother)."""
    if d not in _native:
        _native[d] = _declare(C.CDLL(build_native(d)))
        assert _native[d].port_fixed_d() == int(d)
    return _native[d]


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _model_struct(model, keep):
    params = _arr(model.params)
    stride = params.shape[-1] if params.ndim == 2 else 0
    gamma = getattr(model, 'gamma', None)
    g = _arr(gamma) if gamma is not None else None
    gstride = g.shape[-1] * g.shape[-2] if (g is not None and g.ndim == 3) else 0
    keep += [params, g]
    return _Model(int(model.model_id), int(model.d), int(model.n_harm), int(params.shape[-1]),
                  _p(params), stride, _p(g), gstride)


def _sigma_struct(sigma, keep):
    if sigma is None:
        return None
    xi, w = _arr(sigma.xi), _arr(sigma.w)
    keep += [xi, w]
    return _Sigma(int(xi.shape[0]), int(xi.shape[1]), _p(xi), _p(w), None, 0)


def _strided(x, base_ndim, keep):
    a = _arr(x)
    keep.append(a)
    per = int(np.prod(a.shape[a.ndim - base_ndim:])) if base_ndim else 1
    return _p(a), (per if a.ndim > base_ndim else 0)


def filter(method, model, sigma, H, Xi, m0, P0, dt, ys, nll_final_only=False, use=None, out=None):
    """-> (mfs (B,T,d), Pfs (B,T,d,d), nll (B,T) or (B,)); a 1-D ys gives un-batched outputs.
    ``out``: preallocated (mfs, Pfs, nll) to write into (the timed baseline reuses touched buffers instead of paying
    first-touch page faults for GBs of fresh memory in every call)."""
    ys = _arr(ys)
    single = ys.ndim == 1
    ys2 = ys[None, :] if single else ys
    B, T = ys2.shape
    d = int(model.d)
    keep = []
    ms = _model_struct(model, keep)
    sg = _sigma_struct(sigma, keep)
    init = _Init()
    if H is not None:
        init.H, init.H_stride = _strided(H, 1, keep)
    xi_arr = np.atleast_1d(_arr(Xi))
    keep.append(xi_arr)
    init.Xi, init.Xi_stride = _p(xi_arr), (1 if xi_arr.size > 1 else 0)
    init.m0, init.m0_stride = _strided(m0, 1, keep)
    init.P0, init.P0_stride = _strided(P0, 2, keep)
    if out is not None:
        mfs, Pfs, nll = out
        assert mfs.shape == (B, T, d) and Pfs.shape == (B, T, d, d) and nll.shape == ((B,) if nll_final_only else (B, T))
        assert all(a.flags.c_contiguous and a.dtype == np.float64 for a in out)
    else:
        mfs, Pfs = np.empty((B, T, d)), np.empty((B, T, d, d))
        nll = np.empty((B,) if nll_final_only else (B, T))
    rc = (use or lib()).port_filter(method, C.byref(ms), C.byref(sg) if sg is not None else None, C.byref(init), float(dt),
                           _p(ys2), B, T, _p(mfs), _p(Pfs), _p(nll), NLL_FINAL_ONLY if nll_final_only else 0)
    if rc != 0:
        raise RuntimeError(f'port_filter failed: {rc}')
    return (mfs[0], Pfs[0], nll[0]) if single else (mfs, Pfs, nll)


def smoother(method, model, sigma, dt, mfs, Pfs, use=None, out=None):
    """-> (mss, Pss) with the shapes of (mfs, Pfs)."""
    mfs, Pfs = _arr(mfs), _arr(Pfs)
    single = mfs.ndim == 2
    m3 = mfs[None] if single else mfs
    P4 = Pfs[None] if single else Pfs
    B, T, d = m3.shape
    keep = []
    ms = _model_struct(model, keep)
    sg = _sigma_struct(sigma, keep)
    if out is not None:
        mss, Pss = out
        assert mss.shape == m3.shape and Pss.shape == P4.shape and mss.flags.c_contiguous and Pss.flags.c_contiguous
    else:
        mss, Pss = np.empty_like(m3), np.empty_like(P4)
    rc = (use or lib()).port_smoother(method, C.byref(ms), C.byref(sg) if sg is not None else None, float(dt),
                             _p(m3), _p(P4), B, T, _p(mss), _p(Pss), 0)
    if rc != 0:
        raise RuntimeError(f'port_smoother failed: {rc}')
    return (mss[0], Pss[0]) if single else (mss, Pss)


def gaussian_expectation(ms, sd, xi, w):
    ms, sd, xi, w = _arr(ms).ravel(), _arr(sd).ravel(), _arr(xi).ravel(), _arr(w).ravel()
    out = np.empty_like(ms)
    lib().port_gaussian_expectation(_p(ms), _p(sd), ms.size, 1, _p(xi), _p(w), xi.size, _p(out))
    return out


def num_threads(use=None):
    return int((use or lib()).port_num_threads())


def set_num_threads(n, use=None):
    (use or lib()).port_set_num_threads(int(n))
