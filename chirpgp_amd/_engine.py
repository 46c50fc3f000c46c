"""ctypes binding of libchirpgp_hip.so (include/chirpgp_hip.h) and the marshalling shared by the public functions.

PyTorch-ROCm is used for device memory, streams and torch.distributed only: every array handed to the library is a
``torch.Tensor.data_ptr()`` of a float64 CUDA(HIP) tensor; no torch op takes part in the arithmetic.  There is no
CPU path: a missing library or GPU raises.
"""
import ctypes as C
import os
import threading
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libchirpgp_hip.so')

# enumerations of include/chirpgp_hip.h
F_EKF, F_SGP, F_CD_EKF, F_CD_SGP, F_EKF_KPT = range(5)
S_EKS, S_SGP, S_CD_EKS, S_CD_SGP = range(4)
M_LINEAR, M_HARMONIC_LCD, M_LASCALA_LCD, M_LINEAR_SDE, M_HARMONIC_SDE, M_KPT = range(6)
NLL_FINAL_ONLY, WAVE_PER_TRIAL, THREAD_PER_TRIAL, SEQUENTIAL_SCAN, GENERIC_KERNEL, SIM_FIXED_X0 = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
LITERAL_SIGMA_SUM, DPP_KERNEL, FOUR_TRIALS_PER_WAVE, ONE_TRIAL_PER_WAVE = 0x40, 0x80, 0x200, 0x400
TIME_SPLIT, NO_TIME_SPLIT = 0x800, 0x1000
SIGMA_STANDARD = 0x1
SIGMA_AXIAL = 0x2
MAX_D = 12            # include/chirpgp_hip.h: CGP_MAX_D (9 .. 12: the harmonic LCD model with 4 or 5 harmonics)


def _check_dimension(spec):
    d = int(spec.d)
    limit = MAX_D if int(spec.model_id) == M_HARMONIC_LCD else 8
    if d > limit:
        raise NotImplementedError(f'state dimension {d} > {limit} is not compiled into libchirpgp_hip.so for this model')
    return d

_vp = C.c_void_p


class CgpModel(C.Structure):
    _fields_ = [('model_id', C.c_int32), ('d', C.c_int32), ('n_harm', C.c_int32), ('n_params', C.c_int32),
                ('params', _vp), ('param_stride', C.c_int64), ('gamma', _vp), ('gamma_stride', C.c_int64)]


class CgpSigma(C.Structure):
    _fields_ = [('s', C.c_int32), ('d', C.c_int32), ('xi', _vp), ('w', _vp), ('group_start', _vp), ('n_groups', C.c_int32),
                ('flags', C.c_uint32)]


class CgpInit(C.Structure):
    _fields_ = [('H', _vp), ('H_stride', C.c_int64), ('Xi', _vp), ('Xi_stride', C.c_int64),
                ('m0', _vp), ('m0_stride', C.c_int64), ('P0', _vp), ('P0_stride', C.c_int64)]


class CgpSmoothOut(C.Structure):
    _fields_ = [('mss', _vp), ('Pss', _vp), ('comp', C.c_int32), ('func', C.c_int32), ('comp_mean', _vp), ('comp_var', _vp),
                ('expect', _vp), ('xi', _vp), ('w', _vp), ('order', C.c_int32)]


EXPORTS = ('cgp_version', 'cgp_create', 'cgp_destroy', 'cgp_last_error', 'cgp_filter', 'cgp_smoother',
           'cgp_gaussian_expectation', 'cgp_debug_math', 'cgp_simulate', 'cgp_add_noise', 'cgp_debug_philox',
           'cgp_debug_set', 'cgp_debug_counters', 'cgp_gaussian_expectation_fn', 'cgp_filter_time_split', 'cgp_squared_error_sums',
           'cgp_reserve_workspace', 'cgp_release_workspace', 'cgp_source_hash', 'cgp_smoother_select', 'cgp_ekf_nll_grad', 'cgp_model_from_source', 'cgp_custom_model_destroy',
           'cgp_filter_custom', 'cgp_smoother_custom', 'cgp_smoother_time_split')

_lib = None
_lock = threading.Lock()


def load_library():
    """dlopen libchirpgp_hip.so and declare the prototypes.  Raises if the library has not been built."""
    global _lib
    # torch first: PyTorch-ROCm ships its own libamdhip64; loading it before our library makes both use the same HIP
    # runtime (the reverse order leaves two runtimes in the process and hipGetDeviceCount fails in ours).
    _torch()
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                               f'or `make -C chirpgp_amd/csrc` (there is no CPU fallback)')
        lib = C.CDLL(LIB_PATH)
        lib.cgp_version.restype = C.c_int
        lib.cgp_create.restype = C.c_int
        lib.cgp_create.argtypes = [C.POINTER(_vp), C.c_int]
        lib.cgp_destroy.restype = None
        lib.cgp_destroy.argtypes = [_vp]
        lib.cgp_last_error.restype = C.c_char_p
        lib.cgp_last_error.argtypes = [_vp]
        lib.cgp_filter.restype = C.c_int
        lib.cgp_filter.argtypes = [_vp, C.c_int, C.POINTER(CgpModel), C.POINTER(CgpSigma), C.POINTER(CgpInit), C.c_double,
                                   _vp, C.c_int64, C.c_int64, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_uint32, _vp]
        lib.cgp_filter_time_split.restype = C.c_int
        lib.cgp_filter_time_split.argtypes = [_vp, C.c_int, C.POINTER(CgpModel), C.POINTER(CgpSigma), C.POINTER(CgpInit), C.c_double,
                                              _vp, C.c_int64, C.c_int64, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_uint32,
                                              C.c_int64, C.c_int64, _vp, _vp]
        lib.cgp_smoother.restype = C.c_int
        lib.cgp_smoother.argtypes = [_vp, C.c_int, C.POINTER(CgpModel), C.POINTER(CgpSigma), C.c_double,
                                     _vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_uint32, _vp]
        lib.cgp_smoother_time_split.restype = C.c_int
        lib.cgp_smoother_time_split.argtypes = [_vp, C.c_int, C.POINTER(CgpModel), C.POINTER(CgpSigma), C.c_double, _vp, _vp, C.c_int64, C.c_int64,
                                                _vp, _vp, C.c_uint32, C.c_int64, C.c_int64, _vp, _vp]
        lib.cgp_smoother_select.restype = C.c_int
        lib.cgp_smoother_select.argtypes = [_vp, C.c_int, C.POINTER(CgpModel), C.POINTER(CgpSigma), C.c_double,
                                            _vp, _vp, C.c_int64, C.c_int64, C.POINTER(CgpSmoothOut), C.c_uint32, _vp]
        lib.cgp_ekf_nll_grad.restype = C.c_int
        lib.cgp_ekf_nll_grad.argtypes = [_vp, C.POINTER(CgpModel), C.POINTER(CgpInit), C.c_double, _vp, C.c_int64, C.c_int64, _vp,
                                         C.c_int64, C.c_int64, _vp, C.c_int32, _vp, _vp, C.c_uint32, _vp]
        lib.cgp_model_from_source.restype = C.c_int
        lib.cgp_model_from_source.argtypes = [_vp, C.c_int, C.c_int32, C.c_char_p, C.c_char_p, C.POINTER(_vp)]
        lib.cgp_custom_model_destroy.restype = None
        lib.cgp_custom_model_destroy.argtypes = [_vp]
        lib.cgp_filter_custom.restype = C.c_int
        lib.cgp_filter_custom.argtypes = [_vp, _vp, C.POINTER(CgpSigma), _vp, C.c_int64, _vp, C.c_int64, C.POINTER(CgpInit), C.c_double, _vp, C.c_int64, C.c_int64, _vp,
                                          C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_uint32, _vp]
        lib.cgp_smoother_custom.restype = C.c_int
        lib.cgp_smoother_custom.argtypes = [_vp, _vp, C.POINTER(CgpSigma), _vp, C.c_int64, _vp, C.c_int64, C.c_double, _vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_uint32, _vp]
        lib.cgp_gaussian_expectation.restype = C.c_int
        lib.cgp_gaussian_expectation.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int32, _vp, _vp]
        lib.cgp_gaussian_expectation_fn.restype = C.c_int
        lib.cgp_gaussian_expectation_fn.argtypes = [_vp, C.c_int, _vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int32, _vp, _vp]
        lib.cgp_squared_error_sums.restype = C.c_int
        lib.cgp_squared_error_sums.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _vp, _vp]
        lib.cgp_debug_math.restype = C.c_int
        lib.cgp_debug_math.argtypes = [_vp, C.c_int, _vp, C.c_int64, _vp, _vp, _vp]
        lib.cgp_simulate.restype = C.c_int
        lib.cgp_simulate.argtypes = [_vp, C.POINTER(CgpModel), C.POINTER(CgpInit), C.c_double, C.c_uint64, C.c_int64,
                                     C.c_int64, C.c_int64, _vp, _vp, C.c_uint32, _vp]
        lib.cgp_add_noise.restype = C.c_int
        lib.cgp_add_noise.argtypes = [_vp, _vp, C.c_int64, _vp, C.c_int64, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]
        lib.cgp_debug_philox.restype = C.c_int
        lib.cgp_debug_set.restype = C.c_int
        lib.cgp_debug_set.argtypes = [_vp, C.c_int, C.c_int64]
        lib.cgp_debug_counters.restype = C.c_int
        lib.cgp_debug_counters.argtypes = [_vp, C.POINTER(C.c_uint64), C.c_int, _vp]
        lib.cgp_debug_philox.argtypes = [_vp, _vp, _vp, C.c_int64, _vp, _vp]
        lib.cgp_reserve_workspace.restype = C.c_int
        lib.cgp_reserve_workspace.argtypes = [_vp, C.c_size_t, _vp]
        lib.cgp_release_workspace.restype = C.c_int
        lib.cgp_release_workspace.argtypes = [_vp, _vp]
        lib.cgp_source_hash.restype = C.c_char_p
        lib.cgp_source_hash.argtypes = []
        built_from, tree = lib.cgp_source_hash().decode(), source_hash()
        if tree is not None and built_from != tree:
            raise RuntimeError(f'{LIB_PATH} is stale: built from sources {built_from[:16]}..., the checked-out csrc/ + include/ hash to '
                               f'{tree[:16]}... -- rebuild it (`make -C chirpgp_amd/csrc`); there is no CPU fallback')
        _lib = lib
        return lib


def source_hash():
    """sha256 of the sources a library of this tree is built from -- csrc/*.hip, csrc/*.hpp (byte order of the names), csrc/Makefile,
    include/chirpgp_hip.h, concatenated: what csrc/Makefile embeds as cgp_source_hash().  None where the sources are not shipped
    (a binary-only install), in which case nothing can be compared."""
    import glob
    import hashlib
    src = os.path.join(_HERE, 'csrc')
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(src, '*.hip')) + glob.glob(os.path.join(src, '*.hpp')))
    files = [os.path.join(src, n) for n in names] + [os.path.join(src, 'Makefile'), os.path.join(os.path.dirname(_HERE), 'include', 'chirpgp_hip.h')]
    if not names or not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


_contexts = {}


def _torch():
    import torch
    return torch


def context(device_index=None):
    """One cgp_ctx per (process, device)."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError('chirpgp_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback')
    if device_index is None:
        device_index = torch.cuda.current_device()
    with _lock:
        ctx = _contexts.get(device_index)
    if ctx is None:
        lib = load_library()
        h = _vp()
        rc = lib.cgp_create(C.byref(h), int(device_index))
        if rc != 0:
            raise RuntimeError(f'cgp_create(device={device_index}) failed with {rc}')
        with _lock:
            ctx = _contexts.setdefault(device_index, h)
    return ctx


def _check(ctx, rc, what):
    if rc != 0:
        msg = load_library().cgp_last_error(ctx)
        raise RuntimeError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')


def _is_torch(x):
    return type(x).__module__.startswith('torch')


def dev(x, device=None):
    """-> contiguous float64 tensor on the GPU (uploading NumPy input)."""
    torch = _torch()
    if _is_torch(x):
        t = x
        if not t.is_cuda:
            t = t.cuda(device) if device is not None else t.cuda()
        return t.to(torch.float64).contiguous()
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    t = torch.from_numpy(a)
    return t.cuda(device) if device is not None else t.cuda()


_const_cache = {}


# Optional timing hook used by bench.py: when set to a list, every kernel launch appends
# (name, start_event, end_event), the events recorded on the launch stream immediately around the C-ABI call.
kernel_events = None


def _timed(name, call):
    if kernel_events is None:
        return call()
    torch = _torch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = call()
    e1.record()
    kernel_events.append((name, e0, e1))
    return rc


def dev_const(x):
    """Device copy of a small host constant (model parameters, H, m0, P0, sigma points), cached by value so that
    repeated calls with the same model do not pay a pageable host-to-device copy (which would also serialise the
    host behind the previous kernel)."""
    if _is_torch(x):
        return dev(x)
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if a.size > 8192:
        return dev(a)
    key = (_torch().cuda.current_device(), a.shape, a.tobytes())
    t = _const_cache.get(key)
    if t is None:
        if len(_const_cache) >= 512:
            _const_cache.clear()
        t = _const_cache[key] = dev(a)
    return t


def _index_const(a):
    """Device copy of a small int32 index array, cached by value like dev_const (a lock-step line search passes the same
    few index sets again and again; an uncached pageable upload would serialise the host behind the previous kernel)."""
    torch = _torch()
    if a.size > 65536:
        return torch.from_numpy(a).cuda()
    key = (torch.cuda.current_device(), 'i32', a.shape, a.tobytes())
    t = _const_cache.get(key)
    if t is None:
        if len(_const_cache) >= 512:
            _const_cache.clear()
        t = _const_cache[key] = torch.from_numpy(a).cuda()
    return t


def _ptr(t):
    return _vp(t.data_ptr()) if t is not None else None


def _stream():
    return _vp(_torch().cuda.current_stream().cuda_stream)


def _batched_operand(x, base_ndim, B, name):
    """Device tensor of an operand that is shared (base_ndim dims) or per-trial (leading B); returns (tensor, stride)."""
    t = dev_const(x)
    if t.ndim == base_ndim:
        return t, 0
    if t.ndim == base_ndim + 1 and t.shape[0] == B:
        per = 1
        for s in t.shape[1:]:
            per *= int(s)
        return t, per
    raise ValueError(f'{name}: expected {base_ndim} dims (shared) or a leading batch axis of {B}, got shape {tuple(t.shape)}')


def _model_struct(spec, gamma, B, keep):
    params = dev_const(spec.params)
    if params.ndim == 2 and params.shape[0] != B:
        raise ValueError(f'model parameters have batch {params.shape[0]} but the data has batch {B}')
    keep.append(params)
    m = CgpModel()
    m.model_id, m.d, m.n_harm, m.n_params = int(spec.model_id), int(spec.d), int(spec.n_harm), int(params.shape[-1])
    m.params, m.param_stride = _ptr(params), (int(params.shape[-1]) if params.ndim == 2 else 0)
    if gamma is not None:
        g, gs = _batched_operand(gamma, 2, B, 'dispersion')
        if tuple(g.shape[-2:]) != (spec.d, spec.d):
            raise ValueError(f'b b^T must be {spec.d} x {spec.d}, got {tuple(g.shape)}')
        keep.append(g)
        m.gamma, m.gamma_stride = _ptr(g), gs
    return m


_sigma_cache = {}


def _sigma_struct(sgps, d, keep, nonlinear_coord=None):
    """cgp_sigma for a SigmaPoints.  With `nonlinear_coord` = v (the chirp models: v = d - 2) the points are reordered so
    that points sharing xi[0..v] are contiguous and the group boundaries are passed along: the kernels then evaluate the
    transcendental part of the model once per group (include/chirpgp_hip.h, cgp_sigma).  Reordering only changes the
    order of summation."""
    if sgps is None:
        return None
    torch = _torch()
    xi_h, w_h = np.asarray(sgps.xi, dtype=np.float64), np.asarray(sgps.w, dtype=np.float64)
    if xi_h.ndim != 2 or xi_h.shape[1] != d or w_h.shape != (xi_h.shape[0],):
        raise ValueError(f'sigma points must be (s, {d}) with (s,) weights; got {xi_h.shape}, {w_h.shape}')
    key = (torch.cuda.current_device(), nonlinear_coord, xi_h.shape, xi_h.tobytes(), w_h.tobytes())
    hit = _sigma_cache.get(key)
    if hit is None:
        gs, sflags = None, 0
        if nonlinear_coord is not None:
            v = int(nonlinear_coord)
            prefix = xi_h[:, :v + 1] + 0.0          # exact grouping (-0.0 folded into +0.0)
            _, inverse = np.unique(prefix, axis=0, return_inverse=True)
            order = np.argsort(inverse.ravel(), kind='stable')
            xi_h, w_h, inv = xi_h[order], w_h[order], inverse.ravel()[order]
            starts = np.flatnonzero(np.r_[True, inv[1:] != inv[:-1], True]).astype(np.int32)
            gs = torch.from_numpy(starts).cuda()
            sflags = SIGMA_STANDARD if _is_standard(xi_h, w_h, starts, v) else 0
            if sflags and (np.count_nonzero(xi_h[:, :d - 1], axis=1) <= 1).all():
                sflags |= SIGMA_AXIAL            # cubature: every point on one axis (include/chirpgp_hip.h)
        if len(_sigma_cache) >= 64:
            _sigma_cache.clear()
        hit = _sigma_cache[key] = (dev(xi_h), dev(w_h), gs, sflags)
    xi, w, gs, sflags = hit
    keep += [xi, w, gs]
    return CgpSigma(int(xi.shape[0]), int(d), _ptr(xi), _ptr(w), _ptr(gs), int(gs.numel() - 1) if gs is not None else 0, sflags)


def _is_standard(xi, w, starts, v, tol=1e-13):
    """The assertion behind CGP_SIGMA_STANDARD (include/chirpgp_hip.h): unit weight, zero mean, identity second moment,
    and groups whose members differ in the last coordinate only, with zero weighted mean there."""
    d = xi.shape[1]
    if v != d - 2:
        return False
    if abs(w.sum() - 1) > tol or np.abs(w @ xi).max() > tol or np.abs((xi.T * w) @ xi - np.eye(d)).max() > tol:
        return False
    for a, b in zip(starts[:-1], starts[1:]):
        if np.abs(xi[a:b, :d - 1] - xi[a, :d - 1]).max() != 0 or abs(w[a:b] @ xi[a:b, d - 1]) > tol * 0.1:
            return False
    return True


def _nonlinear_coord(spec):
    """State coordinate the model is nonlinear in (chirp / harmonic / La Scala models: d - 2), None for linear models."""
    return int(spec.d) - 2 if int(spec.model_id) in (M_HARMONIC_LCD, M_LASCALA_LCD, M_HARMONIC_SDE) else None


def _out(t, like_numpy, squeeze):
    if squeeze:
        t = t[0]
    return t.cpu().numpy() if like_numpy else t


class _PerThread(threading.local):
    junction_error = None


_per_thread = _PerThread()


def __getattr__(name):
    # `_engine.last_junction_error`: device tensor [B] of the CALLING THREAD's most recent time-split filter call
    # (cgp_filter_time_split), None when that call fell back to the sequential filter (split_tol) -- the rows returned are then
    # not the split run's.  Prefer `return_junction_error=True`, which hands the tensor back with the results.
    if name == 'last_junction_error':
        return _per_thread.junction_error
    raise AttributeError(name)


def run_filter(method, spec, sgps, gamma, H, Xi, m0, P0, dt, ys, nll_final_only=False, flags=0, want=(True, True, True),
               trials_per_record=None, record_index=None, time_split=None, split_tol=None, return_junction_error=False):
    """cgp_filter with NumPy / torch marshalling.  ys (T,) or (B, T) -> (mfs, Pfs, nll) with matching leading axes.

    Shared records (include/chirpgp_hip.h, cgp_filter): with ``trials_per_record = k`` every record of ys -- (T,) or (R, T) --
    serves k consecutive trials (k parameter vectors of a sweep, the 2 P + 1 probes of a difference gradient), B = R k, and
    the results always carry the batch axis; ``record_index`` (n,) picks and orders the records first (B = n k).  The
    record is read from ONE copy in HBM: nothing is replicated.

    ``time_split = (segments, burn_in)``: the time-split filter with burn-in (include/chirpgp_hip.h, cgp_filter_time_split) -- several
    wavefronts per trial for batches that leave most SIMDs idle.  With ``return_junction_error=True`` the per-trial junction mismatch
    of THIS call comes back as a fourth output (device tensor [B]; None if the results are the sequential filter's); it is also
    left, per calling thread, in ``_engine.last_junction_error``.  With ``split_tol`` the call checks it (one device
    synchronisation) and falls back to the sequential filter if any junction is further off."""
    junction = None
    torch = _torch()
    like_numpy = not _is_torch(ys)
    ys_d = dev(ys)
    shared = trials_per_record is not None or record_index is not None
    squeeze = ys_d.ndim == 1 and not shared
    if ys_d.ndim == 1:
        ys_d = ys_d[None, :]
    if ys_d.ndim != 2:
        raise ValueError(f'ys must be (T,) or (B, T), got {tuple(ys_d.shape)}')
    R, T = int(ys_d.shape[0]), int(ys_d.shape[1])
    rep = 1 if trials_per_record is None else int(trials_per_record)
    if rep < 1:
        raise ValueError('trials_per_record must be >= 1')
    idx_h = None
    if record_index is not None:
        idx_h = np.ascontiguousarray(np.asarray(record_index, dtype=np.int64).reshape(-1))
        if idx_h.size and (idx_h.min() < 0 or idx_h.max() >= R):
            raise ValueError(f'record_index outside 0..{R - 1}')
    B = (R if idx_h is None else int(idx_h.size)) * rep
    d = _check_dimension(spec)
    # constants, stream, outputs and the launch all belong to the device the data lives on, whatever the current one is
    with torch.cuda.device(ys_d.device):
        ctx = context(ys_d.device.index)
        keep = [ys_d]
        idx_d = None
        if idx_h is not None:
            idx_d = _index_const(idx_h.astype(np.int32))
            keep.append(idx_d)
        model = _model_struct(spec, gamma, B, keep)
        sig = _sigma_struct(sgps, d, keep, _nonlinear_coord(spec))
        init = _init_struct(H, Xi, m0, P0, d, B, keep)
        opts = dict(dtype=torch.float64, device=ys_d.device)
        mfs = torch.empty((B, T, d), **opts) if want[0] else None
        Pfs = torch.empty((B, T, d, d), **opts) if want[1] else None
        nll = (torch.empty((B,) if nll_final_only else (B, T), **opts)) if want[2] else None
        fl = int(flags) | (NLL_FINAL_ONLY if nll_final_only else 0)
        lib, st = load_library(), _stream()
        if time_split is not None:
            segments, burn_in = (int(v) for v in time_split)
            err = torch.empty((B,), **opts)
            rc = _timed('filter', lambda: lib.cgp_filter_time_split(ctx, int(method), C.byref(model), C.byref(sig) if sig is not None else None,
                                                                    C.byref(init), float(dt), _ptr(ys_d), T, rep, _ptr(idx_d), B, T,
                                                                    _ptr(mfs), _ptr(Pfs), _ptr(nll), fl, segments, burn_in, _ptr(err), st))
            _check(ctx, rc, 'cgp_filter_time_split')
            junction = err
            if split_tol is not None and not bool((err <= float(split_tol)).all()):      # NaN / inf / too far: the sequential filter
                time_split, junction = None, None
        if time_split is None:
            rc = _timed('filter', lambda: lib.cgp_filter(ctx, int(method), C.byref(model), C.byref(sig) if sig is not None else None,
                                                         C.byref(init), float(dt), _ptr(ys_d), T, rep, _ptr(idx_d), B, T,
                                                         _ptr(mfs), _ptr(Pfs), _ptr(nll), fl, st))
            _check(ctx, rc, 'cgp_filter')
        _per_thread.junction_error = junction
        res = tuple(None if t is None else _out(t, like_numpy, squeeze) for t in (mfs, Pfs, nll))
        return res + (junction,) if return_junction_error else res


# ---- models compiled at run time (include/chirpgp_hip.h: cgp_model_from_source) ----------------------------------------------------
_custom_cache = {}


def custom_model(spec, device_index=None):
    """The compiled form of a models.CustomDiscrete / CustomDrift on a device: built by hiprtc on first use (about a second), then cached by
    (device, kind, d, source).  A source that does not compile raises RuntimeError with the compiler's messages."""
    torch = _torch()
    if device_index is None:
        device_index = torch.cuda.current_device()
    key = (device_index, int(spec.kind), int(spec.d), spec.source)
    h = _custom_cache.get(key)
    if h is None:
        ctx = context(device_index)
        h = _vp()
        rc = load_library().cgp_model_from_source(ctx, int(spec.kind), int(spec.d), spec.source.encode(), os.path.join(_HERE, 'csrc').encode(), C.byref(h))
        _check(ctx, rc, 'cgp_model_from_source')
        _custom_cache[key] = h
    return h


def release_custom_models():
    """Unload every runtime-compiled model of this process (cgp_custom_model_destroy: the code objects stay loaded for the life of the process
    otherwise -- a sweep over many sources would accumulate them).  The streams that used them must have drained: synchronises first."""
    if not _custom_cache:
        return 0
    _torch().cuda.synchronize()
    lib = load_library()
    n = len(_custom_cache)
    for h in _custom_cache.values():
        lib.cgp_custom_model_destroy(h)
    _custom_cache.clear()
    return n


def _custom_params(spec, gamma, B, keep):
    params = dev_const(spec.params)
    if params.ndim == 2 and params.shape[0] != B:
        raise ValueError(f'model parameters have batch {params.shape[0]} but the data has batch {B}')
    keep.append(params)
    g = gs = None
    if gamma is not None:
        g, gs = _batched_operand(gamma, 2, B, 'dispersion')
        if tuple(g.shape[-2:]) != (spec.d, spec.d):
            raise ValueError(f'b b^T must be {spec.d} x {spec.d}, got {tuple(g.shape)}')
        keep.append(g)
    return params, (int(params.shape[-1]) if params.ndim == 2 else 0), g, (gs or 0)


def run_filter_custom(spec, gamma, H, Xi, m0, P0, dt, ys, nll_final_only=False, want=(True, True, True), flags=0, sgps=None):
    """ekf / cd_ekf -- with `sgps`: sgp_filter / cd_sgp_filter -- on a model compiled at run time (cgp_filter_custom): the generic
    one-lane-per-trial kernel instantiated on the caller's source.  ys (T,) or (B, T)."""
    torch = _torch()
    like_numpy = not _is_torch(ys)
    ys_d = dev(ys)
    squeeze = ys_d.ndim == 1
    if squeeze:
        ys_d = ys_d[None, :]
    B, T = int(ys_d.shape[0]), int(ys_d.shape[1])
    d = int(spec.d)
    with torch.cuda.device(ys_d.device):
        ctx = context(ys_d.device.index)
        handle = custom_model(spec, ys_d.device.index)
        keep = [ys_d]
        params, pstride, g, gstride = _custom_params(spec, gamma, B, keep)
        sig = _sigma_struct(sgps, d, keep, None)
        init = _init_struct(H, Xi, m0, P0, d, B, keep)
        opts = dict(dtype=torch.float64, device=ys_d.device)
        mfs = torch.empty((B, T, d), **opts) if want[0] else None
        Pfs = torch.empty((B, T, d, d), **opts) if want[1] else None
        nll = (torch.empty((B,) if nll_final_only else (B, T), **opts)) if want[2] else None
        fl = int(flags) | (NLL_FINAL_ONLY if nll_final_only else 0)
        rc = _timed('filter', lambda: load_library().cgp_filter_custom(ctx, handle, C.byref(sig) if sig is not None else None, _ptr(params), pstride, _ptr(g), gstride, C.byref(init), float(dt),
                                                                       _ptr(ys_d), T, 1, None, B, T, _ptr(mfs), _ptr(Pfs), _ptr(nll), fl, _stream()))
        _check(ctx, rc, 'cgp_filter_custom')
        return tuple(None if t is None else _out(t, like_numpy, squeeze) for t in (mfs, Pfs, nll))


def run_smoother_custom(spec, gamma, dt, mfs, Pfs, flags=0, sgps=None):
    """eks / cd_eks -- with `sgps`: sgp_smoother / cd_sgp_smoother -- on a model compiled at run time (cgp_smoother_custom)."""
    torch = _torch()
    like_numpy = not _is_torch(mfs)
    m, P = dev(mfs), dev(Pfs)
    squeeze = m.ndim == 2
    if squeeze:
        m, P = m[None], P[None]
    if m.ndim != 3 or P.ndim != 4 or P.shape[:2] != m.shape[:2] or P.shape[2] != m.shape[2] or P.shape[3] != m.shape[2]:
        raise ValueError(f'mfs / Pfs must be (B, T, d) / (B, T, d, d); got {tuple(m.shape)} / {tuple(P.shape)}')
    B, T, d = (int(v) for v in m.shape)
    if d != int(spec.d):
        raise ValueError(f'model dimension {spec.d} != data dimension {d}')
    with torch.cuda.device(m.device):
        ctx = context(m.device.index)
        handle = custom_model(spec, m.device.index)
        keep = [m, P]
        params, pstride, g, gstride = _custom_params(spec, gamma, B, keep)
        sig = _sigma_struct(sgps, d, keep, None)
        mss, Pss = torch.empty_like(m), torch.empty_like(P)
        rc = _timed('smoother', lambda: load_library().cgp_smoother_custom(ctx, handle, C.byref(sig) if sig is not None else None, _ptr(params), pstride, _ptr(g), gstride, float(dt), _ptr(m), _ptr(P),
                                                                           B, T, _ptr(mss), _ptr(Pss), int(flags), _stream()))
        _check(ctx, rc, 'cgp_smoother_custom')
        return _out(mss, like_numpy, squeeze), _out(Pss, like_numpy, squeeze)


DIR_DOUBLES = 24        # include/chirpgp_hip.h: CGP_DIR_DOUBLES


def run_ekf_nll_grad(spec, H, Xi, m0, P0, dt, ys, dirs, trials_per_record=None, record_index=None):
    """cgp_ekf_nll_grad: the EKF's final NLL and its derivative along `dirs` (B, n_dir, 24) -- forward tangents through the scan, one
    launch.  ys (T,) or (R, T) with the shared-record addressing of run_filter.  Returns (nll (B,), grad (B, n_dir)) as device tensors."""
    torch = _torch()
    ys_d = dev(ys)
    if ys_d.ndim == 1:
        ys_d = ys_d[None, :]
    R, T = int(ys_d.shape[0]), int(ys_d.shape[1])
    rep = 1 if trials_per_record is None else int(trials_per_record)
    idx_h = None
    if record_index is not None:
        idx_h = np.ascontiguousarray(np.asarray(record_index, dtype=np.int64).reshape(-1))
        if idx_h.size and (idx_h.min() < 0 or idx_h.max() >= R):
            raise ValueError(f'record_index outside 0..{R - 1}')
    B = (R if idx_h is None else int(idx_h.size)) * rep
    d = _check_dimension(spec)
    dirs_h = np.ascontiguousarray(np.asarray(dirs, dtype=np.float64))
    if dirs_h.ndim != 3 or dirs_h.shape[0] != B or dirs_h.shape[2] != DIR_DOUBLES:
        raise ValueError(f'dirs must be ({B}, n_dir, {DIR_DOUBLES}); got {dirs_h.shape}')
    n_dir = int(dirs_h.shape[1])
    with torch.cuda.device(ys_d.device):
        ctx = context(ys_d.device.index)
        keep = [ys_d]
        idx_d = None
        if idx_h is not None:
            idx_d = _index_const(idx_h.astype(np.int32))
            keep.append(idx_d)
        model = _model_struct(spec, None, B, keep)
        init = _init_struct(H, Xi, m0, P0, d, B, keep)
        dirs_d = dev(dirs_h, ys_d.device.index)
        opts = dict(dtype=torch.float64, device=ys_d.device)
        nll, grad = torch.empty((B,), **opts), torch.empty((B, n_dir), **opts)
        rc = _timed('filter', lambda: load_library().cgp_ekf_nll_grad(ctx, C.byref(model), C.byref(init), float(dt), _ptr(ys_d), T, rep, _ptr(idx_d),
                                                                      B, T, _ptr(dirs_d), n_dir, _ptr(nll), _ptr(grad), 0, _stream()))
        _check(ctx, rc, 'cgp_ekf_nll_grad')
        return nll, grad


E_UNSUPPORTED = -2
_FUNCS = {'softplus': 0, 'g': 0, 'exp': 1, 'identity': 2, 'square': 3}
_gh_cache = {}


def _gh_rule(order):
    """Nodes and weights of the reference's 1-D Gauss-Hermite rule (SigmaPoints.gauss_hermite(1, order), quadratures.py:156-196)."""
    if order not in _gh_cache:
        from chirpgp_amd.quadratures import SigmaPoints
        sg = SigmaPoints.gauss_hermite(1, order)
        _gh_cache[order] = (np.asarray(sg.xi, dtype=np.float64).reshape(-1), np.asarray(sg.w, dtype=np.float64).reshape(-1))
    return _gh_cache[order]


class SmootherSelection(dict):
    """The selected outputs of a smoother call: keys 'mean', 'var', 'expect' (those that were asked for), each (T,) or (B, T)."""
    __getattr__ = dict.get


def run_smoother(method, spec, sgps, gamma, dt, mfs, Pfs, flags=0, want=(True, True), select=None, time_split=None, split_tol=None,
                 return_junction_error=False):
    """cgp_smoother / cgp_smoother_select with NumPy / torch marshalling.  (T, d) / (T, d, d) or with a leading batch axis.

    ``select = dict(comp=k, mean=True, var=True, expect='softplus' | 'exp' | 'identity' | 'square' | None, order=10)`` asks the launch
    for the smoothed marginal of state component k as well: its mean mss[..., k], its variance Pss[..., k, k] and / or E[f(V)] by 1-D
    Gauss-Hermite (quadratures.py:234-274 -- the step behind the smoother in every driver of the reference, demos/ekfs_mle.py:69-77).
    The call then returns ``(mss, Pss, selection)``; with ``want=(False, False)`` the full rows are not written at all (None in their
    place) and a d = 4 smoother moves 8 - 24 bytes a step instead of 160 (include/chirpgp_hip.h: cgp_smoother_select)."""
    torch = _torch()
    like_numpy = not _is_torch(mfs)
    m, P = dev(mfs), dev(Pfs)
    squeeze = m.ndim == 2
    if squeeze:
        m, P = m[None], P[None]
    if m.ndim != 3 or P.ndim != 4 or P.shape[:2] != m.shape[:2] or P.shape[2] != m.shape[2] or P.shape[3] != m.shape[2]:
        raise ValueError(f'mfs / Pfs must be (B, T, d) / (B, T, d, d); got {tuple(m.shape)} / {tuple(P.shape)}')
    B, T, d = (int(s) for s in m.shape)
    if d != int(spec.d):
        raise ValueError(f'model dimension {spec.d} != data dimension {d}')
    _check_dimension(spec)
    if select is None and not (want[0] and want[1]):
        raise ValueError('a smoother without `select` writes both mss and Pss')
    with torch.cuda.device(m.device):
        ctx = context(m.device.index)
        keep = [m, P]
        model = _model_struct(spec, gamma, B, keep)
        sig = _sigma_struct(sgps, d, keep, _nonlinear_coord(spec))
        lib, st = load_library(), _stream()
        sig_ref = C.byref(sig) if sig is not None else None
        if time_split is not None:
            # cgp_smoother_time_split (cd_sgp_smoother, cd_eks): several wavefronts per trial, each starting `burn_in` steps later than its piece from the
            # filtering row there; the launch reports the mismatch at its junctions (with split_tol: fall back to the sequential smoother)
            if select is not None:
                raise ValueError('time_split writes full rows only')
            segments, burn_in = (int(v) for v in time_split)
            mss, Pss = torch.empty_like(m), torch.empty_like(P)
            err = torch.empty((B,), dtype=torch.float64, device=m.device)
            rc = _timed('smoother', lambda: lib.cgp_smoother_time_split(ctx, int(method), C.byref(model), sig_ref, float(dt), _ptr(m), _ptr(P), B, T,
                                                                        _ptr(mss), _ptr(Pss), int(flags), segments, burn_in, _ptr(err), st))
            _check(ctx, rc, 'cgp_smoother_time_split')
            junction = err
            if split_tol is not None and not bool((err <= float(split_tol)).all()):
                rc = _timed('smoother', lambda: lib.cgp_smoother(ctx, int(method), C.byref(model), sig_ref, float(dt), _ptr(m), _ptr(P), B, T,
                                                                 _ptr(mss), _ptr(Pss), int(flags), st))
                _check(ctx, rc, 'cgp_smoother')
                junction = None
            _per_thread.junction_error = junction
            res = (_out(mss, like_numpy, squeeze), _out(Pss, like_numpy, squeeze))
            return res + (junction,) if return_junction_error else res
        if select is None:
            mss, Pss = torch.empty_like(m), torch.empty_like(P)
            rc = _timed('smoother', lambda: lib.cgp_smoother(ctx, int(method), C.byref(model), sig_ref,
                                                             float(dt), _ptr(m), _ptr(P), B, T, _ptr(mss), _ptr(Pss), int(flags), st))
            _check(ctx, rc, 'cgp_smoother')
            return _out(mss, like_numpy, squeeze), _out(Pss, like_numpy, squeeze)
        unknown = set(select) - {'comp', 'mean', 'var', 'expect', 'order'}
        if unknown:
            raise ValueError(f'select: unknown keys {sorted(unknown)}')
        comp = int(select['comp'])
        if comp < 0:
            comp += d                                   # "-2": the frequency state of every chirp model
        if not 0 <= comp < d:
            raise ValueError(f'select: component {select["comp"]} outside the state dimension {d}')
        func = select.get('expect')
        if func is not None and func is not False and str(func) not in _FUNCS:
            raise ValueError(f'select: expect must be one of {sorted(_FUNCS)} (an enumerated integrand: a Python callable cannot run in a kernel)')
        want_e = func is not None and func is not False
        order = int(select.get('order', 10))
        opts = dict(dtype=torch.float64, device=m.device)
        o = CgpSmoothOut()
        o.comp, o.func, o.order = comp, (_FUNCS[str(func)] if want_e else 0), order
        sel = {}
        for key, field in (('mean', 'comp_mean'), ('var', 'comp_var')):
            if select.get(key, False):
                sel[key] = torch.empty((B, T), **opts)
                setattr(o, field, _ptr(sel[key]))
        if want_e:
            xi, w = _gh_rule(order)
            xi_d, w_d = dev_const(xi), dev_const(w)
            keep += [xi_d, w_d]
            sel['expect'] = torch.empty((B, T), **opts)
            o.expect, o.xi, o.w = _ptr(sel['expect']), _ptr(xi_d), _ptr(w_d)
        if not sel:
            raise ValueError('select asks for none of mean / var / expect')
        mss = torch.empty_like(m) if want[0] else None
        Pss = torch.empty_like(P) if want[1] else None
        o.mss, o.Pss = _ptr(mss), _ptr(Pss)
        rc = _timed('smoother', lambda: lib.cgp_smoother_select(ctx, int(method), C.byref(model), sig_ref, float(dt), _ptr(m), _ptr(P), B, T,
                                                                C.byref(o), int(flags), st))
        if rc == E_UNSUPPORTED and (mss is None or Pss is None):
            # a kernel that can only gather the selection from its full rows: give it (temporary) rows
            tm = mss if mss is not None else torch.empty_like(m)
            tP = Pss if Pss is not None else torch.empty_like(P)
            o.mss, o.Pss = _ptr(tm), _ptr(tP)
            rc = _timed('smoother', lambda: lib.cgp_smoother_select(ctx, int(method), C.byref(model), sig_ref, float(dt), _ptr(m), _ptr(P), B, T,
                                                                    C.byref(o), int(flags), st))
            keep += [tm, tP]
        _check(ctx, rc, 'cgp_smoother_select')
        selection = SmootherSelection({k: _out(v, like_numpy, squeeze) for k, v in sel.items()})
        return (None if mss is None else _out(mss, like_numpy, squeeze), None if Pss is None else _out(Pss, like_numpy, squeeze), selection)


FN_SOFTPLUS, FN_EXP, FN_IDENTITY, FN_SQUARE = 0, 1, 2, 3


def gaussian_expectation(ms, chol_Ps, xi, w, func=FN_SOFTPLUS):
    """E[f(V)] for scalar marginals on the device (f enumerated: FN_*) -> (n, 1) like the reference's force_shape=True call."""
    torch = _torch()
    like_numpy = not _is_torch(ms)
    m, s = dev(ms).reshape(-1), dev(chol_Ps).reshape(-1)
    if m.shape != s.shape:
        raise ValueError('ms and chol_Ps must have the same number of entries')
    with torch.cuda.device(m.device):
        ctx = context(m.device.index)
        xi_d, w_d = dev(xi).reshape(-1), dev(w).reshape(-1)
        out = torch.empty_like(m)
        rc = load_library().cgp_gaussian_expectation_fn(ctx, int(func), _ptr(m), _ptr(s), m.numel(), 1, _ptr(xi_d), _ptr(w_d), xi_d.numel(),
                                                        _ptr(out), _stream())
        _check(ctx, rc, 'cgp_gaussian_expectation_fn')
        out = out.reshape(-1, 1)
        return out.cpu().numpy() if like_numpy else out


def debug_math(op, x):
    """In-kernel elementary functions on the device (test hook): returns (out0, out1) as NumPy arrays."""
    torch = _torch()
    xd = dev(x).reshape(-1)
    with torch.cuda.device(xd.device):
        ctx = context(xd.device.index)
        o0, o1 = torch.empty_like(xd), torch.empty_like(xd)
        rc = load_library().cgp_debug_math(ctx, int(op), _ptr(xd), xd.numel(), _ptr(o0), _ptr(o1), _stream())
        _check(ctx, rc, 'cgp_debug_math')
        return o0.cpu().numpy(), o1.cpu().numpy()


def _init_struct(H, Xi, m0, P0, d, B, keep):
    init = CgpInit()
    if H is not None:
        Ht, init.H_stride = _batched_operand(H, 1, B, 'H')
        if Ht.shape[-1] != d:
            raise ValueError(f'H must have {d} entries')
        init.H = _ptr(Ht)
        keep.append(Ht)
    if Xi is not None:
        Xit = dev_const(np.asarray(Xi, dtype=np.float64).reshape(-1)) if not _is_torch(Xi) else dev(Xi.reshape(-1))
        if Xit.numel() not in (1, B):
            raise ValueError('Xi must be a scalar or have one entry per trial')
        init.Xi, init.Xi_stride = _ptr(Xit), (1 if Xit.numel() > 1 else 0)
        keep.append(Xit)
    m0t, init.m0_stride = _batched_operand(m0, 1, B, 'm0')
    if m0t.shape[-1] != d:
        raise ValueError(f'm0 must be ({d},)')
    init.m0 = _ptr(m0t)
    keep.append(m0t)
    if P0 is not None:
        P0t, init.P0_stride = _batched_operand(P0, 2, B, 'P0')
        if tuple(P0t.shape[-2:]) != (d, d):
            raise ValueError(f'P0 must be ({d}, {d})')
        init.P0 = _ptr(P0t)
        keep.append(P0t)
    return init


def run_simulate(spec, H, Xi, m0, P0, dt, T, seed, B, trial0=0, want=(True, True), flags=0, device=None):
    """cgp_simulate: B trajectories xs (B, T, d) and measurements ys (B, T) as CUDA tensors (None where not wanted)."""
    torch = _torch()
    d = _check_dimension(spec)
    B, T = int(B), int(T)
    with torch.cuda.device(torch.cuda.current_device() if device is None else device):
        ctx = context(device)
        keep = []
        model = _model_struct(spec, None, B, keep)
        init = _init_struct(H if want[1] else None, Xi if want[1] else None, m0, P0, d, B, keep)
        opts = dict(dtype=torch.float64, device=torch.device('cuda', torch.cuda.current_device() if device is None else device))
        xs = torch.empty((B, T, d), **opts) if want[0] else None
        ys = torch.empty((B, T), **opts) if want[1] else None
        lib, st = load_library(), _stream()
        fl = int(flags) | (SIM_FIXED_X0 if P0 is None else 0)
        rc = _timed('simulate', lambda: lib.cgp_simulate(ctx, C.byref(model), C.byref(init), float(dt), int(seed) & (2 ** 64 - 1),
                                                         int(trial0), B, T, _ptr(xs), _ptr(ys), fl, st))
        _check(ctx, rc, 'cgp_simulate')
        return xs, ys


def add_noise(clean, Xi, seed, B, trial0=0):
    """cgp_add_noise: (B, T) CUDA tensor clean + sqrt(Xi) e; clean is (T,) (shared) or (B, T)."""
    torch = _torch()
    c = dev(clean)
    if c.ndim == 1:
        stride, T = 0, int(c.shape[0])
    elif c.ndim == 2 and c.shape[0] == B:
        stride, T = int(c.shape[1]), int(c.shape[1])
    else:
        raise ValueError(f'clean must be (T,) or ({B}, T), got {tuple(c.shape)}')
    with torch.cuda.device(c.device):
        Xit = dev_const(np.asarray(Xi, dtype=np.float64).reshape(-1)) if not _is_torch(Xi) else dev(Xi.reshape(-1))
        if Xit.numel() not in (1, B):
            raise ValueError('Xi must be a scalar or have one entry per trial')
        ctx = context(c.device.index)
        ys = torch.empty((int(B), T), dtype=torch.float64, device=c.device)
        rc = _timed('add_noise', lambda: load_library().cgp_add_noise(ctx, _ptr(c), stride, _ptr(Xit), 1 if Xit.numel() > 1 else 0,
                                                                      int(seed) & (2 ** 64 - 1), int(trial0), int(B), T, _ptr(ys), _stream()))
        _check(ctx, rc, 'cgp_add_noise')
        return ys


def debug_philox(ctr, key):
    """Raw Philox4x32-10 blocks on the device (test hook): ctr (n, 4) uint32, key (2,) uint32 -> (n, 4) uint32."""
    torch = _torch()
    c = torch.from_numpy(np.ascontiguousarray(np.asarray(ctr, dtype=np.uint32)).view(np.int32)).cuda()
    k = torch.from_numpy(np.ascontiguousarray(np.asarray(key, dtype=np.uint32)).view(np.int32)).cuda()
    ctx = context(c.device.index)
    out = torch.empty_like(c)
    rc = load_library().cgp_debug_philox(ctx, _ptr(c), _ptr(k), c.shape[0], _ptr(out), _stream())
    _check(ctx, rc, 'cgp_debug_philox')
    return out.cpu().numpy().view(np.uint32)


DBG_WALK_SEGMENTS, DBG_COUNT_REGIMES, DBG_LANE_BUFFERS = 1, 2, 3
REGIME_COUNTERS = ('high', 'common', 'redone', 'checked', 'high_left', 'wide', 'low', 'mid')


def reserve_workspace(nbytes, device_index=None):
    """Size the current stream's scratch buffer of the time-split launches ahead of time (include/chirpgp_hip.h:
    cgp_reserve_workspace) -- needed only before capturing such launches into a graph."""
    ctx = context(device_index)
    _check(ctx, load_library().cgp_reserve_workspace(ctx, int(nbytes), _stream()), 'cgp_reserve_workspace')


def release_workspace(device_index=None):
    """Free the current stream's scratch buffer and forget the stream (include/chirpgp_hip.h: cgp_release_workspace)."""
    ctx = context(device_index)
    _check(ctx, load_library().cgp_release_workspace(ctx, _stream()), 'cgp_release_workspace')


def debug_set(key, value, device_index=None):
    """Per-context tuning / measurement knob (include/chirpgp_hip.h: cgp_debug_set)."""
    ctx = context(device_index)
    _check(ctx, load_library().cgp_debug_set(ctx, int(key), int(value)), 'cgp_debug_set')


def debug_counters(reset=True, device_index=None):
    """The context's regime counters as a dict (waits for the current stream): chunks of 64 steps the d = 4 matrix-core EKF kept from
    its HIGH / common regime, repeated with the checked step, ran on the checked step afterwards, and tried in HIGH but left it."""
    ctx = context(device_index)
    out = (C.c_uint64 * 8)()
    _check(ctx, load_library().cgp_debug_counters(ctx, out, 1 if reset else 0, _stream()), 'cgp_debug_counters')
    return {k: int(out[i]) for i, k in enumerate(REGIME_COUNTERS)}


def squared_error_sums(a, r, comps, sums=None):
    """sums[c][0][t] += sum_b (a - r)[b, t, comps[c]]^2, sums[c][1][t] += sum_b of its square (include/chirpgp_hip.h:
    cgp_squared_error_sums) -- the per-step error statistics of the reference's CRLB jobs, reduced on the device.  a, r: (B, T, d)
    device tensors; returns the (n, 2, T) tensor of sums (a fresh zeroed one unless `sums` is passed to accumulate into)."""
    torch = _torch()
    a = dev(a)
    r = dev(r, a.device.index)
    if r.device != a.device:
        r = r.to(a.device)
    if a.shape != r.shape or a.ndim != 3:
        raise ValueError(f'a and r must both be (B, T, d); got {tuple(a.shape)}, {tuple(r.shape)}')
    B, T, d = (int(v) for v in a.shape)
    comps = [int(c) for c in comps]
    with torch.cuda.device(a.device):
        ctx = context(a.device.index)
        if sums is None:
            sums = torch.zeros((len(comps), 2, T), dtype=torch.float64, device=a.device)
        elif (not _is_torch(sums) or tuple(sums.shape) != (len(comps), 2, T) or not sums.is_contiguous() or sums.dtype != torch.float64
              or not sums.is_cuda or sums.device != a.device):
            # the kernel adds float64 atomically through the raw pointer: anything else is a wrong result or a fault
            raise ValueError(f'sums must be a contiguous float64 ({len(comps)}, 2, {T}) tensor on {a.device}')
        arr = (C.c_int32 * len(comps))(*comps)
        for lo in range(0, max(B, 1), 65535 * 512):                    # the kernel's grid limit: slabs of 512 trials
            hi = min(B, lo + 65535 * 512)
            rc = load_library().cgp_squared_error_sums(ctx, _ptr(a[lo:hi]), _ptr(r[lo:hi]), hi - lo, T, d, arr, len(comps), _ptr(sums), _stream())
            _check(ctx, rc, 'cgp_squared_error_sums')
    return sums
