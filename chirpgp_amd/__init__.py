"""chirpgp_amd -- MI355X-native batched Kalman / RTS engine behind the function signatures of
spdes/chirpgp's chirpgp/filters_smoothers.py.

Host code is Python; all filtering / smoothing arithmetic runs in hand-written gfx950 HIP kernels reached
through the C-ABI of include/chirpgp_hip.h (ctypes).  There is no CPU fallback: the filters raise if
libchirpgp_hip.so or a GPU is missing.
"""
from chirpgp_amd import models, quadratures            # host-side descriptors (NumPy only)
from chirpgp_amd.quadratures import SigmaPoints

__all__ = ['models', 'quadratures', 'SigmaPoints', 'filters_smoothers']
__version__ = '0.1.0'


def __getattr__(name):
    if name == 'filters_smoothers':
        import importlib
        return importlib.import_module('chirpgp_amd.filters_smoothers')
    raise AttributeError(name)
