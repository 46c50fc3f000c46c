import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from tests.test_gpu_lane4 import _crlb, _filters
from tests import cases as cs
hip, ref = _filters('ekf')
mc, H, Xi, m0, P0, dt, yss = _crlb(500, 4096)
want = ref(mc, H, Xi, m0, P0, dt, yss.cpu().numpy())
for name, fl in (('lane4', 0x4), ('generic lane', 0x14), ('wave', 0x2)):
    got = hip(mc, H, Xi, m0, P0, dt, yss, flags=fl)
    e = np.abs(got[0].cpu().numpy() - want[0])
    tol = 1e-9 * (np.abs(want[0]) + 1e-3 * np.abs(want[0]).max())
    print(name, 'max abs err', e.max(), 'rel to max', e.max() / np.abs(want[0]).max(), 'worst elementwise x', (e / tol).max())
