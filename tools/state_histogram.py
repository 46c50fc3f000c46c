import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs
wl = bench.make_workload(4, 100, kind='ekf')
for Xi, off in ((0.1, 5.5), (0.1, 20.0), (1.0, 8.0), (0.01, 20.0), (1.0, 5.5)):
    ys = torch.from_numpy(bench.chirp_batch(200, 10000, 0, Xi=Xi, offset=off)).cuda()
    m = fs.ekf(wl['disc'], wl['H'], Xi, wl['m0'], wl['P0'], 1e-3, ys)[0][:, :, 2]
    q = torch.tensor([0.001, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99], device='cuda', dtype=torch.float64)
    v = m.flatten()[::7]
    print(f'Xi={Xi} off={off}: share < -1.5: {float((m < -1.5).double().mean()):.3f}  in (-1.5, 1.5): {float(((m >= -1.5) & (m < 1.5)).double().mean()):.3f}  >= 1.5: {float((m >= 1.5).double().mean()):.3f}  >= 6: {float((m >= 6).double().mean()):.3f}; quantiles', [round(float(x), 2) for x in torch.quantile(v, q)])
