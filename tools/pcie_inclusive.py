import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, bench
from chirpgp_amd import filters_smoothers as fs
wl = bench.make_workload(1000, 10000)
for _ in range(2):
    t0 = time.perf_counter()
    f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], wl['ys'])
    s = fs.eks(wl['disc'], f[0], f[1], wl['dt'])
    t1 = time.perf_counter()
print(f'numpy in -> numpy out (PCIe both ways, pageable): {t1 - t0:.3f} s  = {1e7 / (t1 - t0):.3e} trial-steps/s')
ys = torch.from_numpy(wl['ys']).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
yd = ys.cuda(non_blocking=True); f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], yd); s = fs.eks(wl['disc'], f[0], f[1], wl['dt'])
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'pinned ys upload + kernels, results left on device: {(t1 - t0) * 1e3:.2f} ms')
