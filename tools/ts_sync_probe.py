"""Does a device synchronisation between calls make the time-split smoother's scratch allocation expensive?
(hipMallocAsync from the default pool: its release threshold is 0, so a sync may hand the memory back.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chirpgp_amd import filters_smoothers as fs
wl = bench.make_workload(125, 10000, kind='ekf')
ys = torch.from_numpy(wl['ys']).cuda()
f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
for mode in ('no sync between calls', 'torch.cuda.synchronize() between calls'):
    fs.eks(wl['disc'], f[0], f[1], wl['dt']); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        if mode.startswith('torch'):
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = fs.eks(wl['disc'], f[0], f[1], wl['dt'])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f'{mode:45s} host wall per call incl. sync: min {min(ts):.3f} ms  median {sorted(ts)[5]:.3f} ms')
