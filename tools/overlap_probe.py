"""Does the smoother of batch k hide under the filter of batch k + 1?  Config C2 (1000 x 10^4, EKF + EKS), two HIP streams:
the filter kernel keeps one wavefront per SIMD busy on a latency chain, so the HBM-bound smoother of the previous batch can run
beside it.  Prints ms per pass for the serial loop (bench.py's) and the two-stream pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chirpgp_amd import filters_smoothers as fs

B, T, K = 1000, 10000, 20
wl = bench.make_workload(B, T, seed=0, kind='ekf')
ys = torch.from_numpy(wl['ys']).cuda()
filt = lambda: fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys)
smooth = lambda f: fs.eks(wl['disc'], f[0], f[1], wl['dt'])

def serial(n):
    for _ in range(n):
        f = filt(); s = smooth(f)
    return s

def piped(n, F, S):
    prev = None
    for k in range(n):
        with torch.cuda.stream(F):
            f = filt()
            ev = torch.cuda.Event(); ev.record(F)
        with torch.cuda.stream(S):
            S.wait_event(ev)
            for t in f[:2]:
                t.record_stream(S)
            s = smooth(f)
        prev = (f, s)
    return prev

for name, run in (('serial', serial), ('two streams', None)):
    if run is None:
        F, S = torch.cuda.Stream(), torch.cuda.Stream()
        run = lambda n: piped(n, F, S)
    keep = [run(2), run(2)]; torch.cuda.synchronize(); del keep
    run(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = run(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f'{name:12s}: {dt * 1e3:.3f} ms per pass = {B * T / dt:.3e} trial-steps/s', flush=True)
