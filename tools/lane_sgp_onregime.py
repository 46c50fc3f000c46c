"""tools/lane_sgp_onregime.py [B] [T]: the one-lane-per-trial sigma-point filter (GH-3) on the BENCH's records (frequency state inside the
lean regime: the speculative fan of cgp_steps.hpp: sgp4_prediction_collapsed_impl), kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chirpgp_amd import filters_smoothers as fs, _engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
wl = bench.make_workload(4, 100, kind='sgp')
ys = torch.from_numpy(bench.chirp_batch(64, T, 0)).cuda().repeat(B // 64, 1)
run = lambda: fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], 0.1, wl['m0'], wl['P0'], 1e-3, ys, flags=_engine.THREAD_PER_TRIAL, want=(True, False, False))
run(); torch.cuda.synchronize()
ev = _engine.kernel_events = []
for _ in range(3): run()
torch.cuda.synchronize(); _engine.kernel_events = None
print(f'sgp_filter lane {B} x {T} on the bench records: {min(a.elapsed_time(b) for _, a, b in ev):.3f} ms', flush=True)
