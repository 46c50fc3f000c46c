"""Timing of the on-device simulators (cgp_simulate / cgp_add_noise) at the Monte-Carlo sizes of the reference's jobs."""
import json
import sys
import numpy as np
import torch

sys.path.insert(0, '.')
from chirpgp_amd import _engine as E, models as pm, toymodels   # noqa: E402


def timed(fn, reps=3):
    """Kernel time from the events the engine records around the C-ABI call (allocation of the outputs excluded)."""
    fn()
    torch.cuda.synchronize()
    E.kernel_events = []
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for _, a, b in E.kernel_events]
    E.kernel_events = None
    return sum(ms) / len(ms)


def main():
    drift, disp, disc, m0, P0, H = pm.build_chirp_model(np.array([0.1, 0.1, 0.1, 1., 1., 7.]))
    rows = []
    for B, T, flags, what in [(1000, 10000, 0, 'C2 shape'), (1000, 10000, 4, 'C2 shape, lane per trial'),
                              (8192, 2000, 0, ''), (8192, 2000, 2, 'wave per trial'),
                              (200000, 500, 0, 'crlb_ekf.py shape / 5'), (1000000, 500, 0, 'crlb_ekf.py: 1e6 x 500')]:
        for want in [(True, True), (False, True)]:
            ms = timed(lambda: E.run_simulate(disc, H, 0.1, m0, P0, 1e-3, T, 1, B, want=want, flags=flags))
            nbytes = B * T * 8 * ((4 if want[0] else 0) + 1)
            rows.append(dict(kernel='simulate', B=B, T=T, flags=flags, note=what, xs=want[0], ms=round(ms, 3),
                             steps_per_s=B * T / ms * 1e3, GBs=nbytes / ms / 1e6))
            print(json.dumps(rows[-1]), flush=True)
    clean = np.sin(np.arange(10000) * 0.01)
    for B in (1000, 100000):
        ms = timed(lambda: toymodels.noisy_copies(clean, 0.1, 3, B))
        rows.append(dict(kernel='add_noise', B=B, T=10000, ms=round(ms, 3), GBs=B * 10000 * 8 / ms / 1e6))
        print(json.dumps(rows[-1]), flush=True)


if __name__ == '__main__':
    main()
