"""tools/fill_rate.py: what this GPU's HBM takes as a pure WRITE stream (torch fill_ of 16 GB), as a copy and as a pure read (sum):
the roof of a kernel that is 95 % stores (the large-batch filters write 168 B and read 8 B per trial-step)."""
import torch
n = 2 * 1024 ** 3                              # doubles: 16 GB
a = torch.empty(n, dtype=torch.float64, device='cuda')
b = torch.empty(n, dtype=torch.float64, device='cuda')


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


t = timed(lambda: a.fill_(1.0)); print(f'fill  16 GB: {t:.2f} ms  {16 * 1.073741824 / t:.2f} TB/s written')
t = timed(lambda: b.copy_(a)); print(f'copy  16 GB: {t:.2f} ms  {2 * 16 * 1.073741824 / t:.2f} TB/s read + written')
t = timed(lambda: a.sum()); print(f'sum   16 GB: {t:.2f} ms  {16 * 1.073741824 / t:.2f} TB/s read')
