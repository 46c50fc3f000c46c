#!/usr/bin/env python3
"""bench.py -- filter + smoother throughput of the MI355X engine on BASELINE.json's headline configuration.

A "step" is one pass of the hot path over one batch of synthetic input: the filter launch (ys -> mfs, Pfs, nll) and
the smoother launch (mfs, Pfs -> mss, Pss), inputs already resident in HBM.  Default workload = BASELINE config C2:
discrete EKF + EKS of the demos' chirp model (demos/ekfs_mle.py), d = 4, T = 10 000, B = 1000 Monte-Carlo trials per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ekf|sgp|cd_sgp|cd_ekf|harmonic] [--batch B] [--T T]

N > 1 is launched by torch.distributed.run (one rank per GPU over RCCL); trials shard across ranks with no data-path
collective (weak scaling: every rank runs B trials); one all_gather of the per-trial final NLL happens after the
timed region and is reported as gather_ms.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def chirp_batch(B, T, seed, dt=1e-3, Xi=0.1, num_harmonics=0):
    """Synthetic toy chirp of SURVEY.md 8d: meow frequency law (toymodels.py:226-268) tiled in 3141-step windows,
    constant magnitude 1, y = chirp + sqrt(Xi) N(0, 1); trial i uses numpy default_rng(seed + i)."""
    k = np.arange(T)
    window = 3141
    local = (k % window + 1) * dt
    phase = (k // window) * (8.0 * window * dt) + 500.0 * np.exp(-5.0 / np.sin(local)) + 8.0 * local
    if num_harmonics == 0:
        clean = np.sin(2 * math.pi * phase)
    else:
        clean = sum(np.sin((h + 1) * 2 * math.pi * phase) for h in range(num_harmonics))
    ys = np.empty((B, T))
    for i in range(B):
        ys[i] = clean + math.sqrt(Xi) * np.random.default_rng(seed + i).standard_normal(T)
    return ys


def make_workload(B, T, seed=0, kind='ekf'):
    """Model at the demos' MLE start point [lam, b, delta, ell, sigma, m0_v] = [0.1, 0.1, 0.1, 1, 1, 7]
    (demos/ekfs_mle.py:39) and B x T synthetic measurements."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.])
    wl = dict(kind=kind, dt=1e-3, Xi=0.1, B=B, T=T)
    if kind == 'harmonic':
        drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, 3)
        wl.update(ys=chirp_batch(B, T, seed, num_harmonics=3), sgps=SigmaPoints.cubature(8), d=8)
    else:
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        wl.update(ys=chirp_batch(B, T, seed), sgps=SigmaPoints.gauss_hermite(4, 3), d=4)
    wl.update(drift=drift, disp=disp, disc=disc, m0=m0, P0=P0, H=H)
    return wl


def bytes_per_trial_step(d):
    """filter reads y (8 B), writes mf, Pf, nll (8d + 8d^2 + 8); smoother reads and writes 8d + 8d^2."""
    filt = 8 + 8 * d + 8 * d * d + 8
    smooth = 2 * (8 * d + 8 * d * d)
    return filt, smooth


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of this same command
    (profiles/r01_v17_ekf_eks_pmc.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units,
    FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM").  None if no profile is committed for that kernel."""
    path = os.path.join(ROOT, 'profiles', 'r01_v17_ekf_eks_pmc.json')
    try:
        prof = json.load(open(path))
    except OSError:
        return None
    for name, counters in prof.items():
        if kernel_key in name and 'hbm_bytes_per_launch' in counters:
            return counters['hbm_bytes_per_launch']
    return None


def pmc_issue(kernel_key, units):
    """Instructions and cycles per trial-step of the dominant kernel from the same committed profile (SQ_INSTS_VALU,
    SQ_INSTS_SALU, SQ_WAVE_CYCLES x 4): why a T-serial kernel sits far below the HBM roof at B = 1000."""
    try:
        prof = json.load(open(os.path.join(ROOT, 'profiles', 'r01_v17_ekf_eks_pmc.json')))
    except OSError:
        return None
    for name, c in prof.items():
        if kernel_key in name and all(k in c for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_WAVES')):
            return {"valu_per_step": c['SQ_INSTS_VALU']['mean'] / units, "salu_per_step": c['SQ_INSTS_SALU']['mean'] / units,
                    "cycles_per_step": 4 * c['SQ_WAVE_CYCLES']['mean'] / units, "waves": c['SQ_WAVES']['mean'],
                    "source": "profiles/r01_v17_ekf_eks_pmc.json"}
    return None


def cpu_baseline(wl, target_seconds=12.0):
    """The oracle's C port (oracle/c/port.c, OpenMP over trials) timed on the host cores on a bounded sample of the
    SAME workload: as many trials (full T) as fit ~target_seconds, at least one per thread."""
    from oracle import port
    import copy
    k = wl['kind']
    threads = port.num_threads()
    T = wl['T']
    ys = wl['ys']
    n = min(ys.shape[0], max(threads, 8))
    drift_g = copy.copy(wl['drift'])
    drift_g.gamma = wl['disp'].outer()
    label = {'ekf': 'EKF+EKS', 'sgp': 'sgp_filter+sgp_smoother', 'harmonic': 'sgp_filter+sgp_smoother (cubature, d=8)',
             'cd_sgp': 'cd_sgp_filter+cd_sgp_smoother', 'cd_ekf': 'cd_ekf+cd_eks'}[k]

    def once(nn):
        t0 = time.perf_counter()
        a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys[:nn])
        if k == 'ekf':
            f = port.filter(port.F_EKF, wl['disc'], None, *a)
            port.smoother(port.S_EKS, wl['disc'], None, wl['dt'], f[0], f[1])
        elif k in ('sgp', 'harmonic'):
            f = port.filter(port.F_SGP, wl['disc'], wl['sgps'], *a)
            port.smoother(port.S_SGP, wl['disc'], wl['sgps'], wl['dt'], f[0], f[1])
        elif k == 'cd_sgp':
            f = port.filter(port.F_CD_SGP, drift_g, wl['sgps'], *a)
            port.smoother(port.S_CD_SGP, drift_g, wl['sgps'], wl['dt'], f[0], f[1])
        else:
            f = port.filter(port.F_CD_EKF, drift_g, None, *a)
            port.smoother(port.S_CD_EKS, drift_g, None, wl['dt'], f[0], f[1])
        return time.perf_counter() - t0
    t = once(n)                                   # warm-up + calibration
    per_trial = t / n
    n2 = int(min(ys.shape[0], max(n, target_seconds / max(per_trial, 1e-9))))
    n2 = max(threads, (n2 // threads) * threads)
    n2 = min(n2, ys.shape[0])
    best = min(once(n2) for _ in range(2))
    return {"value": n2 * T / best, "unit": "trial-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n2} of {ys.shape[0]} trials x T={T} ({label}, oracle/c/port.c, OpenMP, best of 2, {best:.2f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='ekf', choices=['ekf', 'sgp', 'cd_sgp', 'cd_ekf', 'harmonic'])
    ap.add_argument('--batch', type=int, default=None, help='trials per GPU (default: BASELINE config)')
    ap.add_argument('--T', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--flags', type=int, default=0, help='CGP_* flag bits forwarded to the engine (e.g. 2 = wave per trial)')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: the batch is the TOTAL over all ranks and is sharded (default: weak, --batch trials per rank)')
    ap.add_argument('--rehearse', action='store_true',
                    help='multi-process dry run on fewer GPUs than ranks: ranks share devices and the collectives go over '
                         'gloo on host copies (RCCL refuses two ranks on one device); timings are then meaningless')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU')
    if args.rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    def coll(t):
        """Tensor as the process group wants it: HBM for RCCL, a host copy for the gloo rehearsal."""
        return t.cpu() if args.rehearse else t

    from chirpgp_amd import filters_smoothers as fs

    defaults = {'ekf': (1000, 10000), 'sgp': (1000, 10000), 'cd_sgp': (512, 50000), 'cd_ekf': (1000, 10000), 'harmonic': (1000, 10000)}
    B = args.batch or defaults[args.workload][0]
    if args.strong and world > 1:                     # contiguous shard of the total batch
        from chirpgp_amd.parallel import shard_bounds
        lo, hi = shard_bounds(B, rank, world)
        B_total, B = B, hi - lo
    else:
        B_total = B * world
    T = args.T or defaults[args.workload][1]
    wl = make_workload(B, T, seed=1000003 * rank, kind=args.workload)
    d = wl['d']
    ys_dev = torch.from_numpy(wl['ys']).cuda()
    kw = dict(flags=args.flags) if args.flags else {}

    from chirpgp_amd import _engine

    def step(record=None):
        k = wl['kind']
        if k == 'ekf':
            f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
        elif k in ('sgp', 'harmonic'):
            f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
        elif k == 'cd_sgp':
            f = fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
        else:
            f = fs.cd_ekf(wl['drift'], wl['disp'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
        if k == 'ekf':
            s = fs.eks(wl['disc'], f[0], f[1], wl['dt'], **kw)
        elif k in ('sgp', 'harmonic'):
            s = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'], **kw)
        elif k == 'cd_sgp':
            s = fs.cd_sgp_smoother(wl['drift'], wl['disp'](None), wl['sgps'], f[0], f[1], wl['dt'], **kw)
        else:
            s = fs.cd_eks(wl['drift'], wl['disp'], f[0], f[1], wl['dt'], **kw)
        return f, s

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Allocator priming (untimed, before the W warm-up steps): the timed loop keeps one result alive while the next step
    # allocates its outputs, so it alternates between two sets of output buffers; two passes whose results are held
    # together make PyTorch's caching allocator own both sets, and no hipMalloc of GB-sized buffers lands in the timed
    # region whatever --warmup is.  Device memory management is not part of the path being measured.
    prime = [step(), step()]
    torch.cuda.synchronize()
    del prime
    for _ in range(args.warmup):
        step()
    sync()
    # HIP events on the launch stream, recorded immediately around each C-ABI call (kernel duration, not host time)
    events = _engine.kernel_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f, s = step(events)
    sync()
    elapsed = time.perf_counter() - t0
    _engine.kernel_events = None
    tmax = coll(torch.tensor([elapsed], dtype=torch.float64, device='cuda'))
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # the single collective of the path: gather the per-trial final NLL (B doubles per rank) -- after the timed region
    gather_ms = None
    if world > 1:
        from chirpgp_amd import parallel
        last = coll(f[2][:, -1].contiguous())
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        out = parallel.all_gather_trials(last, B_total)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert out.shape[0] == B_total

    if rank == 0:
        filt_ms = float(np.mean([a.elapsed_time(b) for n, a, b in events if n == 'filter']))
        smooth_ms = float(np.mean([a.elapsed_time(b) for n, a, b in events if n == 'smoother']))
        bf, bs = bytes_per_trial_step(d)
        units = B * T
        dom = ('filter', filt_ms, bf) if filt_ms >= smooth_ms else ('smoother', smooth_ms, bs)
        achieved = dom[2] * units / (dom[1] * 1e-3) / 1e9
        total_units = B_total * T                       # all ranks (rank 0's own share is `units`)
        total_gbs = (bf + bs) * total_units / (elapsed / args.steps) / 1e9
        result = {
            "metric": "filter+smoother trial-steps/s (batch x T / wall)",
            "value": total_units * args.steps / elapsed,
            "unit": "trial-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (args.strong and world > 1) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": {"ekf": "C2: discrete EKF+EKS (demos/ekfs_mle.py model)", "sgp": "C3: Gauss-Hermite order-3 sgp_filter+sgp_smoother",
                                    "cd_sgp": "C4: cd_sgp_filter+cd_sgp_smoother RK4", "cd_ekf": "cd_ekf+cd_eks RK4",
                                    "harmonic": "C5: 3-harmonic chirp, cubature sgp_filter+sgp_smoother"}[args.workload],
                       "d": d, "T": T, "batch_per_gpu": B, "global_batch": B_total, "parallelism": f"trials sharded x{world}",
                       "sigma_points": int(wl['sgps'].n_points) if args.workload != 'ekf' and args.workload != 'cd_ekf' else None},
            "hbm_gbs_total": total_gbs, "hbm_frac_of_peak_total": total_gbs / (HBM_PEAK_GBS * world),
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (pmc_traffic('ekf4_mfma' if dom[0] == 'filter' else 'tp_smoother')
                                     if (args.workload == 'ekf' and B == 1000 and T == 10000 and not args.flags) else None),
                         "algorithmic_bytes_per_launch": dom[2] * units, "avg_launch_ms": dom[1]},
            "kernels": {"filter_ms": filt_ms, "smoother_ms": smooth_ms,
                        "filter_GBs": bf * units / (filt_ms * 1e-3) / 1e9, "smoother_GBs": bs * units / (smooth_ms * 1e-3) / 1e9},
            "gather_ms": gather_ms,
        }
        if args.workload == 'ekf' and B == 1000 and T == 10000 and not args.flags:
            result["issue_profile"] = pmc_issue('ekf4_mfma' if dom[0] == 'filter' else 'tp_smoother', units)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wl)
            if result["cpu_baseline"]:
                result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
