#!/usr/bin/env python3
"""bench.py -- filter + smoother throughput of the MI355X engine on BASELINE.json's headline configuration.

A "step" is one pass of the hot path over one batch of synthetic input: the filter launch (ys -> mfs, Pfs, nll) and
the smoother launch (mfs, Pfs -> mss, Pss), inputs already resident in HBM.  Default workload = BASELINE config C2:
discrete EKF + EKS of the demos' chirp model (demos/ekfs_mle.py), d = 4, T = 10 000, B = 1000 Monte-Carlo trials per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ekf|sgp|cd_sgp|cd_ekf|harmonic] [--batch B] [--T T]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (torch.distributed.run as a child
process, one rank per GPU over RCCL; the parent never touches the GPU); under an external torchrun it checks that
WORLD_SIZE == N.  Trials shard across ranks with no data-path collective; one all_gather of the per-trial final NLL
happens after the timed region and is reported as gather_ms.  Scaling: C2 (`ekf`) is weak by default (B = 1000 per
rank) and also reports the strong figure (B = 1000 in total) under "strong"; C3 / C5 (`sgp`, `harmonic`) are strong
by default ("batch=1000 sharded 8 GPUs", BASELINE.json); `--scaling weak|strong` overrides.  Rank 0 prints ONE JSON line.
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
# rocprofv3 --pmc passes of the default command (tools/profile.sh), committed; bench.py quotes its traffic / issue figures
PMC_PROFILE = 'profiles/r02_ekf_pmc.json'


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
    if kind in ('harmonic', 'harmonic_ekf'):
        drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, 3)
        wl.update(ys=chirp_batch(B, T, seed, num_harmonics=3), sgps=SigmaPoints.cubature(8), d=8)
    else:
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        wl.update(ys=chirp_batch(B, T, seed), sgps=SigmaPoints.gauss_hermite(4, 3), d=4)
    wl.update(drift=drift, disp=disp, disc=disc, m0=m0, P0=P0, H=H)
    return wl


def frozen_frequency_linear_model(params, dt):
    """BASELINE config C1's linear model: the chirp LCD discretisation with the frequency frozen at its initial value,
    F = blockdiag(exp(-lam dt) Rot(2 pi g(m0_v) dt), M32_F) (models.py:296-301), Sigma = blockdiag(q, q, M32_Sigma)
    (models.py:302-308) -- what `kf` / `rts` run on in the plumbing configuration."""
    from chirpgp_amd import models as pm
    lam, b, delta, ell, sigma, m0_v = (float(x) for x in params)
    th = dt * 2 * math.pi * float(pm.g(m0_v))
    rot = math.exp(-lam * dt) * np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
    Fm, Sm = pm._m32(ell, sigma, dt)
    q = b ** 2 * dt if lam == 0. else b ** 2 / (2 * lam) * (1 - math.exp(-2 * lam * dt))
    return pm._blkdiag([rot, Fm]), pm._blkdiag([q, q, Sm])


def bytes_per_trial_step(d):
    """filter reads y (8 B), writes mf, Pf, nll (8d + 8d^2 + 8); smoother reads and writes 8d + 8d^2."""
    filt = 8 + 8 * d + 8 * d * d + 8
    smooth = 2 * (8 * d + 8 * d * d)
    return filt, smooth


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of this same command
    (profiles/r01_v17_ekf_eks_pmc.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units,
    FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM").  None if no profile is committed for that kernel."""
    path = os.path.join(ROOT, PMC_PROFILE)
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
        prof = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
    except OSError:
        return None
    for name, c in prof.items():
        if kernel_key in name and all(k in c for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_WAVES')):
            return {"valu_per_step": c['SQ_INSTS_VALU']['mean'] / units, "salu_per_step": c['SQ_INSTS_SALU']['mean'] / units,
                    "cycles_per_step": 4 * c['SQ_WAVE_CYCLES']['mean'] / units, "waves": c['SQ_WAVES']['mean'],
                    "source": PMC_PROFILE}
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
    label = {'ekf': 'EKF+EKS', 'harmonic_ekf': 'EKF+EKS (d=8)', 'sgp': 'sgp_filter+sgp_smoother', 'harmonic': 'sgp_filter+sgp_smoother (cubature, d=8)',
             'cd_sgp': 'cd_sgp_filter+cd_sgp_smoother', 'cd_ekf': 'cd_ekf+cd_eks'}[k]

    def once(nn):
        t0 = time.perf_counter()
        a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys[:nn])
        if k in ('ekf', 'harmonic_ekf'):
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


WORKLOADS = {
    # kind: (label, default trials, default T, default scaling, which roof bounds it in the large-batch limit)
    'ekf': ("C2: discrete EKF+EKS (demos/ekfs_mle.py model)", 1000, 10000, 'weak', 'hbm'),
    'sgp': ("C3: Gauss-Hermite order-3 sgp_filter+sgp_smoother", 1000, 10000, 'strong', 'valu_f64'),
    'cd_sgp': ("C4: cd_sgp_filter+cd_sgp_smoother RK4", 512, 50000, 'weak', 'valu_f64'),
    'cd_ekf': ("cd_ekf+cd_eks RK4", 1000, 10000, 'weak', 'valu_f64'),
    'harmonic': ("C5: 3-harmonic chirp, cubature sgp_filter+sgp_smoother", 1000, 10000, 'strong', 'valu_f64'),
    'harmonic_ekf': ("3-harmonic chirp (d = 8), ekf+eks", 1000, 10000, 'weak', 'valu_f64'),
}
# float64 vector peak: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2 flop x 2.4 GHz (a wave64 v_fma_f64 occupies its SIMD for
# 4 cycles: measured 4.0 cycles per independent instruction, one wave per SIMD, tools/ubench/f64_issue.hip)
F64_VALU_PEAK_TFLOPS = 78.6
N_SIMDS = 1024


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='ekf', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None,
                    help='trials: per GPU under weak scaling, in total under strong scaling (default: BASELINE config)')
    ap.add_argument('--T', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--flags', type=int, default=0, help='CGP_* flag bits forwarded to the engine (e.g. 2 = wave per trial)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default=None,
                    help='weak: --batch trials per rank; strong: --batch trials in total, sharded (default per workload: '
                         'ekf / cd_* weak, sgp / harmonic strong as BASELINE.json words them)')
    ap.add_argument('--strong', action='store_true', help='same as --scaling strong')
    ap.add_argument('--rehearse', action='store_true',
                    help='multi-process dry run on fewer GPUs than ranks: ranks share devices and the collectives go over '
                         'gloo on host copies (RCCL refuses two ranks on one device); timings are then meaningless')
    return ap.parse_args(argv)


def launcher_command(gpus, argv, port):
    """The child process that runs this script on `gpus` ranks of one node (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(gpus)}',
            '--master-addr', '127.0.0.1', '--master-port', str(int(port)), os.path.abspath(__file__), *argv]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def valu_issue(workload, kernel, units, ms):
    """Executed float64 vector work of one launch from the committed PMC profile of this workload
    (profiles/r02_issue_table.json, written by tools/issue_table.py from rocprofv3 --pmc passes): wave-instructions by
    class per trial-step, hence executed FLOP (all 64 lanes counted, FMA = 2) and the share of VALU issue slots used."""
    try:
        tab = json.load(open(os.path.join(ROOT, 'profiles', 'r02_issue_table.json')))[workload][kernel]
    except (OSError, KeyError):
        return None
    flop_per_step = 64 * (2 * tab['fma_f64'] + tab['mul_f64'] + tab['add_f64']) + 2 * 256 * tab.get('mfma_f64', 0)
    return {"executed_tflops": flop_per_step * units / (ms * 1e-3) / 1e12, "valu_per_step": tab['valu'],
            "f64_per_step": tab['fma_f64'] + tab['mul_f64'] + tab['add_f64'], "cycles_per_step": tab.get('cycles'),
            "source": tab.get('source', 'profiles/r02_issue_table.json')}


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Start the ranks as a child process BEFORE anything here imports torch or touches the GPU; never exec.
        import subprocess
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        sys.exit(subprocess.run(launcher_command(args.gpus, sys.argv[1:], free_port()), env=env).returncode)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch it as '
                         f'`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` '
                         f'or let `python bench.py --gpus {args.gpus}` start the ranks itself')

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL; must be set before HIP initialises
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU')
    if args.rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    ranks_seen = dist.get_world_size() if world > 1 else 1

    def coll(t):
        """Tensor as the process group wants it: HBM for RCCL, a host copy for the gloo rehearsal."""
        return t.cpu() if args.rehearse else t

    from chirpgp_amd import filters_smoothers as fs
    from chirpgp_amd import _engine
    from chirpgp_amd.parallel import shard_bounds

    label, B_default, T_default, scaling_default, bound = WORKLOADS[args.workload]
    scaling = 'strong' if args.strong else (args.scaling or scaling_default)
    batch = args.batch or B_default
    T = args.T or T_default
    kw = dict(flags=args.flags) if args.flags else {}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(mode):
        """W warm-up and K timed passes of the workload under `mode` scaling -> measurements of this rank (times are the
        maximum over ranks)."""
        if mode == 'strong':                          # contiguous shard of the total batch
            lo, hi = shard_bounds(batch, rank, world)
            B, B_total = hi - lo, batch
        else:
            B, B_total = batch, batch * world
        wl = make_workload(max(B, 1), T, seed=1000003 * rank, kind=args.workload)
        if B == 0:                                    # more ranks than trials: this rank idles through the barriers
            wl['ys'] = wl['ys'][:0]
        ys_dev = torch.from_numpy(wl['ys']).cuda()

        def step():
            k = wl['kind']
            if k in ('ekf', 'harmonic_ekf'):
                f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.eks(wl['disc'], f[0], f[1], wl['dt'], **kw)
            elif k in ('sgp', 'harmonic'):
                f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'], **kw)
            elif k == 'cd_sgp':
                f = fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.cd_sgp_smoother(wl['drift'], wl['disp'](None), wl['sgps'], f[0], f[1], wl['dt'], **kw)
            else:
                f = fs.cd_ekf(wl['drift'], wl['disp'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.cd_eks(wl['drift'], wl['disp'], f[0], f[1], wl['dt'], **kw)
            return f, s

        # Allocator priming (untimed, before the W warm-up steps): the timed loop keeps one result alive while the next
        # step allocates its outputs, so it alternates between two sets of output buffers; two passes whose results are
        # held together make PyTorch's caching allocator own both sets, and no hipMalloc of GB-sized buffers lands in the
        # timed region whatever --warmup is.  Device memory management is not part of the path being measured.
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
            f, s = step()
        sync()
        elapsed = time.perf_counter() - t0
        _engine.kernel_events = None
        tmax = coll(torch.tensor([elapsed], dtype=torch.float64, device='cuda'))
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

        # the single collective of the path: gather the per-trial final NLL -- after the timed region
        gather_ms = None
        if world > 1:
            from chirpgp_amd import parallel
            last = coll(f[2][:, -1].contiguous())
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            out = parallel.all_gather_trials(last, B_total) if mode == 'strong' else parallel.all_gather_trials(last, B * world)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            assert out.shape[0] == B_total
        filt = [a.elapsed_time(b) for n, a, b in events if n == 'filter']
        smooth = [a.elapsed_time(b) for n, a, b in events if n == 'smoother']
        return dict(wl=wl, B=B, B_total=B_total, elapsed=elapsed, gather_ms=gather_ms,
                    filt_ms=float(np.mean(filt)) if filt else 0.0, smooth_ms=float(np.mean(smooth)) if smooth else 0.0)

    m = measure(scaling)
    # C2's north star also asks for the strong figure (1000 trials in total): measured right after, reported beside
    other = measure('strong') if (world > 1 and scaling == 'weak' and args.workload == 'ekf') else None

    if rank == 0:
        wl, B, B_total, elapsed = m['wl'], m['B'], m['B_total'], m['elapsed']
        d = wl['d']
        filt_ms, smooth_ms = m['filt_ms'], m['smooth_ms']
        bf, bs = bytes_per_trial_step(d)
        units = B * T                                   # rank 0's own share (kernel figures are rank 0's)
        dom = ('filter', filt_ms, bf) if filt_ms >= smooth_ms else ('smoother', smooth_ms, bs)
        achieved = dom[2] * units / (dom[1] * 1e-3) / 1e9
        total_units = B_total * T                       # all ranks
        total_gbs = (bf + bs) * total_units / (elapsed / args.steps) / 1e9
        bench_shape = args.workload == 'ekf' and B == 1000 and T == 10000 and not args.flags
        traffic = pmc_traffic('ekf4_mfma' if dom[0] == 'filter' else 'tp_smoother') if bench_shape else None
        hbm = {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
               "traffic_source": (PMC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                  "committed; not re-measured by this run)") if traffic is not None else None,
               "algorithmic_bytes_per_launch": dom[2] * units, "avg_launch_ms": dom[1],
               # one wavefront per trial below ~2.5 trials per SIMD (cgp_api.hip:choose_wave): occupancy of the 1024 SIMDs
               "waves_per_simd": round(B / N_SIMDS, 3) if B < 2560 else None}
        roofline = hbm
        if bound == 'valu_f64':
            # sigma-point / RK4 workloads: 80-220 flop per byte against a machine balance of ~10, so the
            # float64 vector pipe is the roof; the HBM fraction stays beside it
            iss = valu_issue(args.workload, dom[0], units, dom[1])
            roofline = {"bound": "valu_f64", "kernel": dom[0],
                        "achieved": iss["executed_tflops"] if iss else None, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": iss["executed_tflops"] / F64_VALU_PEAK_TFLOPS if iss else None,
                        "executed_flop_source": iss["source"] if iss else None,
                        "valu_per_step": iss["valu_per_step"] if iss else None, "f64_per_step": iss["f64_per_step"] if iss else None,
                        "cycles_per_step": iss["cycles_per_step"] if iss else None,
                        "waves_per_simd": hbm["waves_per_simd"], "avg_launch_ms": dom[1], "traffic": None, "traffic_source": None,
                        "hbm": {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}}
        result = {
            "metric": "filter+smoother trial-steps/s (batch x T / wall)",
            "value": total_units * args.steps / elapsed,
            "unit": "trial-steps/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "d": d, "T": T, "batch_per_gpu": B, "global_batch": B_total,
                       "parallelism": f"trials sharded x{world}",
                       "sigma_points": int(wl['sgps'].n_points) if args.workload not in ('ekf', 'cd_ekf', 'harmonic_ekf') else None},
            "hbm_gbs_total": total_gbs, "hbm_frac_of_peak_total": total_gbs / (HBM_PEAK_GBS * world),
            "roofline": roofline,
            "kernels": {"filter_ms": filt_ms, "smoother_ms": smooth_ms,
                        "filter_GBs": bf * units / (filt_ms * 1e-3) / 1e9 if filt_ms else None,
                        "smoother_GBs": bs * units / (smooth_ms * 1e-3) / 1e9 if smooth_ms else None},
            "gather_ms": m['gather_ms'],
        }
        if other is not None:
            result["strong"] = {"value": other['B_total'] * T * args.steps / other['elapsed'], "unit": "trial-steps/s",
                                "global_batch": other['B_total'], "batch_per_gpu": other['B'],
                                "ms_per_step": other['elapsed'] / args.steps * 1e3, "gather_ms": other['gather_ms'],
                                "filter_ms": other['filt_ms'], "smoother_ms": other['smooth_ms']}
        if bench_shape:
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
