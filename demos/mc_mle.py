"""Monte-Carlo study in the style of the reference's tetralith/jobs/ekfs_mle.py -- MLE, EKF, EKS, E[g(V)], RMSE for many
noisy realisations of the same chirp -- with the MLE of ALL records run in lock step (chirpgp_amd.mle.fit_many: every
line-search probe of every record in one kernel launch) and the filtering / smoothing of all records in one launch each.

    python demos/mc_mle.py [--num-mcs 100] [--T 3141] [--compare 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 demos/mc_mle.py --num-mcs 800

--compare N also fits the first N records one at a time with SciPy's L-BFGS-B (the reference's optimiser) for timing.
Under torchrun the records shard over the ranks (one process per GPU, contiguous blocks: chirpgp_amd.parallel), every rank runs
its shard with no exchange, and ONE all_gather at the end collects the per-record RMSEs (RCCL; `--rehearse`: several ranks on
one GPU with gloo) -- the reference's counterpart is one OS process per job (tetralith/run_local.sh:15-25).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chirpgp_amd import filters_smoothers as fs, mle, toymodels                       # noqa: E402
from chirpgp_amd.models import g, build_chirp_model                                   # noqa: E402
from chirpgp_amd.quadratures import gaussian_expectation                              # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--num-mcs', type=int, default=100)
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--compare', type=int, default=3)
    ap.add_argument('--seed', type=int, default=666)
    ap.add_argument('--rehearse', action='store_true', help='ranks share GPUs, collectives over gloo (tests on a one-GPU box)')
    args = ap.parse_args(argv)

    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        from chirpgp_amd import parallel
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local % torch.cuda.device_count() if args.rehearse else local)
        if args.rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    R_total = args.num_mcs
    lo, hi = (0, R_total) if world == 1 else parallel.shard_bounds(R_total, rank, world)

    dt, T, Xi, R = 1e-3, args.T, 0.1, hi - lo
    ts = np.linspace(dt, dt * T, T)
    freq, phase = toymodels.meow_freq(offset=8.)
    clean = toymodels.gen_chirp(ts, toymodels.constant_mag(1.), phase)
    yss = toymodels.noisy_copies(clean, Xi, args.seed, R, trial0=lo)       # (R, T) drawn on the device: record i is the same on any sharding
    yss_h = yss.cpu().numpy()
    init = [0.1, 0.1, 0.1, 1., 1., 7.]
    mle.fit_many('ekf', build_chirp_model, init, yss_h[:2], Xi, dt, maxiter=2)          # warm-up (library load, allocations)

    torch.cuda.synchronize(); t0 = time.time()
    params, info = mle.fit_many('ekf', build_chirp_model, init, yss_h, Xi, dt)
    torch.cuda.synchronize(); t1 = time.time()
    drift, disp, m_and_cov, m0, P0, H = build_chirp_model(params)          # one parameter vector per record
    mfs, Pfs, _ = fs.ekf(m_and_cov, H, Xi, m0, P0, dt, yss)
    mss, Pss = fs.eks(m_and_cov, mfs, Pfs, dt)
    est = gaussian_expectation(ms=mss[:, :, 2], chol_Ps=torch.sqrt(Pss[:, :, 2, 2]), func=g, force_shape=True)
    est = est.reshape(R, T)
    rmses = torch.sqrt(torch.mean((est - torch.from_numpy(freq(ts)).cuda()) ** 2, dim=1)).cpu().numpy()
    torch.cuda.synchronize(); t2 = time.time()
    if world > 1:                                          # the single collective: every rank ends up with all RMSEs
        mine = torch.from_numpy(rmses).cuda() if not args.rehearse else torch.from_numpy(rmses)
        rmses_all = parallel.all_gather_trials(mine, R_total).cpu().numpy()
        dist.barrier()
        if rank == 0:
            print(f'{world} ranks, {R_total} records: RMSE {np.nanmean(rmses_all):.3f} +- {np.nanstd(rmses_all):.3f} Hz over all shards')
        dist.destroy_process_group()
        if rank != 0:
            return rmses_all
        rmses = rmses_all
    print(f'{R} records x {T} steps: lock-step MLE {t1 - t0:.2f} s ({info["launches"]} launches, '
          f'{int(info["nit"].mean())} iterations on average), EKF + EKS + E[g(V)] + RMSE {t2 - t1:.3f} s')
    print(f'  RMSE of the frequency estimate: {np.nanmean(rmses):.3f} +- {np.nanstd(rmses):.3f} Hz   '
          f'(NaN results: {int(np.isnan(rmses).sum())})')
    if args.compare and world == 1:
        t3 = time.time()
        worse = 0.0
        for r in range(min(args.compare, R)):
            _, res = mle.fit('ekf', build_chirp_model, init, yss_h[r], Xi, dt)
            worse = max(worse, (info['fun'][r] - res.fun) / abs(res.fun))
        per = (time.time() - t3) / min(args.compare, R)
        print(f'  one record at a time (SciPy L-BFGS-B, same objective): {per:.3f} s per record -> {per * R:.1f} s for {R}; '
              f'lock-step optimum worse by at most {worse:.1e} (relative NLL)')

    return rmses


if __name__ == '__main__':
    main()
