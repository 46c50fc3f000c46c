"""Counterparts of the reference's Monte-Carlo jobs tetralith/jobs/*_mle.py on the MI355X engine: for every Monte-Carlo run
and every magnitude law (constant, damped, Ornstein-Uhlenbeck) MLE of the model parameters through the filter -> filter ->
smoother -> E[g(V)] -> RMSE, saved as ``<results>/<job>_<mag>_<mc>.npz`` (smoothing_mean, smoothing_cov, rmse; NaN for a
diverged run) -- the files paper_plots_tables/print_rmse_table.py:40-50 reads.

    python demos/jobs.py <job> [--num-mcs 100] [--T 3141] [--results ./results] [--maxiter 200] [--lockstep]

job (reference file, lines of its objective):
    ekfs_mle            tetralith/jobs/ekfs_mle.py:40-43            chirp model, ekf + eks
    ghfs_mle            tetralith/jobs/ghfs_mle.py:43-46            chirp model, Gauss-Hermite order 3
    cd_ekfs_mle         tetralith/jobs/cd_ekfs_mle.py               chirp SDE, cd_ekf + cd_eks
    cd_ghfs_mle         tetralith/jobs/cd_ghfs_mle.py:47-50         chirp SDE, cd_sgp_filter + cd_sgp_smoother
    lascala_ekfs_mle    tetralith/jobs/lascala_ekfs_mle.py:40-43    La Scala model (4 parameters), ekf + eks
    lascala_ghfs_mle    tetralith/jobs/lascala_ghfs_mle.py:43-46    La Scala model, Gauss-Hermite order 3
    harmonic_ekfs_mle   tetralith/jobs/harmonic_ekfs_mle.py:43-46   3-harmonic chirp model (d = 8), ekf + eks, 3-harmonic signal
    harmonic_ckfs_mle   tetralith/jobs/harmonic_ckfs_mle.py         3-harmonic chirp model, cubature filter + smoother
    kpt_mle             tetralith/jobs/kpt_mle.py:41-44             KPT model, ekf_for_kpt + rts, 1 harmonic
    harmonic_kpt_mle    tetralith/jobs/harmonic_kpt_mle.py:44-47    KPT model with 3 harmonics, 3-harmonic signal

The reference seeds run `mc` from tetralith/rnd_keys.npy (jax.random keys, not reproducible without JAX); here run `mc`
uses numpy.random.default_rng(seed + mc).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from _pipeline import demo, records_of_run, run_records       # noqa: E402
from chirpgp_amd import results as res_io                     # noqa: E402
from chirpgp_amd.quadratures import SigmaPoints               # noqa: E402
from chirpgp_amd.toymodels import meow_freq                   # noqa: E402
from chirpgp_amd.tools import rmse                            # noqa: E402

GH4 = lambda: SigmaPoints.gauss_hermite(d=4, order=3)         # noqa: E731
# job -> (method, family, model harmonics, signal harmonics, sigma points)
JOBS = {
    'ekfs_mle': ('ekfs', 'chirp', 0, 0, None),
    'ghfs_mle': ('ghfs', 'chirp', 0, 0, GH4),
    'cd_ekfs_mle': ('cd_ekfs', 'chirp', 0, 0, None),
    'cd_ghfs_mle': ('cd_ghfs', 'chirp', 0, 0, GH4),
    'lascala_ekfs_mle': ('ekfs', 'lascala', 0, 0, None),
    'lascala_ghfs_mle': ('ghfs', 'lascala', 0, 0, GH4),
    'harmonic_ekfs_mle': ('ekfs', 'harmonic', 3, 3, None),
    'harmonic_ckfs_mle': ('ghfs', 'harmonic', 3, 3, lambda: SigmaPoints.cubature(d=8)),
    'kpt_mle': ('kpt', 'kpt', 1, 0, None),
    'harmonic_kpt_mle': ('kpt', 'kpt', 3, 3, None),
}


def run_job(job, num_mcs=100, T=3141, results=None, maxiter=200, seed=0, mags=None, quiet=False, lockstep=False):
    """The job's Monte-Carlo loop; result files go to `results` (None: not saved).  `lockstep`: all runs of a magnitude law
    at once -- one lock-step maximum-likelihood fit, one batched filter and one batched smoother launch (_pipeline.run_records).
    -> [(mc, mag, rmse, nll at the start, nll at the optimum), ...]"""
    method, family, model_h, signal_h, sg = JOBS[job]
    rows = []
    if lockstep:
        dt, Xi = 0.001, 0.1
        truth = meow_freq(offset=8.)[0](np.linspace(dt, dt * T, T))
        per_mag = {}
        for mc in range(num_mcs):
            for k, name, ys in records_of_run(seed + mc, T, dt, Xi, signal_h, mags):
                per_mag.setdefault(name, []).append(ys)
        for name, recs in per_mag.items():
            t0 = time.time()
            r = run_records(method, np.stack(recs), Xi, dt, sgps=sg() if sg else None, num_harmonics=model_h, maxiter=maxiter, family=family)
            for mc in range(num_mcs):
                ok = np.isfinite(r['fun'][mc])
                err = float(rmse(truth, r['est_freq'][mc])) if ok else float('nan')
                if results:       # tetralith/jobs/ekfs_mle.py:75-81: NaN results for a diverged run
                    res_io.save_result(results, job, name, mc, r['mss'][mc] if ok else np.nan, r['Pss'][mc] if ok else np.nan, err)
                rows.append((mc, name, err, float(r['nll0'][mc]), float(r['fun'][mc])))
            if not quiet:
                errs = np.array([x[2] for x in rows if x[1] == name])
                print(f'{job} {name:7s} {num_mcs} runs in lock step: {r["launches"]} launches, {int(np.mean(r["nit"]))} iterations on average, '
                      f'RMSE {np.nanmean(errs):.3f} +- {np.nanstd(errs):.3f} Hz  [{time.time() - t0:.2f} s]')
        return sorted(rows)
    for mc in range(num_mcs):
        out = demo(method, sgps=sg() if sg else None, num_harmonics=model_h, signal_harmonics=signal_h, family=family, T=T,
                   seed=seed + mc, maxiter=maxiter, save_dir=results, result_name=job, mc=mc, mags=mags, quiet=quiet)
        for name, err, nll0, nll1 in out:
            rows.append((mc, name, err, nll0, nll1))
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('job', choices=sorted(JOBS))
    ap.add_argument('--num-mcs', type=int, default=100)
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--results', default='./results')
    ap.add_argument('--maxiter', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--lockstep', action='store_true', help='all Monte-Carlo runs of a magnitude law at once (lock-step MLE, batched filter / smoother)')
    a = ap.parse_args(argv)
    return run_job(a.job, num_mcs=a.num_mcs, T=a.T, results=a.results, maxiter=a.maxiter, seed=a.seed, lockstep=a.lockstep)


if __name__ == '__main__':
    main()
