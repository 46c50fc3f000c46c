"""Counterpart of the reference's tetralith/jobs/crlb_ghf.py: the CRLB job with the Gauss-Hermite (order 3) sigma-point filter
instead of the EKF -- 10^6 simulated chirp-SDE trajectories filtered in batched launches, per-step error statistics out.

    python demos/crlb_ghf.py [-lam 0.1 -b 0.1 -delta 0.1 -ell 1 -sigma 1 -Xi 0.1] [--num-mcs 1000000] [--T 500] [--chunk 250000]
"""
import sys

from crlb_ekf import main

if __name__ == '__main__':
    main(sys.argv[1:] + ['--filter', 'ghf'])
