"""Counterpart of the reference's demos/ghfs_harmonics_mle.py (BASELINE config C5's driver): three harmonics (d = 8),
cubature rule (16 points), sigma-point filter and smoother, parameters by MLE through the filter.

    python demos/ghfs_harmonics_mle.py [--T 3141] [--seed 777] [--harmonics 3] [--save DIR]
"""
import argparse

from _pipeline import demo
from chirpgp_amd.quadratures import SigmaPoints

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--seed', type=int, default=777)
    ap.add_argument('--harmonics', type=int, default=3)
    ap.add_argument('--save', default=None)
    a = ap.parse_args()
    demo('ghfs', sgps=SigmaPoints.cubature(d=2 * a.harmonics + 2), num_harmonics=a.harmonics, T=a.T, seed=a.seed, save_dir=a.save)
