"""Counterpart of the reference's demos/cd_ghfs_mle.py (BASELINE config C4's driver): continuous-discrete Gauss-Hermite
filter and smoother (RK4 on the sigma-point moment ODEs) on the chirp SDE, parameters by MLE through the filter.

    python demos/cd_ghfs_mle.py [--T 3141] [--seed 555] [--save DIR]
"""
import argparse

from _pipeline import demo
from chirpgp_amd.quadratures import SigmaPoints

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--seed', type=int, default=555)
    ap.add_argument('--save', default=None)
    a = ap.parse_args()
    demo('cd_ghfs', sgps=SigmaPoints.gauss_hermite(d=4, order=3), T=a.T, seed=a.seed, save_dir=a.save)
