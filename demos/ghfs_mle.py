"""Counterpart of the reference's demos/ghfs_mle.py (BASELINE config C3's driver): Gauss-Hermite (order 3, 81 points)
sigma-point filter and smoother on the chirp model, parameters by MLE through the filter.

    python demos/ghfs_mle.py [--T 3141] [--seed 555] [--save DIR]
"""
import argparse

from _pipeline import demo
from chirpgp_amd.quadratures import SigmaPoints

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=3141)
    ap.add_argument('--seed', type=int, default=555)
    ap.add_argument('--save', default=None, help='directory for <method>_<mag>_<mc>.npz result files')
    a = ap.parse_args()
    demo('ghfs', sgps=SigmaPoints.gauss_hermite(d=4, order=3), T=a.T, seed=a.seed, save_dir=a.save)
