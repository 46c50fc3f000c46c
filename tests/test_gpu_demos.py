"""Smoke tests of the counterpart driver scripts (demos/): the reference's five-step pipeline MLE -> filter -> smoother ->
E[g(V)] -> RMSE on the engine for the sigma-point, continuous-discrete and harmonic configurations (BASELINE C3, C4, C5),
and the CRLB job with the Gauss-Hermite filter.  Short records; asserts a finite RMSE and that the MLE lowered the NLL."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'demos'))


def _check(rows, tmp_path, method):
    from chirpgp_amd import results
    assert len(rows) == 1
    name, err, nll0, nll1 = rows[0]
    assert np.isfinite(err) and err < 20.0, rows
    assert nll1 < nll0 - 1.0, rows
    z = np.load(results.result_path(str(tmp_path), method, name, 0))
    assert set(z.files) == {'smoothing_mean', 'smoothing_cov', 'rmse'} and np.isfinite(z['smoothing_mean']).all()


def test_ghfs_mle_pipeline(tmp_path):
    """demos/ghfs_mle.py:34-90 (config C3's driver)."""
    import _pipeline
    from chirpgp_amd.quadratures import SigmaPoints
    rows = _pipeline.demo('ghfs', sgps=SigmaPoints.gauss_hermite(d=4, order=3), T=1200, maxiter=40, save_dir=str(tmp_path), mags=('const',), quiet=True)
    _check(rows, tmp_path, 'ghfs')


def test_cd_ghfs_mle_pipeline(tmp_path):
    """demos/cd_ghfs_mle.py:28-80 (config C4's driver)."""
    import _pipeline
    from chirpgp_amd.quadratures import SigmaPoints
    rows = _pipeline.demo('cd_ghfs', sgps=SigmaPoints.gauss_hermite(d=4, order=3), T=600, maxiter=30, save_dir=str(tmp_path), mags=('const',), quiet=True)
    _check(rows, tmp_path, 'cd_ghfs')


def test_ghfs_harmonics_mle_pipeline(tmp_path):
    """demos/ghfs_harmonics_mle.py:27-80 (config C5's driver): three harmonics, d = 8, cubature."""
    import _pipeline
    from chirpgp_amd.quadratures import SigmaPoints
    rows = _pipeline.demo('ghfs', sgps=SigmaPoints.cubature(d=8), num_harmonics=3, T=1200, seed=777, maxiter=40, save_dir=str(tmp_path),
                          mags=('const',), quiet=True)
    _check(rows, tmp_path, 'ghfs')


def test_crlb_ghf_job(tmp_path):
    """tetralith/jobs/crlb_ghf.py:64-95 at a reduced number of trials: error statistics finite, filter error below the prior's."""
    import crlb_ekf
    out = str(tmp_path / 'crlb_ghf.npz')
    stats = crlb_ekf.main(['--filter', 'ghf', '--num-mcs', '4000', '--T', '100', '--chunk', '2000', '--save', out])
    mean_c, std_c, mean_v, std_v = (x.cpu().numpy() for x in stats)
    assert np.isfinite(mean_c).all() and np.isfinite(std_v).all() and (mean_c > 0).all()
    z = np.load(out)
    assert set(z.files) == {'ts', 'err_mean_chirps', 'err_std_chirps', 'err_mean_vs', 'err_std_vs'} and z['ts'].shape == (100,)
    # the EKF job on the same draws gives errors of the same size (the two filters agree closely on this mildly nonlinear model)
    ekf = crlb_ekf.main(['--filter', 'ekf', '--num-mcs', '4000', '--T', '100', '--chunk', '2000'])
    assert np.allclose(mean_c, ekf[0].cpu().numpy(), rtol=0.2)
