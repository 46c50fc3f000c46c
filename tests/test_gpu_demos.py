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


def test_mc_mle_sharded_over_two_ranks_equals_one_process(tmp_path):
    """demos/mc_mle.py under torchrun (two ranks rehearsed on the one GPU, gloo): records sharded in contiguous blocks, no
    exchange during the run, one all_gather of the per-record RMSEs -- and the same numbers as the one-process run, because a
    record's noise depends on its global number only (counter-based generator) and its MLE only on the record."""
    import subprocess
    script = os.path.join(ROOT, 'demos', 'mc_mle.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    common = ['--num-mcs', '7', '--T', '800', '--compare', '0']
    one = subprocess.run([sys.executable, script, *common], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), script, *common, '--rehearse'], env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    line1 = [l for l in one.stdout.splitlines() if 'RMSE of the frequency estimate' in l][0]
    line2 = [l for l in two.stdout.splitlines() if 'over all shards' in l][0]
    import re
    m1 = re.search(r'estimate: ([0-9.]+) \+- ([0-9.]+)', line1)
    m2 = re.search(r'RMSE ([0-9.]+) \+- ([0-9.]+)', line2)
    assert m1 and m2, (line1, line2)
    assert m1.group(1) == m2.group(1) and m1.group(2) == m2.group(2), (line1, line2)


def test_print_time_and_bats_shape_scripts():
    """Counterparts of paper_plots_tables/print_time.py and of the bat-call analyses' timed section
    (real_applications/bats/myotis_myotis_analysis.py:76-85), shortened: they run, time a pass and, for the bat shape,
    follow the sweep."""
    import print_time
    import bats_shape
    assert 0.0 < print_time.main(['--T', '1200', '--maxiter', '30']) < 5.0
    elapsed, err = bats_shape.main(['--T', '3000'])
    assert 0.0 < elapsed < 5.0 and err < 1500.0


def test_lorenz_demo_on_runtime_compiled_models(capsys):
    """demos/lorenz_custom.py: the four filter / smoother pairs on Lorenz-63 handed over as source; smoothed states near the simulated ones."""
    import lorenz_custom
    old = sys.argv
    sys.argv = ['lorenz_custom.py', '--T', '500', '--batch', '4']
    try:
        lorenz_custom.main()
    finally:
        sys.argv = old
    out = capsys.readouterr().out
    assert out.count('RMSE of the smoothed state') == 4 and 'nan' not in out


def test_ekfs_demo_with_exact_gradients(capsys):
    import ekfs_mle
    old = sys.argv
    sys.argv = ['ekfs_mle.py', '--T', '800', '--exact']
    try:
        ekfs_mle.main()
    finally:
        sys.argv = old
    out = capsys.readouterr().out
    assert out.count('RMSE') == 3 and 'nan' not in out
