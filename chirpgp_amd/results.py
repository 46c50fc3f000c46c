"""Result files in the layout of the reference's Monte-Carlo drivers (tetralith/jobs/ekfs_mle.py:80-81), so that
consumers written for those files (paper_plots_tables/print_rmse_table.py:40-50) keep working:
``<dir>/<method>_<mag>_<mc>.npz`` with arrays ``smoothing_mean``, ``smoothing_cov``, ``rmse``; a diverged run stores
NaN (tetralith/jobs/ekfs_mle.py:75-78)."""
import os
import numpy as np

__all__ = ['result_path', 'save_result', 'load_rmse_table']


def result_path(directory, method, mag, mc):
    return os.path.join(directory, f'{method}_{mag}_{mc}.npz')


def save_result(directory, method, mag, mc, smoothing_mean, smoothing_cov, rmse):
    os.makedirs(directory, exist_ok=True)
    np.savez(result_path(directory, method, mag, mc), smoothing_mean=np.asarray(smoothing_mean),
             smoothing_cov=np.asarray(smoothing_cov), rmse=np.asarray(rmse))


def load_rmse_table(directory, method, mags, num_mcs):
    """-> {mag: (mean, std, number of NaN runs)} like print_rmse_table.py:40-50."""
    table = {}
    for mag in mags:
        vals = np.array([float(np.load(result_path(directory, method, mag, mc))['rmse']) for mc in range(num_mcs)])
        ok = np.isfinite(vals)
        table[mag] = (float(np.mean(vals[ok])) if ok.any() else np.nan, float(np.std(vals[ok])) if ok.any() else np.nan, int((~ok).sum()))
    return table
