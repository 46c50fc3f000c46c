"""bench.py on the GPU box: the multi-rank path rehearsed with two ranks sharing the one GPU (gloo for the collectives,
RCCL refuses two ranks on a device), and the contract line of a small single-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_two_rank_rehearsal_prints_one_line_for_both_ranks():
    r = _run('--gpus', '2', '--rehearse', '--batch', '8', '--T', '300', '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
    assert r['n_gpus'] == 2 and r['ranks_seen'] == 2 and r['scaling'] == 'weak'
    assert r['config']['global_batch'] == 16 and r['config']['batch_per_gpu'] == 8
    assert r['gather_ms'] is not None and r['value'] > 0
    assert r['strong']['global_batch'] == 8 and r['strong']['batch_per_gpu'] == 4 and r['strong']['gather_ms'] is not None


def test_strong_default_for_the_sigma_point_configs():
    r = _run('--gpus', '2', '--rehearse', '--workload', 'sgp', '--batch', '6', '--T', '200', '--steps', '1', '--warmup', '0', '--no-cpu-baseline')
    assert r['scaling'] == 'strong' and r['config']['global_batch'] == 6 and r['config']['batch_per_gpu'] == 3
    assert r['roofline']['bound'] == 'valu_f64' and r['roofline']['hbm']['unit'] == 'GB/s'


def test_single_gpu_line_carries_roofline_and_cpu_baseline():
    r = _run('--batch', '64', '--T', '500', '--steps', '2', '--warmup', '1')
    assert r['n_gpus'] == 1 and r['roofline']['bound'] == 'hbm' and r['roofline']['frac'] > 0
    assert 'traffic_source' in r['roofline']
    assert r['cpu_baseline']['kind'] == 'port' and r['cpu_baseline']['value'] > 0
