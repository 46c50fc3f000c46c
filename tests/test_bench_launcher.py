"""bench.py --gpus N must start N ranks itself (the flag used to be parsed and ignored) -- CPU-side checks of
the launcher: the child command, the exit-code hand-through, the WORLD_SIZE / --gpus consistency check, and that the
parent never imports torch."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return env


def test_child_command_is_one_rank_per_gpu_on_loopback():
    import bench
    cmd = bench.launcher_command(8, ['--gpus', '8', '--steps', '3', '--workload', 'sgp'], 29555)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29555'
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '3', '--workload', 'sgp']      # the ranks see the caller's arguments
    assert 0 < bench.free_port() < 65536


def test_defaults_follow_the_baseline_wording():
    import bench
    assert bench.WORKLOADS['ekf'][3] == 'weak' and bench.WORKLOADS['sgp'][3] == 'strong' and bench.WORKLOADS['harmonic'][3] == 'strong'
    a = bench.parse_args([])
    assert a.gpus == 1 and a.workload == 'ekf' and not a.strong and a.scaling is None


def test_parent_spawns_and_hands_the_exit_code_through(monkeypatch):
    """--gpus 2 without WORLD_SIZE: main() runs the launcher command as a child and exits with its code, without importing torch."""
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '1'])
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    assert '--nproc-per-node=2' in seen['cmd'] and seen['cmd'][-4:] == ['--gpus', '2', '--steps', '1']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_world_size_mismatch_fails_loudly(monkeypatch):
    import bench
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4'])
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=2' in str(e.value.code)


def test_real_launch_reaches_both_ranks():
    """End to end in this container: the parent starts two real ranks through torch.distributed.run; with no GPU here each
    rank stops at bench.py's own 'needs a GPU' and the parent reports the failure (it must not print a 1-rank line)."""
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip('GPU box: covered by the -m gpu rehearsal test')
    assert p.returncode != 0
    # the launcher terminates the surviving rank as soon as the first one fails, so the message may be seen once;
    # its failure report names both ranks
    assert p.stderr.count('bench.py needs a GPU') >= 1, p.stderr[-2000:]
    assert 'local_rank: 0' in p.stderr and 'local_rank: 1' in p.stderr, p.stderr[-2000:]
    assert '"n_gpus"' not in p.stdout


def test_contract_line_is_compact_strict_json():
    """the driver could not parse a 24.7 KB line.  The line is built by bench.contract_line from the full record; fed round 5's
    committed record (profiles/r05_bench_c2.json: 45 spread rows, every other config) plus non-finite values and absurdly long strings it
    must stay one strict-JSON line under 4 KB that still carries the contract's fields, `roofline` and `cpu_baseline`."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_c2.json')))
    assert len(json.dumps(full)) > 20000
    full['gather_ms'] = float('nan')
    full['roofline']['traffic_source'] = 'x' * 5000
    full['cpu_baseline']['sample'] = 'y' * 5000
    full['other_configs']['C4']['value'] = float('inf')
    text = bench.contract_line(full, '/somewhere/bench_details.json')
    assert '\n' not in text and len(text) < bench.LINE_LIMIT == 4096

    def no_constants(tok):
        raise AssertionError(tok)
    line = json.loads(text, parse_constant=no_constants)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'ranks_seen', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'kernels', 'roofline', 'cpu_baseline', 'gpu_over_cpu', 'regimes', 'details'):
        assert k in line, k
    assert line['gather_ms'] is None and line['other_configs']['C4']['value'] is None
    rf = line['roofline']
    assert rf['bound'] == 'hbm' and rf['frac'] == pytest.approx(full['roofline']['frac'], rel=1e-5) and rf['peak'] == 8000.0
    assert rf['traffic'] == pytest.approx(full['roofline']['traffic'], rel=1e-5) and 'algorithmic_frac' not in rf
    cb = line['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] == full['cpu_baseline']['cores'] and cb['value'] > 0 and len(cb['sample']) <= 160
    assert line['config']['workload'].startswith('C2') and 'model' not in line['config']
    # a pathological record still yields a line: the last-resort form
    full['config']['workload'] = 'C2 ' + 'z' * 9000
    text = bench.contract_line(full, '/somewhere/bench_details.json')
    line = json.loads(text, parse_constant=no_constants)
    assert len(text) < 4096 and line['value'] == pytest.approx(full['value'], rel=1e-5) and line['roofline']['frac'] > 0 and line['cpu_baseline']['cores'] > 0
    # the details file is strict JSON too
    import tempfile
    assert json.loads(json.dumps(bench._finite(full), allow_nan=False))['steps'] == full['steps']
