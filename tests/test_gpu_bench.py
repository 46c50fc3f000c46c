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
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), p.stdout           # exactly one line on stdout, and it is the JSON line
    assert len(lines[0]) < 4096, len(lines[0])                              # a 24.7 KB line was not parsed by the driver
    line = json.loads(lines[0], parse_constant=_no_constants)               # strict JSON: no NaN / Infinity tokens
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'details'):
        assert k in line, k
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(line['roofline'])
    assert 'algorithmic_frac' not in line['roofline']                       # a convention, kept in the details file only
    # the full record (every field the line had up to round 5) sits in the details file the line names
    with open(line['details']) as f:
        full = json.load(f, parse_constant=_no_constants)
    assert full['value'] == pytest.approx(line['value'], rel=1e-5) and full['steps'] == line['steps']
    return full


def _no_constants(tok):
    raise AssertionError(f'non-finite token {tok} in bench.py output')


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
    assert 'traffic_source' in r['roofline'] and r['roofline']['algorithmic_frac'] > 0
    cb = r['cpu_baseline']
    assert cb['sample'] and cb['unit'] == 'trial-steps/s'
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1 and cb['best_of'] == 3
    assert cb['one_core']['cores'] == 1 and 0 < cb['one_core']['value'] <= cb['value'] * 1.5
    assert '-march=native' in cb['build'] and '-DFIXED_D=4' in cb['build']
    assert 'other_configs' not in r                       # a non-default shape runs the named workload only


def test_world_one_rccl_path():
    """--force-dist: the never-otherwise-exercised RCCL branch (init_process_group('nccl', device_id), barrier, all_reduce(MAX),
    all_gather_into_tensor on DEVICE tensors, destroy) at world size 1 -- all the RCCL coverage a one-GPU box allows."""
    r = _run('--force-dist', '--batch', '32', '--T', '400', '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
    assert r['n_gpus'] == 1 and r['ranks_seen'] == 1 and r['collectives'] == 'rccl'
    assert r['gather_ms'] is not None and r['gather_ms'] >= 0 and r['value'] > 0
    k = r['kernels']
    assert k['filter_ms_max_over_ranks'] == pytest.approx(k['filter_ms']) and k['smoother_ms_max_over_ranks'] == pytest.approx(k['smoother_ms'])


def test_default_line_carries_the_other_baseline_configs():
    """The driver's command (default workload and sizes, fewer steps): C1, C3, C4, C5 ride on the same JSON line, each with its
    kernel times, throughput and both roofline fractions (executed and algorithmic) for the float64-bound ones."""
    r = _run('--steps', '3', '--warmup', '1', '--other-steps', '1', '--no-cpu-baseline')
    assert r['config']['batch_per_gpu'] == 1000 and r['config']['T'] == 10000
    # counter traffic is quoted only from a profile of the very library that ran (sha256 recorded in profiles/r05_ekf_pmc.json); any other
    # library gets null and the reason -- never a stale figure
    rf0 = r['roofline']
    assert rf0['traffic'] is not None or 'counters withheld' in rf0['traffic_source'], rf0
    if rf0['traffic'] is not None:
        assert 0.99 < rf0['traffic'] / rf0['algorithmic_bytes_per_launch'] < 1.05 and 'sha256' in rf0['traffic_source']
    oc = r['other_configs']
    assert set(oc) == {'C1', 'C2_low_freq', 'C3', 'C4', 'C5', 'CRLB_ekf', 'CRLB_ghf', 'C2_spread', 'time_split_filters'}
    # the regimes the headline filter ran its 64-step chunks in: counted by the kernel itself (cgp_debug_counters)
    rg = r['regimes']
    assert rg['chunks'] == 1000 * 157 and rg['high'] + rg['common'] + rg['low'] + rg['mid'] + rg['redone'] + rg['wide'] + rg['checked'] == rg['chunks']
    assert 0.7 < rg['high_share'] < 0.85 and rg['high_left'] < 0.05 * rg['chunks'] and rg['redone'] < 0.01 * rg['chunks']
    low = oc['C2_low_freq']
    assert (low['batch_per_gpu'], low['T'], low['d']) == (1000, 10000, 4) and low['filter_ms'] > 0 and low['smoother_ms'] > 0
    assert low['regimes']['high'] == 0 and low['regimes']['common'] > 0.98 * low['regimes']['chunks']
    crlb = oc['CRLB_ekf']
    assert (crlb['batch_per_gpu'], crlb['T']) == (262144, 500)
    for k in ('means_only', 'full_outputs'):
        assert crlb[k]['filter_ms'] > 0 and 0 < crlb[k]['hbm_frac'] < 1
        # counter traffic only from a profile of the very library that ran (else withheld, with the reason)
        assert (crlb[k]['traffic'] is None) == (crlb[k]['traffic_over_algorithmic'] is None) and crlb[k]['traffic_source']
        if crlb[k]['traffic'] is not None:
            assert 0.99 < crlb[k]['traffic_over_algorithmic'] < 1.16
    ghf = oc['CRLB_ghf']
    assert (ghf['batch_per_gpu'], ghf['T'], ghf['sigma_points']) == (262144, 500, 81) and ghf['means_only']['filter_ms'] > 0
    # the headline's dependence on its data, measured: 45 record sets of the headline shape
    sp = oc['C2_spread']
    assert sp['combinations'] == 45 and len(sp['rows']) == 45 and sp['value_min'] <= sp['value_median'] <= sp['value_max']
    # Bounded from both sides.  Round 5 first held "no record set more than 1.15 x the median pass" -- true at 1.05, with the median itself
    # at 0.63 of the headline.  With the LOW / MID / ANY regimes of the speculative step EVERY record set is faster (slowest pass 4.5 -> 3.9
    # ms) and the median sits at 0.84 of the headline; the ratio of the two moved to 1.2 because the median gained more than the slowest.
    # (1.18 - 1.21 measured, wall times of one to three passes per record set; the gate leaves room for a box's noise)
    assert sp['slowest_over_median_time'] < 1.35, sp['slowest']
    assert sp['value_median'] > 0.75 * r['value'], (sp['value_median'], r['value'])   # ... and the median within 25 % of the headline
    assert sp['redone_plus_checked_share_at_reference_inputs'] <= 0.01                # the reference's inputs, any seed: <= 1 % of the chunks repeated
    ts = oc['time_split_filters']
    for k in ('C2_shard', 'C3_shard', 'C4_per_gpu'):                      # the chirp filters forget: junctions at 1e-6 or better, >= 1.4 x
        assert ts[k]['accepted_at_1e-5'] and ts[k]['junction_mismatch'] < 1e-5 and ts[k]['worst_output_difference'] <= 5 * ts[k]['junction_mismatch']
        assert ts[k]['speedup'] > 1.4, (k, ts[k])
    assert not ts['C5_shard']['accepted_at_1e-5']                         # the 3-harmonic filter does not, and its junctions say so
    sm = ts['C4_per_gpu_smoother']                                        # round 6: the smoother's counterpart (cgp_smoother_time_split)
    assert sm['accepted_at_1e-5'] and sm['junction_mismatch'] < 1e-7 and sm['worst_output_difference'] <= max(1e-12, 5 * sm['junction_mismatch'])
    assert sm['speedup'] > 1.4, sm
    assert (oc['C1']['batch_per_gpu'], oc['C1']['T'], oc['C1']['d']) == (1, 1000, 4)
    assert (oc['C3']['batch_per_gpu'], oc['C3']['T']) == (1000, 10000) and oc['C3']['scaling'] == 'strong'
    assert (oc['C4']['batch_per_gpu'], oc['C4']['T']) == (512, 50000)
    assert (oc['C5']['batch_per_gpu'], oc['C5']['T'], oc['C5']['d']) == (1000, 10000, 8)
    for tag in ('C3', 'C4', 'C5'):
        rf = oc[tag]['roofline']
        assert rf['bound'] == 'valu_f64' and 0 < rf['algorithmic_frac'] < 1 and oc[tag]['value'] > 0
        assert oc[tag]['filter_ms'] > 0 and oc[tag]['smoother_ms'] > 0
        # (`frac` counts the executed f64 FMA / MUL / ADD / MFMA flop of the dominant kernel; `algorithmic_frac` prices SURVEY 8d's op
        # count of filter + smoother, a transcendental at 30 flop-equivalents -- the engine's lean polynomials execute fewer, so since
        # round 4 the algorithmic figure of C3 exceeds the executed one: the two are reported side by side, not ordered)
        if rf['frac'] is not None:
            assert 0 < rf['frac'] < 1
            # the two columns are labelled for what they are (`algorithmic_convention`) and may not drift apart silently: useful work
            # priced by the convention stays within 25 % above the executed figure (C3 today: 1.03 x; C4, C5 below 1)
            assert rf['algorithmic_frac'] <= 1.25 * rf['frac'] and 'convention' in rf['algorithmic_convention']


def test_three_rank_rehearsal_of_the_scaling_command():
    """First contact with a multi-GPU node should be uneventful: the driver's N > 1 command with the default shapes on THREE ranks sharing
    the one GPU -- the box admits six processes on its card and this test runner is one of them (a six-rank rehearsal was killed by the
    box's process guard: 8 processes on the GPU); world = 8 itself, with empty shards, is covered on CPU by tests/test_dist_gloo.py.  The
    parent spawns a fresh child (never re-execs), every rank goes through the headline, the strong C2 figure (ragged: 334 + 334 + 332),
    other_configs and the gathers, rank 0 alone times the host CPU once, and the whole run stays far inside the driver's 600 s."""
    import time
    t0 = time.time()
    r = _run('--gpus', '3', '--rehearse', '--steps', '2', '--warmup', '1', '--other-steps', '1')
    took = time.time() - t0
    assert r['n_gpus'] == 3 and r['ranks_seen'] == 3 and r['collectives'].startswith('gloo')
    assert r['scaling'] == 'weak' and r['config']['batch_per_gpu'] == 1000 and r['config']['global_batch'] == 3000
    assert r['strong']['global_batch'] == 1000 and r['strong']['batch_per_gpu'] == 334 and r['strong']['gather_ms'] is not None
    assert r['gather_ms'] is not None and r['gather_ms'] >= 0
    oc = r['other_configs']
    assert oc['C3']['batch_per_gpu'] == 334 and oc['C3']['global_batch'] == 1000 and oc['C5']['batch_per_gpu'] == 334
    assert oc['C4']['batch_per_gpu'] == 512 and oc['C4']['global_batch'] == 3 * 512 and oc['C1']['global_batch'] == 3
    assert oc['CRLB_ekf']['batch_per_gpu'] == 262144 // 3
    assert 'C2_spread' not in oc                                       # a single-GPU diagnostic: not repeated on every rank
    assert r['cpu_baseline']['value'] > 0 and r['cpu_baseline']['one_core']['value'] > 0
    print(f'three-rank rehearsal of the default line: {took:.0f} s')
    assert took < 400, took


def test_two_rank_rehearsal_of_the_default_line():
    """The driver's N > 1 command in rehearsal (two ranks on the one GPU, gloo collectives): the default shapes, so the ranks go
    through `other_configs` (C3 / C5 sharded 500 + 500, C4 and C1 per rank), the strong C2 figure, the final gathers, and rank 0
    times the host CPU AFTER the process group is gone -- the control flow a multi-GPU node will run, which no other test reaches."""
    r = _run('--gpus', '2', '--rehearse', '--steps', '2', '--warmup', '1', '--other-steps', '1')
    assert r['n_gpus'] == 2 and r['ranks_seen'] == 2 and r['collectives'].startswith('gloo')
    assert r['config']['batch_per_gpu'] == 1000 and r['config']['global_batch'] == 2000 and r['strong']['batch_per_gpu'] == 500
    oc = r['other_configs']
    assert oc['C3']['batch_per_gpu'] == 500 and oc['C3']['global_batch'] == 1000 and oc['C5']['batch_per_gpu'] == 500
    assert oc['C4']['batch_per_gpu'] == 512 and oc['C4']['global_batch'] == 1024 and oc['C1']['global_batch'] == 2
    assert all(v['value'] > 0 and v['filter_ms'] > 0 for k, v in oc.items() if k not in ('CRLB_ekf', 'CRLB_ghf', 'time_split_filters'))
    assert oc['CRLB_ekf']['full_outputs']['filter_ms'] > 0
    assert r['cpu_baseline']['value'] > 0 and r['cpu_baseline']['one_core']['value'] > 0
