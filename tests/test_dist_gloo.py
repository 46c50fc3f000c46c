"""The N > 1 path on CPU: world_size = 2 and 8 (the driver's scaling run) over gloo.  Sharding and the final gather are the product's
(chirpgp_amd/parallel.py); the per-shard compute here is the oracle's C port standing in for the GPU kernels, so the
test checks that sharded-then-gathered results equal the single-process result bit for bit, including a ragged
last shard."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port_no, B, T, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port_no))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from chirpgp_amd import models as pm
        from chirpgp_amd import parallel as par
        from oracle import port
        from tests.refcases import chirp_measurements
        params = np.array([0.1, 0.1, 0.1, 1., 1., 7.]) * (1 + 0.05 * np.arange(B))[:, None]
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        ys = np.stack([chirp_measurements(T, 300 + i)[2] for i in range(B)])
        lo, hi = par.shard_bounds(B, rank, world)
        disc_l = pm.disc_chirp_lcd(*[params[lo:hi, i] for i in (0, 1, 3, 4)])
        if hi > lo:
            nll_local = port.filter(port.F_EKF, disc_l, None, H, 0.1, par.shard(m0, rank, world, B), par.shard(P0, rank, world, B),
                                    1e-3, par.shard(ys, rank, world, B), nll_final_only=True)[2]
        else:
            nll_local = np.empty((0,))                     # more ranks than blocks: this rank's shard is empty, the gather still runs
        full = par.all_gather_trials(torch.from_numpy(nll_local), B)
        if rank == 0:
            want = port.filter(port.F_EKF, disc, None, H, 0.1, m0, P0, 1e-3, ys, nll_final_only=True)[2]
            ret['ok'] = bool(np.array_equal(full.numpy(), want))
            ret['n'] = int(full.shape[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,B', [(2, 8), (2, 7), (8, 24), (8, 17)])
def test_sharded_sweep_matches_single_process(world, B):
    """(8, 17): blocks of ceil(17 / 8) = 3 -- five full shards, one of two trials, two EMPTY ranks that still join the collective."""
    T = 60
    with mp.Manager() as mgr:
        ret = mgr.dict()
        port_no = 29500 + (os.getpid() % 2000) + B + 40 * world
        mp.spawn(_worker, args=(world, port_no, B, T, ret), nprocs=world, join=True)
        assert ret.get('ok') is True and ret.get('n') == B


def test_shard_bounds_cover_the_batch():
    from chirpgp_amd import parallel as par
    for B in (0, 1, 7, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [par.shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= -(-B // world) if B else hi == lo for lo, hi in spans)
