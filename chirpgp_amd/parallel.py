"""Multi-GPU sharding of the trial axis.

Trials (and parameter-grid points) are independent: rank r of W takes the contiguous block
[r * ceil(B / W), min(B, (r + 1) * ceil(B / W))) of the batch axis, runs filter and smoother locally with no exchange,
and the only collective of the path is one all_gather at the end -- of whatever the caller wants on every rank
(per-trial final NLLs for an MLE sweep: B doubles; per-step error sums for a CRLB job: T doubles).  Trajectories stay
sharded by default (gathering them costs as much as computing them: SURVEY.md 8e).  One process per GPU,
torch.distributed with backend "nccl" (RCCL over xGMI) on GPUs; the helpers are backend-agnostic so the same code runs
under "gloo" on CPU tensors in the tests.
"""
import math

__all__ = ['shard_bounds', 'shard', 'all_gather_trials']


def shard_bounds(B, rank, world):
    """Contiguous block of the batch axis owned by `rank` (blocks differ by at most ceil(B / world) - the tail)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f'bad rank {rank} / world {world}')
    per = math.ceil(B / world) if B else 0
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def shard(x, rank, world, B=None):
    """This rank's block of a batched operand; operands without the batch axis (shared model, H, m0 ...) pass through."""
    if B is None:
        B = x.shape[0]
    if getattr(x, 'ndim', 0) == 0 or x.shape[0] != B:
        return x
    lo, hi = shard_bounds(B, rank, world)
    return x[lo:hi]


def all_gather_trials(local, B, group=None):
    """all_gather of a per-trial tensor whose leading axis is this rank's shard -> the full (B, ...) tensor on every rank.
    Ragged last shards are padded to ceil(B / world) rows for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = math.ceil(B / world) if B else 0
    pad = per - local.shape[0]
    if pad:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))])
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:B]
