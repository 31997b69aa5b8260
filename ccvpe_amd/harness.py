"""Replica / data-parallel timing harness shared by bench.py and the gloo CPU test.

Inference shards by sample: every rank owns `batch_per_rank` independent image
pairs and runs the same step on them; there is NO data-path collective.  The only collectives
are the barrier that brackets the timed region and the MAX over ranks of the elapsed time.
`shard_indices` is the (deterministic) assignment of a global batch to ranks used by callers that
evaluate a fixed dataset.
"""
import time

import torch
import torch.distributed as dist


def shard_indices(n_items, world, rank):
    """Contiguous, balanced shard of range(n_items) for `rank` (first `n_items % world` ranks get
    one extra item).  Union over ranks == range(n_items), shards disjoint."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def timed_steps(step_fn, steps, warmup, sync_fn=None, device=None):
    """Run `warmup` untimed + exactly `steps` timed calls of step_fn, bracketed by
    barrier + device sync on both sides; returns the MAX elapsed seconds over ranks."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def fence():
        if distributed:
            dist.barrier()
        if sync_fn is not None:
            sync_fn()

    for _ in range(warmup):
        step_fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_throughput(units_per_rank_per_step, steps, elapsed_max):
    """Whole-job units/s: all ranks' units divided by the slowest rank's time."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return units_per_rank_per_step * world * steps / elapsed_max
