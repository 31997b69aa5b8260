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


class GradientAllReducer:
    """Data-parallel training (BASELINE C3): average the parameter gradients over the ranks.

    One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI.  Gradients are packed into a few
    LARGE flat buckets (default 64 MiB: the ~58 M-parameter model is 4 buckets) because xGMI is point-to-point and
    a ring all-reduce is bound per link — few, large messages amortise the per-collective latency; the buckets are
    issued asynchronously back to back and unpacked after the last one completes.  The bucket layout is fixed at
    construction (parameter order), so every rank reduces the same bytes in the same order."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            nbytes = p.numel() * 4
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = None

    # ---- overlapped mode: reduce inside the model's backward --------------------------------------------
    def attach(self, model):
        """Start the all-reduce of each gradient group as soon as the model's backward has produced it
        (ccvpe_amd/train.py calls ready() after the decoders, after the aerial encoder and at the end): the
        decoder group (~80 % of the bytes) travels over xGMI while the encoders' backward is still computing.
        With the reducer attached, calling it after loss.backward() is a no-op for that step."""
        model._grad_sync = self
        return self

    @staticmethod
    def active():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def begin(self):
        self._pending, self._sent = [], set()

    def ready(self, grads):
        names = [n for n in grads if n not in self._sent and grads[n] is not None]
        if not names:
            return
        self._sent.update(names)
        shapes = [tuple(grads[n].shape) for n in names]
        flat = torch.cat([grads[n].reshape(-1) for n in names])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._pending.append((work, flat, names, shapes))

    def finish(self, grads):
        world = dist.get_world_size()
        for work, flat, names, shapes in self._pending:
            work.wait()
            flat.mul_(1.0 / world)
            off = 0
            for n, shp in zip(names, shapes):
                cnt = 1
                for d in shp:
                    cnt *= d
                grads[n] = flat[off:off + cnt].view(shp)
                off += cnt
        self._pending = []
        self._done_in_backward = True

    def __call__(self):
        if getattr(self, "_done_in_backward", False):      # already averaged inside the backward of this step
            self._done_in_backward = False
            return
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        if self._flat is None:
            self._flat = [torch.empty((sum(p.numel() for p in b),), device=b[0].device, dtype=torch.float32)
                          for b in self.buckets]
        works = []
        for flat, bucket in zip(self._flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for w, flat, bucket in zip(works, self._flat, self.buckets):
            w.wait()
            flat.mul_(1.0 / world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
