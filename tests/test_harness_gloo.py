"""N>1 path on CPU: two processes over gloo exercise the replica harness bench.py uses
(barrier-bracketed timing, MAX over ranks, whole-job aggregation, sample sharding)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccvpe_amd import harness
    import time
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 * (rank + 1))         # rank 1 is the slow one

    elapsed = harness.timed_steps(step, steps=5, warmup=2)
    thr = harness.aggregate_throughput(4, 5, elapsed)
    shard = harness.shard_indices(11, world, rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (len(calls), elapsed, thr, shard))
    if rank == 0:
        torch.save(gathered, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replica_harness(tmp_path):
    out = str(tmp_path / "res.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=False)
    (c0, e0, t0, s0), (c1, e1, t1, s1) = res
    assert c0 == c1 == 7                                  # 2 warm-up + exactly 5 timed steps each
    assert abs(e0 - e1) < 1e-9                            # both ranks hold the MAX
    assert e0 >= 5 * 0.04 * 0.9                           # ... which is the slow rank's time
    assert abs(t0 - 4 * 2 * 5 / e0) < 1e-9                # whole-job units / slowest time
    assert sorted(s0 + s1) == list(range(11)) and not set(s0) & set(s1)
    assert len(s0) == 6 and len(s1) == 5


def _grad_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccvpe_amd import harness
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((300, 7), (5,), (64, 3, 3, 3), (1000,), (2, 2))]
    params[1].requires_grad_(False)
    for i, p in enumerate(params):
        if p.requires_grad and i != 4:                   # params[4] has no gradient on any rank
            p.grad = torch.full_like(p, float(rank + 1)) * (i + 1)
    red = harness.GradientAllReducer(params, bucket_bytes=9000)      # forces several buckets
    red()
    res = [None if p.grad is None else p.grad.clone() for p in params]
    if rank == 0:
        torch.save((res, len(red.buckets)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    out = str(tmp_path / "grads.pt")
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res, nb = torch.load(out, weights_only=False)
    assert nb >= 3
    for i in (0, 2, 3):
        assert torch.allclose(res[i], torch.full_like(res[i], 1.5 * (i + 1)))     # mean of (1, 2) * (i + 1)
    assert res[1] is None and res[4] is None


def _overlap_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccvpe_amd import harness
    red = harness.GradientAllReducer([])
    assert red.active()
    red.begin()
    grads = {}
    # three groups arriving one after the other, with a permuted (non-contiguous) gradient and a None in between
    grads["a"] = torch.full((4, 3), float(rank + 1))
    grads["b"] = torch.arange(24.0).reshape(2, 3, 4).permute(0, 2, 1) * (rank + 1)
    red.ready(grads)
    grads["c"] = None
    grads["d"] = torch.full((7,), 10.0 * (rank + 1))
    red.ready(grads)
    red.ready(grads)                                  # nothing new: must be a no-op
    red.finish(grads)
    if rank == 0:
        torch.save(grads, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlapped_gradient_groups(tmp_path):
    """The in-backward (overlapped) path of GradientAllReducer: groups are reduced as they become ready."""
    out = str(tmp_path / "ov.pt")
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g = torch.load(out, weights_only=False)
    assert torch.allclose(g["a"], torch.full((4, 3), 1.5))
    assert torch.allclose(g["b"], torch.arange(24.0).reshape(2, 3, 4).permute(0, 2, 1) * 1.5)
    assert g["c"] is None and torch.allclose(g["d"], torch.full((7,), 15.0))
