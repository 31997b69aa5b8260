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


# ----------------------------------------------------------------------------------------------------------
# The REAL autograd node (ccvpe_amd.train.CVMFunction) with the flat gradient arena: forward_train / backward_train are
# replaced by CPU stand-ins that produce gradients in the same three groups, through the same on_ready callback, as the
# HIP backward; everything else (CVMFunction.backward -> sync.begin / ready / finish, the arena slots, p.grad views,
# the fall-back to autograd accumulation) is the product code.
# ----------------------------------------------------------------------------------------------------------
_STUB_SHAPES = {"conv6.0.weight": (8, 5, 3, 3), "deconv6.bias": (7,), "sat_feature_to_descriptors.1.weight": (6, 10),
                "sat_efficientnet._conv_stem.weight": (4, 3, 3, 3), "sat_efficientnet._fc.weight": (5, 4),
                "grd_feature_to_descriptor1.0.weight": (3, 2, 1, 1), "grd_efficientnet._conv_stem.weight": (4, 3, 3, 3),
                "grd_efficientnet._bn0.weight": (4,)}


class _StubModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._names = {}
        for i, (n, shp) in enumerate(_STUB_SHAPES.items()):
            key = n.replace(".", "__")
            self.register_parameter(key, torch.nn.Parameter(torch.zeros(shp)))
            self._names[key] = n

    def named_parameters(self, *a, **k):           # the reference's dotted names
        for key, p in super().named_parameters(*a, **k):
            yield self._names[key], p


def _stub_grad(name, rank, step):
    shp = _STUB_SHAPES[name]
    base = torch.arange(float(torch.Size(shp).numel())).reshape(shp) + 100.0 * (sorted(_STUB_SHAPES).index(name) + 1)
    return base * (rank + 1) + step


def _arena_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccvpe_amd import harness, train
    state = {"step": 0, "order": []}

    def fake_forward(model, grd, sat, drop_masks=None, rec=False):
        outs = tuple(torch.zeros(2, 3) for _ in range(9))
        return outs, {"live": dict(model.named_parameters())}

    def fake_backward(model, tape, gout, on_ready=None):
        from ccvpe_amd.harness import grad_group
        grads = {}
        for g in range(3):
            for n in _STUB_SHAPES:
                if grad_group(n) == g and "._fc." not in n:
                    v = _stub_grad(n, rank, state["step"])
                    # like the HIP backward: some gradients arrive as permuted (non-contiguous) views
                    grads[n] = v.permute(*reversed(range(v.dim()))).contiguous().permute(*reversed(range(v.dim()))) if v.dim() == 4 else v
            if on_ready is not None:
                state["order"].append(g)
                on_ready(grads)
        return grads

    train.forward_train, train.backward_train = fake_forward, fake_backward
    model = _StubModel()
    red = harness.GradientAllReducer(model.parameters()).attach(model)
    res = {}
    for step in range(2):
        state["step"] = step
        for p in model.parameters():
            p.grad = None
        outs = train.apply(model, torch.zeros(1), torch.zeros(1))
        sum(o.sum() for o in outs).backward()
        red()                                                     # a no-op: already averaged inside the backward
        res[step] = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
        res[step, "is_view"] = all(p.grad is None or p.grad.untyped_storage().data_ptr() == red._arena["flat"].untyped_storage().data_ptr()
                                   for p in model.parameters())
    res["calls_after_two_steps"] = red.allreduce_calls
    # third backward WITHOUT clearing the gradients: accumulation goes through autograd, the reducer then averages the
    # accumulated gradients with the bucketed post-backward path
    state["step"] = 2
    outs = train.apply(model, torch.zeros(1), torch.zeros(1))
    sum(o.sum() for o in outs).backward()
    red()
    res["acc"] = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
    res["order"] = state["order"]
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_arena_through_the_real_autograd_node(tmp_path):
    out = str(tmp_path / "arena.pt")
    mp.spawn(_arena_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out, weights_only=False)
    for step in range(2):
        for n in _STUB_SHAPES:
            g = res[step][n]
            if "._fc." in n:
                assert g is None
            else:
                want = (_stub_grad(n, 0, step) + _stub_grad(n, 1, step)) / 2
                assert torch.allclose(g, want), (step, n)
        assert res[step, "is_view"]
    assert res["calls_after_two_steps"] == 6                      # three group collectives per step, nothing else
    assert res["order"][:3] == [0, 1, 2]
    for n in _STUB_SHAPES:                                        # step 1's mean + mean of step 2's gradients ...
        if "._fc." in n:
            continue
        mean1 = (_stub_grad(n, 0, 1) + _stub_grad(n, 1, 1)) / 2
        mean2 = (_stub_grad(n, 0, 2) + _stub_grad(n, 1, 2)) / 2
        # ... averaged once more by the post-backward path: ranks hold (mean1 + g_r), its rank mean is mean1 + mean2
        assert torch.allclose(res["acc"][n], mean1 + mean2), n


def _infonce_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccvpe_amd import harness
    from oracle import ccvpe_oracle as O
    torch.manual_seed(5)
    scores_all = torch.rand(4, 257) * 2 - 1
    labels_all = torch.rand(4, 257) * (torch.rand(4, 257) > 0.7)       # ragged label mass per sample
    labels_all[3] *= 3.0
    w = torch.randn(257, 257) * 0.05
    w.requires_grad_(True)                                         # a shared "parameter" the scores depend on
    mine = slice(2 * rank, 2 * rank + 2)
    sc = scores_all[mine] @ w
    lab = labels_all[mine]
    local = O.infonce_loss(sc, lab)
    den = torch.where(lab > 1e-2, lab, torch.zeros_like(lab)).sum()
    loss = harness.global_ratio_loss(local, den)
    loss.backward()
    g = w.grad.clone()
    dist.all_reduce(g)
    g /= world                                                     # data-parallel gradient averaging
    if rank == 0:
        torch.save((loss.detach(), g, local.detach()), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exact_infonce_equals_one_rank_big_batch(tmp_path):
    """losses.py:18 normalises by the label mass of the WHOLE batch: the 2-scalar all-reduce (harness.global_ratio_loss,
    used by losses.infoNCELoss_global) makes two ranks x 2 samples give the loss AND the averaged gradient of one process
    with 4 samples — which the mean of per-rank losses does not."""
    sys.path.insert(0, ROOT)
    from oracle import ccvpe_oracle as O
    out = str(tmp_path / "nce.pt")
    mp.spawn(_infonce_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    loss2, grad2, local0 = torch.load(out, weights_only=False)
    torch.manual_seed(5)
    scores_all = torch.rand(4, 257) * 2 - 1
    labels_all = torch.rand(4, 257) * (torch.rand(4, 257) > 0.7)
    labels_all[3] *= 3.0
    w = torch.randn(257, 257) * 0.05
    w.requires_grad_(True)
    big = O.infonce_loss(scores_all @ w, labels_all)
    big.backward()
    assert abs(float(loss2) - float(big.detach())) <= 1e-6 * abs(float(big.detach()))
    assert torch.allclose(grad2, w.grad, rtol=1e-4, atol=1e-7)
    assert abs(float(local0) - float(big.detach())) > 1e-4 * abs(float(big.detach()))      # the per-rank loss is a different number


def test_step_in_backward_refuses_a_backward_without_the_optimizer_step_in_between():
    """with step_in_backward=True a second backward without optimizer.step() (gradient accumulation, a step
    skipped after an inf / nan check) used to make step_subset() a silent no-op and the next step() skip every parameter."""
    import pytest
    import torch
    from ccvpe_amd import harness

    class Opt(object):
        _pre_stepped = set()

        def step_subset(self, params):
            pass

    p = torch.nn.Parameter(torch.zeros(4))
    red = harness.GradientAllReducer([p])
    red._optimizer, red._step_in_backward = Opt(), True
    red.begin()                                   # nothing pending: fine
    red._optimizer._pre_stepped = {id(p)}         # an in-backward update happened, optimizer.step() did not
    with pytest.raises(RuntimeError, match="optimizer.step"):
        red.begin()
    red._optimizer._pre_stepped = set()           # what optimizer.step() leaves behind
    red.begin()
