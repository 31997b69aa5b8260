"""RCCL on the one GPU a test box has: a single-rank `nccl` process group.  A 1-rank all-reduce moves no data between
devices, but it runs the real RCCL code path the 8-GPU job uses — communicator creation, ncclAvg support, the collective
enqueued on RCCL's stream behind the backward kernels of the launching stream, work.wait() — and the flat gradient arena
end to end with the HIP backward: gradients must come out bit-identical to the run without a process group.
(Two ranks cannot share one device under RCCL, and the GPU boxes of this pool have one; N > 1 semantics are covered over
gloo in tests/test_harness_gloo.py.)  Runs in a child process: a process group is per-process state."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import golden_util as G
from ccvpe_amd import harness, models, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
SD = synth.synthetic_state_dict(G.TRAIN_CASE["kind"], G.TRAIN_CASE["wseed"])     # generated once (seconds on the host)

def step(with_pg):
    c = G.TRAIN_CASE
    net = models.CVM_VIGOR("cuda", c["circular"])
    net.load_state_dict(SD, strict=True)
    net = net.to(dev).train()
    red = harness.GradientAllReducer(net.parameters()).attach(net)
    if with_pg:
        harness.GradientAllReducer.active = staticmethod(lambda: True)      # a 1-rank group: force the collective path
    grd, sat = synth.synthetic_pair(c["batch"], c["grd"], c["pseed"])
    masks, _, _ = G.train_drop_masks(c["batch"])
    out = net(grd.cuda(), sat.cuda(), drop_masks=masks)
    G.train_loss(out).backward()
    red()
    torch.cuda.synchronize()
    flat = red._arena["flat"]
    views_ok = all(p.grad is None or p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in net.parameters())
    return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}, red.allreduce_calls, views_ok

base, calls0, ok0 = step(False)
assert calls0 == 0 and ok0
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "%(port)d"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.full((1024,), 3.0, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.AVG)                                    # ncclAvg is available in this RCCL
assert float(t[0]) == 3.0
got, calls1, ok1 = step(True)
assert calls1 == 3 and ok1, (calls1, ok1)
assert set(got) == set(base) and len(got) >= 500
for n in base:
    assert torch.equal(base[n], got[n]), n

# Adam per gradient group inside the backward (attach(step_in_backward=True)): same parameters, bit for bit, as backward ->
# reducer() -> optimizer.step(); the caller's step() afterwards must not update anything twice
from ccvpe_amd import optim
def train2(in_backward):
    c = G.TRAIN_CASE
    torch.manual_seed(7)
    net = models.CVM_VIGOR("cuda", c["circular"])
    net.load_state_dict(SD, strict=True)
    net = net.to(dev).train()
    opt = optim.Adam(net.parameters(), lr=1e-4)
    red = harness.GradientAllReducer(net.parameters()).attach(net, optimizer=opt, step_in_backward=in_backward)
    grd, sat = synth.synthetic_pair(c["batch"], c["grd"], c["pseed"])
    masks, _, _ = G.train_drop_masks(c["batch"])
    for it in range(1):
        opt.zero_grad(set_to_none=True)
        out = net(grd.cuda(), sat.cuda(), drop_masks=masks)
        G.train_loss(out).backward()
        red()
        opt.step()
    torch.cuda.synchronize()
    steps = {float(opt.state[p]["step"]) for p in net.parameters() if p in opt.state and len(opt.state[p])}
    return {n: p.detach().clone() for n, p in net.named_parameters()}, steps
pa, sa = train2(False)
pb, sb = train2(True)
assert sa == {1.0} and sb == {1.0}, (sa, sb)      # every parameter stepped exactly once: step() skipped what step_subset() did
moved = 0
sd0 = SD
for n in pa:
    assert torch.equal(pa[n], pb[n]), n
    moved += int(not torch.equal(pa[n], sd0[n].to(dev)))
assert moved >= 450, moved      # (42 tensors have a mathematically zero gradient; their noise-level updates may round to nothing)
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK", len(got))
'''


def test_rccl_allreduce_and_arena_on_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, port=port)], capture_output=True, text=True, env=env,
                         timeout=900)
    assert res.returncode == 0 and "RCCL_SINGLE_RANK_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
