"""Eager Python forward vs ccvpe_forward (one C call per forward) vs hipGraph replay, ms per forward.
   gpurun -- python tools/plan_probe.py [fp32|bf16] [batches...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccvpe_amd import models, plan, synth
from ccvpe_amd.graph import GraphedForward

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
batches = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 16, 64]
sd = synth.synthetic_state_dict("vigor", 0)
net = models.CVM_VIGOR_ori_prior("cuda", 0, True)
net.load_state_dict(sd, strict=True)
net = net.to("cuda:0").eval().set_precision(prec)


def wall(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for b in batches:
    grd, sat = synth.synthetic_pair(b, "vigor", 1)
    grd, sat = grd.cuda(), sat.cuda()
    n = 50 if b <= 16 else 15
    te = wall(lambda: net(grd, sat), n)
    pf = plan.PlannedForward(net, grd, sat)
    tp = wall(lambda: pf(grd, sat), n)
    gf = GraphedForward(net, grd, sat)
    tg = wall(lambda: gf(grd, sat), n)
    print("%s B=%-3d eager (2 streams) %.3f ms | ccvpe_forward (%d calls incl. stream waits, workspace %.0f MiB) %.3f ms | hipGraph replay %.3f ms"
          % (prec, b, te, len(pf.plan.calls), pf.plan.workspace_bytes / 2 ** 20, tp, tg), flush=True)
    del pf, gf
